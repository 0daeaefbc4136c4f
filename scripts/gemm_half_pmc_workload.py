"""30 launches of zk_gemm_f16x2 on 16384 x 512 x 512 (forward shape of RealNVP cfg4's hidden layers) for rocprofv3 counter passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd import coupling_train as ct
dev = torch.device("cuda:0")
M, K, N = 16384, 512, 512
a = torch.randn(M, K, device=dev).clamp_min(0); w = torch.randn(N, K, device=dev) / K**0.5; b = torch.randn(N, device=dev)
am = torch.zeros(3, ct.AMAX_WORDS, dtype=torch.int32, device=dev)
ct.amax([(a, am[0]), (w, am[1])])
img = torch.empty(ct.image_words(N, K), dtype=torch.int32, device=dev)
ct.wsplit([(w, False, am[1], img)])
for _ in range(30):
    c = ct.gemm(a, am[0], img, am[1], N, b, 1, None, am[2])
torch.cuda.synchronize()
