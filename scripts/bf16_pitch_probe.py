"""Is the bf16 GEMM limited by the operands' row pitch (power-of-two leading dimensions -> channel conflicts)?
Dense zk_linear_bf16 at N = 2^17, OUT = 8192 for several IN (= row pitch of both operands).  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd import ops
dev = torch.device("cuda:0")
N, OUT = 1 << 17, 8192
g = torch.Generator(device=dev).manual_seed(0)
for IN in (1024, 1088, 960, 1152, 2048, 2112, 512, 576):
    x = torch.randn(N, IN, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(OUT, IN, generator=g, device=dev) / 32).to(torch.bfloat16)
    with torch.no_grad():
        for _ in range(2): y = ops.linear_bf16(x, w, None, None, 1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): y = ops.linear_bf16(x, w, None, None, 1)
        b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"IN={IN:5d} (pitch {IN*2:5d} B): {ms:7.3f} ms  {2*N*IN*OUT/ms/1e9:7.1f} TF/s = {2*N*IN*OUT/ms/1e9/25:.1f}% of 2.5 PF")
    del x, w, y
