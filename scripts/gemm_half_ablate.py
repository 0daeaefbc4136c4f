"""Removal probes of zk_gemm_f16x2 (scripts/probes/ab/gh_abl_<n>.so = csrc/gemm_half.hip built with -DGH_ABL=<n>; wrong results, timing only)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zuko_amd import _C, coupling_train as ct

dev = torch.device("cuda:0")
names = {0: "product build", 1: "no conversion arithmetic", 2: "no matrix instructions", 3: "no raw-tile DMA", 4: "no weight DMA", 5: "no stores", 6: "no fragment reads (one LDS address)"}
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
for (M, K, N, gate) in ((16384, 512, 512, False), (16384, 512, 512, True), (65536, 512, 512, False)):
    a = torch.randn(M, K, device=dev).clamp_min(0)
    w = torch.randn(N, K, device=dev) / K**0.5
    b = torch.randn(N, device=dev)
    gt = torch.randn(M, N, device=dev) if gate else None
    am = torch.zeros(3, ct.AMAX_WORDS, dtype=torch.int32, device=dev)
    ct.amax([(a, am[0]), (w, am[1])])
    img = torch.empty(ct.image_words(N, K), dtype=torch.int32, device=dev)
    ct.wsplit([(w, False, am[1], img)])
    c = torch.empty(M, N, device=dev)
    for n in range(7):
        lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "probes", "ab", f"gh_abl_{n}.so"))
        f = lib.zk_gemm_f16x2
        f.argtypes = _C.SIGNATURES["zk_gemm_f16x2"]
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: f(M, K, N, P(a), a.stride(0), P(am[0]), P(img), P(am[1]), P(b), 0 if gate else 1, P(gt), 0 if gt is None else gt.stride(0), 1, P(c), c.stride(0), P(am[2]), st)
        for _ in range(3): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        print(f"M={M} K={K} N={N} gate={int(gate)}  {names[n]:38s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
