"""zk_gemm_f16x2 (csrc/gemm_half.hip) against float64 matmul and against zk_gemm_f32_skip: accuracy and time (run on the GPU box)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd import _C

dev = torch.device("cuda:0")
lib = _C.lib()
P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
S = lambda: torch.cuda.current_stream().cuda_stream


def amax(items):
    cls = _C.STRUCTS["zk_amax_desc_v1"]
    arr = (cls * len(items))()
    for d, (t, out) in zip(arr, items):
        d.struct_size, d.rows, d.cols, d.ld, d.src, d.out = ctypes.sizeof(cls), t.shape[0], t.shape[1], t.stride(0), t.data_ptr(), out.data_ptr()
    _C.check(lib.zk_amax_f32(len(items), ctypes.cast(arr, ctypes.c_void_p), S()), "zk_amax_f32")


def wsplit(items):
    cls = _C.STRUCTS["zk_wsplit_desc_v1"]
    arr = (cls * len(items))()
    for d, (w, units, k, su, sk, am, dst) in zip(arr, items):
        d.struct_size, d.units, d.k, d.unit_stride, d.k_stride, d.src, d.amax, d.dst = ctypes.sizeof(cls), units, k, su, sk, w.data_ptr(), am.data_ptr(), dst.data_ptr()
    _C.check(lib.zk_wsplit_f16(len(items), ctypes.cast(arr, ctypes.c_void_p), S()), "zk_wsplit_f16")


def images(units, k):
    return torch.empty(-(-units // 128) * -(-k // 32) * 16384 // 4, dtype=torch.int32, device=dev)


def run(M, K, N, act, gate, scale_a=1.0, scale_w=1.0, time_it=True):
    g = torch.Generator(device="cpu").manual_seed(M + K + N)
    a = (torch.randn(M, K, generator=g) * scale_a).to(dev)
    if act:  # post-activation like inputs
        a = a.clamp_min(0)
    w = (torch.randn(N, K, generator=g) * scale_w / K**0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev) * scale_a * scale_w
    gt = torch.randn(M, N, generator=g).to(dev) if gate else None
    am = torch.zeros(4, 2048, dtype=torch.int32, device=dev)
    amax([(a, am[0]), (w, am[1])])
    img = images(N, K)
    wsplit([(w, N, K, K, 1, am[1], img)])
    c = torch.empty(M, N, device=dev)
    call = lambda: _C.check(lib.zk_gemm_f16x2(M, K, N, P(a), a.stride(0), P(am[0]), P(img), P(am[1]), P(b), act, P(gt), 0 if gt is None else gt.stride(0), 1, P(c), c.stride(0), P(am[2]), S()), "zk_gemm_f16x2")
    call()
    ref = a.double() @ w.double().t() + b.double()
    if act:
        ref = ref.clamp_min(0)
    if gate:
        ref = ref * (gt > 0)
    c32 = torch.empty(M, N, device=dev)
    old = lambda: _C.check(lib.zk_gemm_f32_skip(M, K, N, P(a), a.stride(0), P(w), None, P(b), act, P(gt), 0 if gt is None else gt.stride(0), 1, P(c32), c32.stride(0), S()), "zk_gemm_f32_skip")
    old()
    den = ref.abs().max().item()
    e_new, e_old = (c.double() - ref).abs().max().item() / den, (c32.double() - ref).abs().max().item() / den
    cmax = torch.tensor([c.abs().max().item()]).view(torch.int32).item()
    line = f"M={M} K={K} N={N} act={act} gate={int(gate)} scale=({scale_a:g},{scale_w:g}): max err / max|ref| two-part f16 {e_new:.2e}  f32 MFMA {e_old:.2e}  amax out ok {am[2].max().item() == cmax}"
    if time_it:
        for fn, name in ((call, "f16x2"), (old, "f32")):
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 20
            line += f"  {name} {t*1e3:.1f} us ({2.0*M*K*N/t/1e9:.0f} TF/s)"
    print(line, flush=True)
    return e_new


worst = 0.0
for (M, K, N) in ((16384, 512, 512), (16384, 128, 512), (16384, 512, 256), (16384, 256, 512), (16384, 512, 128), (65536, 512, 512)):
    worst = max(worst, run(M, K, N, 1, False))
    worst = max(worst, run(M, K, N, 0, True))
for (M, K, N) in ((1000, 72, 40), (129, 8, 4), (1, 512, 512), (300, 200, 132)):
    worst = max(worst, run(M, K, N, 1, False, time_it=False))
worst = max(worst, run(4096, 512, 512, 1, False, 1e-6, 1e3, time_it=False))
worst = max(worst, run(4096, 512, 512, 0, True, 1e5, 1e-4, time_it=False))
print("worst", worst)
assert worst < 2e-6
