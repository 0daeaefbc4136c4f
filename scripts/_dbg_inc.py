import sys, torch, ctypes, os, time
sys.path.insert(0, '/root/repo')
import zuko_amd
from zuko_amd import _C
import zuko_amd.flows as F
dev = torch.device('cuda:0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
orig = _C.lib().zk_ar_inverse_incremental
libs = {}
for tag in ("product", "base", "nop", "drain", "nr4"):
    if tag != "product":
        cd = ctypes.CDLL(os.path.join(ROOT, "zuko_amd", "lib", "incdbg", f"libinc_{tag}.so"))
        fn = cd.zk_ar_inverse_incremental
        fn.argtypes = _C.SIGNATURES["zk_ar_inverse_incremental"]; fn.restype = ctypes.c_int
        libs[tag] = cd
        _C.lib().zk_ar_inverse_incremental = fn
    else:
        _C.lib().zk_ar_inverse_incremental = orig
    for kind in ("maf", "nsf"):
        torch.manual_seed(0)
        flow = (F.MAF(64, 0, transforms=1, hidden_features=[256] * 3) if kind == "maf" else F.NSF(64, 0, transforms=1, hidden_features=[256] * 3)).to(dev)
        z = torch.randn(1 << 16, 64, generator=torch.Generator().manual_seed(4)).to(dev)
        res, tms = {}, {}
        with torch.no_grad():
            for mode in ("bf16x3", "f16x2"):
                zuko_amd.set_matmul_precision(mode)
                t = flow.transform.transforms[0]
                x = t().inv(z); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3): x = t().inv(z)
                torch.cuda.synchronize()
                tms[mode] = (time.perf_counter() - t0) / 3 * 1e3
                res[mode] = x
        d = (res["bf16x3"] - res["f16x2"]).abs()
        print(f"{tag:8s} {kind}: max |dx| {d.max().item():.2e}, bad wave tiles {int((d.view(-1, 16, 64).amax(dim=(1, 2)) > 1e-4).sum())} of {d.shape[0] // 16}; ms f32 {tms['bf16x3']:.3f} half {tms['f16x2']:.3f}", flush=True)
