"""GPU: flow().transform.inv(z) of the cfg2 / cfg3 flows — the incremental inverse launch with its pull phase on the f32 matrix instruction
("bf16x3" mode: nothing of this kernel is split) and on the f16 one with the two-part operand split (HALF instantiation, "f16x2" mode)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zuko_amd
from zuko_amd.flows import MAF, NSF

dev = torch.device("cuda:0")
for lg in (18, 20):
    B = 1 << lg
    for name, make in (("NSF cfg2", lambda: NSF(64, 0, transforms=8, hidden_features=[256] * 3)), ("MAF cfg3", lambda: MAF(64, 0, transforms=8, hidden_features=[256] * 3))):
        torch.manual_seed(0)
        flow = make().to(dev)
        z = torch.randn(B, 64, generator=torch.Generator().manual_seed(1)).to(dev)
        res = {}
        for mode in ("bf16x3", "f16x2"):
            zuko_amd.set_matmul_precision(mode)
            with torch.no_grad():
                x = flow().transform.inv(z); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    x = flow().transform.inv(z)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 3
                zz = flow().transform(x)
            res[mode] = (dt, x, (zz - z).abs().max().item())
        d = (res["bf16x3"][1] - res["f16x2"][1]).abs().max().item()
        print(f"{name}: batch 2^{lg}: f32-instruction pulls {res['bf16x3'][0]*1e3:.1f} ms ({B/res['bf16x3'][0]/1e6:.2f} M samples/s, round trip {res['bf16x3'][2]:.2e}); "
              f"two-part f16 pulls {res['f16x2'][0]*1e3:.1f} ms ({B/res['f16x2'][0]/1e6:.2f} M samples/s, round trip {res['f16x2'][2]:.2e}); max |dx| = {d:.2e}", flush=True)
