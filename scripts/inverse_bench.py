"""Side measurement: autoregressive inverse (sampling direction) — partial (wavefront) sweeps vs full sweeps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zuko_amd
from zuko_amd.flows import MAF, NSF

dev = torch.device("cuda:0")
B = 1 << int(os.environ.get("LOG2B", 18))
for name, make in (("NSF cfg2", lambda: NSF(64, 0, transforms=8, hidden_features=[256] * 3)), ("MAF cfg3", lambda: MAF(64, 0, transforms=8, hidden_features=[256] * 3))):
    torch.manual_seed(0)
    flow = make().to(dev)
    z = torch.randn(B, 64, generator=torch.Generator().manual_seed(1)).to(dev)
    res = {}
    for mode in ("0", "1"):
        os.environ["ZUKO_AMD_FULL_SWEEPS"] = mode
        with torch.no_grad():
            x = flow().transform.inv(z); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                x = flow().transform.inv(z)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 2
            zz = flow().transform(x)
        res[mode] = (dt, x, (zz - z).abs().max().item())
    d = (res["0"][1] - res["1"][1]).abs().max().item()
    print(f"{name}: batch 2^{B.bit_length()-1}: partial sweeps {res['0'][0]*1e3:.1f} ms ({B/res['0'][0]/1e6:.2f} M samples/s), full sweeps {res['1'][0]*1e3:.1f} ms "
          f"({B/res['1'][0]/1e6:.2f} M samples/s); max |x_partial - x_full| = {d:.2e}; round trip {res['0'][2]:.2e}")
