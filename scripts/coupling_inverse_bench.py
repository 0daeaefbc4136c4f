"""Side measurement: RealNVP cfg4 inverse (sampling direction) on the fused coupling kernel vs the layer-wise path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zuko_amd
from zuko_amd.flows import RealNVP

dev = torch.device("cuda:0")
B = 1 << int(os.environ.get("LOG2B", 19))
torch.manual_seed(0)
flow = RealNVP(features=256, context=0, transforms=16, hidden_features=[512] * 3).to(dev)
z = torch.randn(B, 256, generator=torch.Generator().manual_seed(1)).to(dev)
res = {}
for mode in ("0", "1"):
    os.environ["ZUKO_AMD_NO_FUSED_COUPLING"] = mode
    with torch.no_grad():
        x = flow().transform.inv(z); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            x = flow().transform.inv(z)
        torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t0) / 3, x)
os.environ["ZUKO_AMD_NO_FUSED_COUPLING"] = "0"
with torch.no_grad():
    back = flow().transform(res["0"][1])
print(f"RealNVP cfg4 inverse, batch 2^{B.bit_length()-1}: fused {res['0'][0]*1e3:.1f} ms ({B/res['0'][0]/1e6:.2f} M samples/s), layer-wise {res['1'][0]*1e3:.1f} ms "
      f"({B/res['1'][0]/1e6:.2f} M samples/s); max |x_fused - x_layerwise| = {(res['0'][1]-res['1'][1]).abs().max().item():.2e}; round trip {(back - z).abs().max().item():.2e}")
