#!/usr/bin/env python
"""One-transform NSF: per-sample error of d loss / d x and the spline's per-sample state (bin, position in the bin) for the worst samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for kv in sys.argv[2:]:
    k, v = kv.split("="); os.environ[k] = v
import torch
from oracle import zuko_oracle as O
import zuko_amd.flows as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
full = F.NSF(64, 0, transforms=8, bins=8, hidden_features=[256] * 3)
x0 = torch.randn(n, 64, generator=torch.Generator().manual_seed(22))
# the input the LAST transform sees in the 8-transform flow (so that the case is the failing one), computed by the oracle
sdf = {k: v.detach() for k, v in full.state_dict().items() if v is not None}
specf = O.spec_from_state_dict(sdf, "ar", O.uni_rqs(8), 64)
with torch.no_grad():
    h = x0
    for layer in specf.layers[:7]:
        h, _ = O.layer_forward(layer, h, None)
x = h.detach()
flow = F.NSF(64, 0, transforms=1, bins=8, hidden_features=[256] * 3)
sd1 = flow.state_dict()
for k in list(sd1):
    if k.startswith("transform.transforms.0."):
        sd1[k] = sdf[k.replace("transforms.0.", "transforms.7.")].clone()
flow.load_state_dict(sd1)

def oracle(dtype):
    sd = {k: (v.detach().to(dtype) if v.is_floating_point() else v.detach()) for k, v in flow.state_dict().items() if v is not None}
    leaves = {k: v.requires_grad_() for k, v in sd.items() if v.is_floating_point() and ("weight" in k or "bias" in k)}
    sd.update(leaves)
    spec = O.spec_from_state_dict(sd, "ar", O.uni_rqs(8), 64)
    xr = x.to(dtype).requires_grad_()
    lp = O.flow_log_prob(spec, xr, None)
    (-lp.mean()).backward()
    return {k: v.grad for k, v in leaves.items()}, xr.grad, lp.detach()
g64, gx64, lp64 = oracle(torch.float64)
g32, gx32, _ = oracle(torch.float32)
dev = torch.device("cuda:0")
flow = flow.to(dev)
xg = x.to(dev).requires_grad_()
lp = flow().log_prob(xg)
(-lp.mean()).backward()
print("switches:", " ".join(sys.argv[2:]) or "(none)")
print("log_prob max abs err vs f64: %.2e" % (lp.detach().cpu().double() - lp64).abs().max().item())
for k, g in g64.items():
    sc = g.abs().max()
    print(f"  {k:40s} hip {((dict(flow.named_parameters())[k].grad.cpu().double() - g).abs().max() / sc).item():.2e}   float32 reference {((g32[k].double() - g).abs().max() / sc).item():.2e}")
# forward noise: what the last transform is fed by the HIP path / by the float32 reference, against float64
with torch.no_grad():
    h64 = x0.double()
    spec64 = O.spec_from_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sdf.items()}, "ar", O.uni_rqs(8), 64)
    for layer in spec64.layers[:7]:
        h64, _ = O.layer_forward(layer, h64, None)
    fullg = full.to(dev)
    hh = x0.to(dev)
    for t in list(fullg.transform.transforms)[:7]:
        hh = t()(hh)
    print("input of the last transform: |hip - f64| max %.2e, |float32 reference - f64| max %.2e" % ((hh.cpu().double() - h64).abs().max().item(), (x.double() - h64).abs().max().item()))
    d = (hh.cpu().double() - h64).abs()
    print("   rows with |hip - f64| > 1e-5:", int((d.max(dim=1).values > 1e-5).sum()), "of", n, "; 99.9th percentile %.2e" % torch.quantile(d.flatten(), 0.999).item())
# gradient of the one-transform flow fed with the HIP path's own input
def oracle_in(inp):
    sd = {k: (v.detach().double() if v.is_floating_point() else v.detach()) for k, v in flow.state_dict().items() if v is not None}
    sd = {k: v.cpu() for k, v in sd.items()}
    leaves = {k: v.requires_grad_() for k, v in sd.items() if v.is_floating_point() and ("weight" in k or "bias" in k)}
    sd.update(leaves)
    spec = O.spec_from_state_dict(sd, "ar", O.uni_rqs(8), 64)
    (-O.flow_log_prob(spec, inp.double(), None).mean()).backward()
    return {k: v.grad for k, v in leaves.items()}
ga, gb = oracle_in(h64), oracle_in(hh.cpu())
for k in ga:
    print(f"  sensitivity {k:40s} float64 gradient at the HIP input vs at the float64 input: {((ga[k] - gb[k]).abs().max() / ga[k].abs().max()).item():.2e}")
