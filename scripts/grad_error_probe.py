#!/usr/bin/env python
"""Where does the training path's gradient error come from?  NSF cfg2, N rows: every parameter gradient of the HIP path against float64 autograd
through the oracle, next to the float32 reference's own distance, under the environment switches that select the backward kernels.
    python scripts/grad_error_probe.py [rows] [switch=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    os.environ[k] = v
import torch
from oracle import zuko_oracle as O
import zuko_amd.flows as F

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
flow = F.NSF(64, 0, transforms=8, bins=8, hidden_features=[256] * 3)
x = torch.randn(n, 64, generator=torch.Generator().manual_seed(22))

def oracle_grads(dtype):
    sd = {k: (v.detach().to(dtype) if v.is_floating_point() else v.detach()) for k, v in flow.state_dict().items() if v is not None}
    leaves = {k: v.requires_grad_() for k, v in sd.items() if v.is_floating_point() and ("weight" in k or "bias" in k)}
    sd.update(leaves)
    spec = O.spec_from_state_dict(sd, "ar", O.uni_rqs(8), 64)
    xr = x.to(dtype).requires_grad_()
    loss = -O.flow_log_prob(spec, xr, None).mean()
    loss.backward()
    return {k: v.grad for k, v in leaves.items()}, xr.grad

g64, gx64 = oracle_grads(torch.float64)
g32, gx32 = oracle_grads(torch.float32)
dev = torch.device("cuda:0")
flow = flow.to(dev)
xg = x.to(dev).requires_grad_()
loss = -flow().log_prob(xg).mean()
loss.backward()
params = dict(flow.named_parameters())
print("switches:", " ".join(sys.argv[2:]) or "(none)", " rows:", n)
worst = 0
for k, g in g64.items():
    sc = g.abs().max().clamp_min(1e-12)
    mine = params[k].grad.cpu().double()
    e, er = ((mine - g).abs().max() / sc).item(), ((g32[k].double() - g).abs().max() / sc).item()
    worst = max(worst, e)
    if e > 1e-5:
        d = (mine - g).abs() / sc
        nbad = int((d > 1e-5).sum())
        print(f"  {k:44s} err {e:.2e} (float32 reference {er:.2e})  elements above 1e-5: {nbad} of {d.numel()}  mean {d.mean().item():.2e}  rel-to-own-mean {((mine - g).abs().mean() / g.abs().mean()).item():.2e}")
sx = gx64.abs().max()
print(f"  worst {worst:.2e}; grad x {((xg.grad.cpu().double() - gx64).abs().max() / sx).item():.2e} (float32 reference {((gx32.double() - gx64).abs().max() / sx).item():.2e})")
