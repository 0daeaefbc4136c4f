#!/usr/bin/env python
"""Per-step shader-clock trace of the 32-sample split kernel (a -DARX3_TRACE build selected through ZUKO_AMD_CACHE_DIR):
prints the cycles of every step of workgroup 0 / wavefront 0 in its first two passes, summed per layer / feature group."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import zuko_amd.flows as ZF  # noqa: E402
from zuko_amd import static_ar  # noqa: E402
from zuko_amd.nn import MaskedLinear  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
flow = ZF.NSF(64, 0, transforms=1, bins=8, hidden_features=[256] * 3).to(dev)
lazy = flow.transform.transforms[0]
st = lazy.fused_state(dev)
assert st.ready(1 << 20) and st.static is not None and st.static[0].meta.get("split")
st.refresh([m for m in lazy.hyper if isinstance(m, MaskedLinear)])
t = st.t3
assert st.static3 is not None
N = 1 << 17  # (four passes per workgroup; the knots buffer must be real: without a trace build the diagnostic kernel writes it)
x = torch.randn(N, 64, generator=torch.Generator().manual_seed(1)).to(dev)
y, ladj = torch.empty_like(x), torch.empty(N, device=dev)
bins = torch.zeros(N, 64, dtype=torch.int32, device=dev)
knots = torch.zeros(N, 64, 9, device=dev)
for _ in range(3):
    st.run_diag(x, y, ladj, bins, knots)
torch.cuda.synchronize()
nsteps = t["HS_OFF"][-1] + t["LS_OFF"][-1]
raw = bins.flatten()[: 2 * (nsteps + 1)].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
out = {}
for p in range(2):
    ts = raw[p * (nsteps + 1) : (p + 1) * (nsteps + 1)]
    d = np.diff(ts) & 0xFFFFFFFF
    hs, ls = t["HS_OFF"], t["LS_OFF"]
    layers = [int(d[hs[l] : hs[l + 1]].sum()) for l in range(t["NH"])]
    groups = [int(d[hs[-1] + ls[g] : hs[-1] + ls[g + 1]].sum()) for g in range(t["NG3"])]
    out[f"pass{p}"] = {"total": int(d.sum()), "hidden_layers": layers, "hidden_steps": [hs[l + 1] - hs[l] for l in range(t["NH"])],
                       "groups": groups, "group_steps": [ls[g + 1] - ls[g] for g in range(t["NG3"])],
                       "per_step": d.tolist()}
print(json.dumps(out))
