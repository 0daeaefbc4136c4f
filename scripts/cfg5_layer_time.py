#!/usr/bin/env python
"""One autoregressive layer at cfg5's shape (D = 1024, 16 bins, hidden 1024^3, bf16) at N = 2^LOG2N rows: per-kernel time of the
fused last layer (zk_linear_bf16_rqs) and of the hidden layers (zk_linear_bf16), plus checksums of y / ladj so that library
variants (ZUKO_AMD_LIB=...) can be compared for identical results.  One JSON line."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd import _C
from zuko_amd.flows import MaskedAutoregressiveTransform
from zuko_amd.transforms import MonotonicRQSTransform

dev = torch.device("cuda:0")
N = 1 << int(os.environ.get("LOG2N", "19"))
D, K = 1024, 16
torch.manual_seed(5)
t = MaskedAutoregressiveTransform(D, 0, univariate=MonotonicRQSTransform, shapes=[(K,), (K,), (K - 1,)], hidden_features=[1024] * 3).to(dev).to(torch.bfloat16)
x = torch.randn(N, D, generator=torch.Generator().manual_seed(9)).to(torch.bfloat16).to(dev)
with torch.no_grad():
    for _ in range(2):
        y, l = t().call_and_ladj(x)
    torch.cuda.synchronize()
    _C.PROFILE = {}
    for _ in range(int(os.environ.get("REPS", "3"))):
        y, l = t().call_and_ladj(x)
    torch.cuda.synchronize()
    prof, _C.PROFILE = _C.PROFILE, None
out = {"lib": os.path.basename(_C.LIB_PATH), "rows": N}
for name, recs in prof.items():
    ts = sorted(a.elapsed_time(b) for a, b, _ in recs)
    out[name] = {"calls": len(ts), "median_ms": round(ts[len(ts) // 2], 3), "min_ms": round(ts[0], 3)}
out["y_sha"] = hashlib.sha256(y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
out["ladj_sum"] = float(l.double().sum())
print(json.dumps(out))
