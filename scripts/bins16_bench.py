#!/usr/bin/env python
"""NSF(64, bins=16, H=[256]^3) log_prob at 2^20 rows: the generated operand-split kernel against the generic f32 kernel (ZUKO_AMD_NO_STATIC_AR=1)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import zuko_amd.flows as ZF  # noqa: E402

dev = torch.device("cuda", 0)
out = {}
x = torch.randn(1 << 20, 64, generator=torch.Generator().manual_seed(1)).to(dev)
for name, env in (("split_static", "0"), ("generic_f32", "1")):
    os.environ["ZUKO_AMD_NO_STATIC_AR"] = env
    torch.manual_seed(0)
    flow = ZF.NSF(64, 0, transforms=8, bins=16, hidden_features=[256] * 3).to(dev)
    with torch.no_grad():
        for _ in range(3):
            lp = flow().log_prob(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            lp = flow().log_prob(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
    out[name] = {"ms_per_log_prob": ms, "samples_per_s": (1 << 20) / (ms * 1e-3), "lp_mean": float(lp.mean())}
print(json.dumps(out))
