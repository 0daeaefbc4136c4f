#!/usr/bin/env python
"""257 - 512 wide conditioners: the operand-split kernel (one wavefront per SIMD) against the f32-instruction static-shape kernel (ZUKO_AMD_SPLIT_WIDE=0),
NSF log_prob at batch 2^18."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zuko_amd.flows as F
from zuko_amd.flows import autoregressive as AR

dev = torch.device("cuda:0")
B = 1 << 18
for D, ctx, hidden in ((32, 0, [512, 512]), (24, 8, [384, 512, 320])):
    torch.manual_seed(0)
    flow = F.NSF(D, ctx, transforms=4, bins=8, hidden_features=hidden).to(dev)
    x = torch.randn(B, D, device=dev)
    c = torch.randn(B, ctx, device=dev) if ctx else None
    out = {}
    for mode in ("split", "f32"):
        os.environ["ZUKO_AMD_SPLIT_WIDE"] = "1" if mode == "split" else "0"
        for lazy in flow.transform.transforms:
            AR._FUSED_CACHE.pop(lazy, None)
        with torch.no_grad():
            for _ in range(2):
                lp = flow(c).log_prob(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                lp = flow(c).log_prob(x)
            torch.cuda.synchronize()
            out[mode] = ((time.perf_counter() - t0) / 5, lp)
    d = ((out["split"][1] - out["f32"][1]).abs().max() / out["f32"][1].abs().max()).item()
    print(f"NSF({D}, ctx {ctx}, T=4, H={hidden}) log_prob at 2^18: operand-split {out['split'][0]*1e3:.2f} ms ({B/out['split'][0]/1e6:.1f} M samples/s), f32 instruction "
          f"{out['f32'][0]*1e3:.2f} ms ({B/out['f32'][0]/1e6:.1f} M samples/s); max rel log_prob difference {d:.1e}", flush=True)
