#!/usr/bin/env python
"""Training step of a CONDITIONAL flow — NSF(3, context 5, transforms 3, hidden [128] * 3) = BASELINE cfg1, and NSF(60, context 4, hidden [256] * 3) — at batch 2^16:
the one-node path (cat(x, c) through AutoregressiveFn) against the two-node path (ZUKO_AMD_NO_FUSED_AR_TRAIN=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd.flows import NSF

dev = torch.device("cuda:0")
B = 1 << 16
for D, C, T, H in ((3, 5, 3, [128] * 3), (60, 4, 8, [256] * 3)):
    for mode in ("one node", "two nodes"):
        os.environ["ZUKO_AMD_NO_FUSED_AR_TRAIN"] = "0" if mode == "one node" else "1"
        os.environ["ZUKO_AMD_JIT_MIN_ROWS"] = "1"
        torch.manual_seed(0)
        flow = NSF(D, C, transforms=T, bins=8, hidden_features=H).to(dev)
        opt = torch.optim.Adam(flow.parameters(), lr=1e-3)
        x, c = torch.randn(B, D, device=dev), torch.randn(B, C, device=dev)

        def step():
            loss = -flow(c).log_prob(x).mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            return loss

        for _ in range(3):
            l0 = step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            l = step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"NSF({D}, context {C}, T={T}, H={H}) Adam step at 2^16, {mode}: {dt*1e3:.2f} ms ({B/dt/1e6:.2f} M samples/s), loss {float(l0):.3f} -> {float(l):.3f}", flush=True)
