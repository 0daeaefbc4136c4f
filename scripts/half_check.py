#!/usr/bin/env python
"""GPU: the two-part (f16 x 2) inference kernel beside the three-part (bf16 x 3) one on NSF cfg2 — log_prob / z / ladj against the float32 and
float64 oracle on 4096 rows, per-launch time at 2^20 rows (events around whole log_prob calls: 8 transforms + base density)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zuko_amd  # noqa: E402
import zuko_amd.flows as ZF  # noqa: E402
from oracle import zuko_oracle as O  # noqa: E402

dev = torch.device("cuda:0")
torch.set_num_threads(16)
kind = sys.argv[1] if len(sys.argv) > 1 else "nsf"
torch.manual_seed(0)
flow_cpu = ZF.NSF(features=64, context=0, transforms=8, bins=8, hidden_features=[256] * 3) if kind == "nsf" else ZF.MAF(features=64, context=0, transforms=8, hidden_features=[256] * 3)
sd = {k: v for k, v in flow_cpu.state_dict().items() if v is not None}
uni = O.uni_rqs(8) if kind == "nsf" else O.UNI_AFFINE
spec = O.spec_from_state_dict(sd, "ar", uni, 64)
spec64 = O.spec_from_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, "ar", uni, 64)
flow = (ZF.NSF(features=64, context=0, transforms=8, bins=8, hidden_features=[256] * 3) if kind == "nsf" else ZF.MAF(features=64, context=0, transforms=8, hidden_features=[256] * 3))
flow.load_state_dict(flow_cpu.state_dict())
flow = flow.to(dev)
xs = torch.randn(4096, 64, generator=torch.Generator().manual_seed(1))
with torch.no_grad():
    z32, l32 = O.flow_forward(spec, xs)
    lp32 = O.diag_normal_log_prob(z32, spec.loc, spec.scale) + l32
    z64, l64 = O.flow_forward(spec64, xs.double())
    lp64 = O.diag_normal_log_prob(z64, spec64.loc, spec64.scale) + l64
print(f"float32 reference vs float64: z {(z32 - z64).abs().max():.2e} ladj {(l32 - l64).abs().max():.2e} log_prob rel {((lp32 - lp64).abs() / lp64.abs()).max():.2e}")
x = torch.randn(1 << 20, 64, device=dev)
res = {}
for mode in ("bf16x3", "f16x2", "bf16x3", "f16x2"):
    zuko_amd.set_matmul_precision(mode)
    with torch.no_grad():
        d = flow()
        z, ladj = d.transform.call_and_ladj(xs.to(dev))
        lp = d.log_prob(xs.to(dev))
        st = flow.transform.transforms[0].fused_state(dev)
        used = "two-part" if st._half_serves(torch.empty(4, 64, device=dev)) else "three-part"
        z, ladj, lp = z.cpu(), ladj.cpu(), lp.cpu()
        for _ in range(3):
            flow().log_prob(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            flow().log_prob(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{mode:7s} ({used:10s}) vs f64: z {(z - z64).abs().max():.2e} ladj {(ladj - l64).abs().max():.2e} log_prob rel {((lp - lp64).abs() / lp64.abs()).max():.2e} | vs f32 reference: z {(z - z32).abs().max():.2e} "
          f"ladj {(ladj - l32).abs().max():.2e} log_prob rel {((lp - lp32).abs() / lp32.abs()).max():.2e} | log_prob at 2^20: {ms:.2f} ms = {(1 << 20) / ms / 1e3:.1f} M samples/s", flush=True)
