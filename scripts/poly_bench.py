#!/usr/bin/env python
"""SOSPF / BPF log_prob (64 features, 3 transforms, hidden [256] * 3) at batch 2^18: the fused operand-split kernels (uni kinds 5 / 6) against the
layer-wise path they replace (conditioner GEMMs, phi in HBM, stand-alone polynomial kernel), same weights."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd.flows import BPF, SOSPF
from zuko_amd.flows import autoregressive as AR

dev = torch.device("cuda:0")
B = 1 << 18
for name, ctor in (("SOSPF", SOSPF), ("BPF", BPF)):
    torch.manual_seed(0)
    flow = ctor(64, 0, transforms=3, hidden_features=[256] * 3).to(dev)
    x = torch.randn(B, 64, device=dev)
    res = {}
    for mode in ("fused", "layer-wise"):
        orig = AR.FusedAutoregressiveTransform._fused
        if mode == "layer-wise":
            AR.FusedAutoregressiveTransform._fused = lambda self, x, need_generic=False: None
        try:
            with torch.no_grad():
                for _ in range(2):
                    lp = flow().log_prob(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    lp = flow().log_prob(x)
                torch.cuda.synchronize()
                res[mode] = ((time.perf_counter() - t0) / 5, lp)
        finally:
            AR.FusedAutoregressiveTransform._fused = orig
    d = (res["fused"][1] - res["layer-wise"][1]).abs().max().item()
    print(f"{name}(64, T=3, H=[256]*3) log_prob at 2^18: fused {res['fused'][0]*1e3:.2f} ms ({B/res['fused'][0]/1e6:.1f} M samples/s), layer-wise {res['layer-wise'][0]*1e3:.2f} ms "
          f"({B/res['layer-wise'][0]/1e6:.1f} M samples/s); max |log_prob difference| {d:.2e}", flush=True)
