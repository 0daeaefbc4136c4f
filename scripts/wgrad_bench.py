#!/usr/bin/env python
"""Weight-gradient kernel alone: dW = mask * g^T h for the cfg2 conditioner's layers at batch 2^16 (ms per call, f32-equivalent TFLOP/s on
the live 128 x 128 blocks).  ZUKO_AMD_LIB=<probe build> times an ablation (scripts/build_tu_variant.sh train wgK -DZK_WG_ABL=K)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd import train
from zuko_amd.flows import NSF
from zuko_amd.nn import MaskedLinear

dev = torch.device("cuda:0")
torch.manual_seed(0)
flow = NSF(64, 0, transforms=1, bins=8, hidden_features=[256] * 3).to(dev)
lins = [m for m in flow.transform.transforms[0].hyper if isinstance(m, MaskedLinear)]
plan = train.SortedPlan(lins, 1, dev)
N = 1 << 16
tot = 0.0
for l, (out_f, in_f) in enumerate(plan.shapes):
    g, h = torch.randn(N, out_f, device=dev), torch.randn(N, in_f, device=dev)
    for _ in range(3):
        plan.wgrad(l, g, h, want_bias=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        plan.wgrad(l, g, h, want_bias=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    tot += ms
    live = plan.pairs[l].shape[0]
    print(f"layer {l}: dW[{out_f}, {in_f}] {live} live blocks: {ms:.3f} ms = {2.0 * N * live * 128 * 128 / ms / 1e9:.1f} TFLOP/s", flush=True)
print(f"sum {tot:.3f} ms per transform ({os.environ.get('ZUKO_AMD_LIB', 'library')})")
