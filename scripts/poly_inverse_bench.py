#!/usr/bin/env python
"""SOSPF / BPF flow().transform.inv (64 features, 3 transforms, hidden [256] * 3): ONE incremental launch per transform with the bisection in the kernel's group
epilogue (round 6) against the wavefront form of the reference's loop on the layer-wise kernels (round 5; ZUKO_AMD_NO_INCREMENTAL=1), same weights, same z."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd.flows import BPF, SOSPF

dev = torch.device("cuda:0")
for name, ctor in (("SOSPF", SOSPF), ("BPF", BPF)):
    torch.manual_seed(0)
    flow = ctor(64, 0, transforms=3, hidden_features=[256] * 3).to(dev)
    for lg in (14, 18):
        B = 1 << lg
        with torch.no_grad():
            tr = flow().transform
            x0 = 0.8 * torch.randn(B, 64, device=dev)
            z = tr(x0)
            res = {}
            for mode in ("incremental", "layer-wise wavefront"):
                if mode != "incremental":
                    if lg > 14:
                        continue
                    os.environ["ZUKO_AMD_NO_INCREMENTAL"] = "1"
                try:
                    xs = tr.inv(z); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(2):
                        xs = tr.inv(z)
                    torch.cuda.synchronize()
                    res[mode] = ((time.perf_counter() - t0) / 2, xs)
                finally:
                    os.environ.pop("ZUKO_AMD_NO_INCREMENTAL", None)
            back = (res["incremental"][1] - x0).abs().max().item()
            line = f"{name} inverse at 2^{lg}: incremental launch {res['incremental'][0]*1e3:.2f} ms ({B/res['incremental'][0]/1e6:.2f} M samples/s), round trip max |x - x0| {back:.2e}"
            if "layer-wise wavefront" in res:
                d = (res["incremental"][1] - res["layer-wise wavefront"][1]).abs().max().item()
                line += f"; layer-wise wavefront {res['layer-wise wavefront'][0]*1e3:.1f} ms ({B/res['layer-wise wavefront'][0]/1e6:.2f} M samples/s), max |dx| {d:.2e}"
            print(line, flush=True)
