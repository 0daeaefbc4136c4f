#!/usr/bin/env python
"""Per-launch time of one autoregressive transform on (a) its generated static-shape kernel, (b) the generic tile-skipping kernel
(widths <= 256) and (c) the layer-wise kernels, with the fraction of the matrix peak on the non-zero weights (operand-split kernels: 2500 / 6
TFLOP/s, f32-instruction kernels: 157.3) and the largest relative difference of (y, ladj) to the other two paths.

    python scripts/static_shapes_bench.py [log2 batch]      -> one JSON line per shape
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import zuko_amd.flows as F
from zuko_amd.flows import autoregressive as AR
from zuko_amd.nn import MaskedLinear

PEAK = 157.3e12          # f32 matrix instruction
PEAK_SPLIT = 2500e12 / 6  # dense bf16 / six partial products per f32 product
SHAPES = [("nsf", 64, 0, [256] * 3), ("maf", 64, 0, [256] * 3), ("nsf", 3, 5, [128] * 3), ("nsf", 32, 0, [256, 256]), ("nsf", 32, 0, [512, 512]), ("maf", 16, 0, [128, 128]),
          ("nsf", 128, 0, [256] * 3), ("nsf", 64, 8, [256] * 2), ("nsf", 16, 0, [512] * 3), ("nsf16", 64, 0, [256] * 3)]


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    lb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    os.environ["ZUKO_AMD_JIT_MIN_ROWS"] = "1"
    for kind, D, C, hidden in SHAPES:
        N = 1 << (lb if max(hidden) <= 256 or True else lb - 1)
        torch.manual_seed(0)
        flow = (F.NSF(D, C, transforms=2, bins=16 if kind == "nsf16" else 8, hidden_features=hidden) if kind.startswith("nsf") else F.MAF(D, C, transforms=2, hidden_features=hidden)).to(dev)
        x = torch.randn(N, D, device=dev)
        c = torch.randn(N, C, device=dev) if C else None
        lazy = flow.transform.transforms[0]
        nnz = 2 * sum(int(m.mask.sum()) for m in lazy.hyper if isinstance(m, MaskedLinear))
        rec = {"shape": f"{kind}({D}, ctx {C}, {hidden})", "rows": N, "nnz_flop_per_row": nnz}
        with torch.no_grad():
            t0 = time.time()
            t = lazy(c)
            st = t._fused(x)
            rec["jit_or_load_s"] = round(time.time() - t0, 2)
            rec["static"] = None if st is None or st.static is None else {"waves": st.static[0].meta["WAVES"], "xlds": st.static[0].meta["XLDS"], "split": bool(st.static[0].meta.get("split"))}
            if st is not None and st.static is not None:
                ms = timed(lambda: lazy(c).call_and_ladj(x))
                rec["static_ms"] = round(ms, 4)
                rec["static_frac_nnz"] = round(N * nnz / (ms * 1e-3) / (PEAK_SPLIT if rec["static"]["split"] else PEAK), 4)
                rec["static_tflops_nnz"] = round(N * nnz / (ms * 1e-3) / 1e12, 1)
                y_s, l_s = lazy(c).call_and_ladj(x)
            if st is not None and st.generic_ok:
                os.environ["ZUKO_AMD_NO_STATIC_AR"] = "1"  # (static_ar.lookup returns None: the plan's generic kernels run)
                AR._FUSED_CACHE.pop(lazy, None)
                stg = lazy(c)._fused(x)
                assert stg.static is None
                if stg._gsplit() is not None:  # the generic operand-split kernel (csrc/fused_ar_gsplit.hip): what such a plan runs on by default
                    ms = timed(lambda: lazy(c).call_and_ladj(x))
                    y_x, l_x = lazy(c).call_and_ladj(x)
                    rec["generic_split_ms"] = round(ms, 4)
                    rec["generic_split_frac_nnz"] = round(N * nnz / (ms * 1e-3) / PEAK_SPLIT, 4)
                    if rec["static"] is not None and rec["static"]["split"]:
                        rec["generic_split_over_static"] = round(ms / rec["static_ms"], 3)
                        rec["generic_split_bit_identical_to_static"] = bool(torch.equal(y_x, y_s) and torch.equal(l_x, l_s))
                os.environ["ZUKO_AMD_GSPLIT"] = "0"  # the generic kernel on the f32 matrix instruction
                AR._FUSED_CACHE.pop(lazy, None)
                ms = timed(lambda: lazy(c).call_and_ladj(x))
                y_g, l_g = lazy(c).call_and_ladj(x)
                os.environ.pop("ZUKO_AMD_NO_STATIC_AR")
                os.environ.pop("ZUKO_AMD_GSPLIT")
                AR._FUSED_CACHE.pop(lazy, None)
                rec["generic_ms"] = round(ms, 4)
                rec["generic_frac_nnz"] = round(N * nnz / (ms * 1e-3) / PEAK, 4)
                if rec["static"] is not None:
                    rec["max_rel_vs_generic_f32_kernel"] = [float(((y_s - y_g).abs().max() / y_g.abs().max())), float(((l_s - l_g).abs().max() / l_g.abs().max()))]
            if N * D * lazy.total * 4 <= 24 << 30:  # layer-wise: phi [N, D * total] through HBM
                if True:
                    from functools import partial

                    from zuko_amd.transforms import AutoregressiveTransform

                    lw = lambda: AutoregressiveTransform(partial(lazy.meta, c), lazy.passes).call_and_ladj(x)
                    ms = timed(lw, reps=3)
                    rec["layerwise_ms"] = round(ms, 4)
                    if rec["static"] is not None:  # (the layer-wise kernels share no code with the fused ones; both are held to the oracle in tests/)
                        y_l, l_l = lw()
                        rec["max_rel_vs_layerwise_kernels"] = [float(((y_s - y_l).abs().max() / y_l.abs().max())), float(((l_s - l_l).abs().max() / l_l.abs().max()))]
        print(json.dumps(rec), flush=True)
        del flow, x, c
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
