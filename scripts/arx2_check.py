#!/usr/bin/env python
"""Two-set operand-split kernel (csrc/fused_ar_split2_impl.h) against its 8-wavefront form (csrc/fused_ar_split_impl.h) on the GPU:
bit-identity of y / ladj on several conditioners and batch shapes, then launch times of both at the headline batch.

    python scripts/arx2_check.py [--time-only] [--label NAME]       (ZUKO_AMD_ARX2_QB / _FILL / ZUKO_AMD_CACHE_DIR select a variant build)
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import zuko_amd.flows as ZF  # noqa: E402
from zuko_amd import _C  # noqa: E402

dev = torch.device("cuda", 0)


def both(flow, x, c=None):
    out = []
    for v1 in ("1", "0"):
        os.environ["ZUKO_AMD_SPLIT_V1"] = v1
        with torch.no_grad():
            d = flow(c) if c is not None else flow()
            z, ladj = d.transform.call_and_ladj(x)
        out.append((z.clone(), ladj.clone()))
    os.environ["ZUKO_AMD_SPLIT_V1"] = "0"
    return out


def same(a, b):
    return bool(torch.equal(torch.nan_to_num(a, nan=12345.0, posinf=2e30, neginf=-2e30), torch.nan_to_num(b, nan=12345.0, posinf=2e30, neginf=-2e30)))


def identity_checks():
    cases = [
        ("NSF cfg2", lambda: ZF.NSF(64, 0, transforms=2, bins=8, hidden_features=[256] * 3), 64, 0),
        ("MAF cfg3", lambda: ZF.MAF(64, 0, transforms=2, hidden_features=[256] * 3), 64, 0),
        ("NSF 32 [256]^2", lambda: ZF.NSF(32, 0, transforms=2, bins=8, hidden_features=[256] * 2), 32, 0),
        ("MAF 16 [128]^2", lambda: ZF.MAF(16, 0, transforms=2, hidden_features=[128] * 2), 16, 0),
        ("NSF 20 ctx 3 [100,72]", lambda: ZF.NSF(20, 3, transforms=2, bins=8, hidden_features=[100, 72]), 20, 3),
    ]
    ok = True
    for name, make, D, C in cases:
        torch.manual_seed(3)
        flow = make().to(dev)
        st = flow.transform.transforms[0].fused_state(dev)
        split = bool(st is not None and st.static is not None and st.static[0].meta.get("split"))
        for N in (1, 100, 128, 4133, 1 << 16):
            g = torch.Generator().manual_seed(N)
            x = (1.5 * torch.randn(N, D, generator=g)).to(dev)
            c = torch.randn(N, C, generator=g).to(dev) if C else None
            if N >= 100:  # poisoned rows: NaN / inf inputs, a value outside the spline's support
                x[7, 3] = float("nan")
                x[11, D - 1] = float("inf")
                x[13, 0] = 7.5
            (z1, l1), (z2, l2) = both(flow, x, c)
            good = same(z1, z2) and same(l1, l2)
            ok &= good
            nd = int((torch.nan_to_num(z1, nan=1.0) != torch.nan_to_num(z2, nan=1.0)).sum())
            print(f"{name:24s} N={N:6d} split={split} bit-identical={good}" + ("" if good else f"  differing y: {nd}, max |dy| {float((z1 - z2).abs().nan_to_num().max()):.3e}, max |dladj| {float((l1 - l2).abs().nan_to_num().max()):.3e}"), flush=True)
    return ok


def timing(label):
    torch.manual_seed(0)
    res = {"label": label, "qb": os.environ.get("ZUKO_AMD_ARX2_QB", "8"), "fill": os.environ.get("ZUKO_AMD_ARX2_FILL", "2")}
    for name, make, D in (("cfg2", lambda: ZF.NSF(64, 0, transforms=8, bins=8, hidden_features=[256] * 3), 64), ("cfg3", lambda: ZF.MAF(64, 0, transforms=8, hidden_features=[256] * 3), 64)):
        flow = make().to(dev)
        x = torch.randn(1 << 20, D, generator=torch.Generator().manual_seed(1)).to(dev)
        for v1 in ("1", "0"):
            os.environ["ZUKO_AMD_SPLIT_V1"] = v1
            try:
                with torch.no_grad():
                    for _ in range(3):
                        flow().log_prob(x)
                    torch.cuda.synchronize()
                    _C.PROFILE = {}
                    for _ in range(5):
                        flow().log_prob(x)
                    torch.cuda.synchronize()
                    prof, _C.PROFILE = _C.PROFILE, None
            except RuntimeError:  # (probe builds hold the two-set kernel only)
                _C.PROFILE = None
                res[f"{name}_{'v1_8wave' if v1 == '1' else 'v2_twoset'}_ms"] = None
                continue
            ts = [a.elapsed_time(b) for a, b, _ in prof.get("zk_ar_forward_static", [])]
            ts.sort()
            res[f"{name}_{'v1_8wave' if v1 == '1' else 'v2_twoset'}_ms"] = {"median": ts[len(ts) // 2], "min": ts[0], "calls": len(ts)} if ts else None
        os.environ["ZUKO_AMD_SPLIT_V1"] = "0"
        del flow, x
    print(json.dumps(res), flush=True)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--time-only", action="store_true")
    ap.add_argument("--label", default="default")
    args = ap.parse_args()
    good = True
    if not args.time_only:
        good = identity_checks()
        print("ALL BIT-IDENTICAL" if good else "MISMATCH", flush=True)
    timing(args.label)
    sys.exit(0 if good else 1)
