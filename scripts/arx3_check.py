#!/usr/bin/env python
"""32-sample operand-split kernel (csrc/fused_ar_split3_impl.h) against the 8-wavefront one (csrc/fused_ar_split_impl.h) on the GPU:
agreement of y / ladj on several conditioners and batch shapes (same formulation, another summation order inside the matrix
instruction: a few f32 ulps), identical NaN patterns, then launch times of both at the headline batch.

    python scripts/arx3_check.py [--time-only] [--label NAME]       (ZUKO_AMD_ARX2_FILL / ZUKO_AMD_CACHE_DIR select a variant build)
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


def close(a, b):
    """(identical NaN / inf pattern, max |a - b| over the finite part, its scale)"""
    fa, fb = torch.isfinite(a), torch.isfinite(b)
    pattern = bool(torch.equal(fa, fb) and torch.equal(torch.isnan(a), torch.isnan(b)))
    ok = fa & fb
    if not ok.any():
        return pattern, 0.0, 1.0
    return pattern, float((a[ok] - b[ok]).abs().max()), float(b[ok].abs().max())


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
        v3 = bool(st is not None and getattr(st, "static3", None) is not None)
        for N in (1, 100, 128, 4133, 1 << 16):
            g = torch.Generator().manual_seed(N)
            x = (1.5 * torch.randn(N, D, generator=g)).to(dev)
            c = torch.randn(N, C, generator=g).to(dev) if C else None
            if N >= 100:  # poisoned rows: NaN / inf inputs, a value outside the spline's support
                x[7, 3] = float("nan")
                x[11, D - 1] = float("inf")
                x[13, 0] = 7.5
            (z1, l1), (z2, l2) = both(flow, x, c)
            pz, ez, sz = close(z2, z1)
            pl, el, sl = close(l2, l1)
            good = pz and pl and ez <= 2e-5 * max(sz, 1.0) and el <= 1e-4 * max(sl, 1.0)
            ok &= good
            print(f"{name:24s} N={N:6d} v3={v3} patterns={pz and pl} max|dy|={ez:.2e} (scale {sz:.1f}) max|dladj|={el:.2e} (scale {sl:.1f}) {'ok' if good else 'MISMATCH'}", flush=True)
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
                res[f"{name}_{'v1_8wave' if v1 == '1' else 'v3_32sample'}_ms"] = None
                continue
            ts = [a.elapsed_time(b) for a, b, _ in prof.get("zk_ar_forward_static", [])]
            ts.sort()
            res[f"{name}_{'v1_8wave' if v1 == '1' else 'v3_32sample'}_ms"] = {"median": ts[len(ts) // 2], "min": ts[0], "calls": len(ts)} if ts else None
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
        print("ALL CLOSE" if good else "MISMATCH", flush=True)
    timing(args.label)
    sys.exit(0 if good else 1)
