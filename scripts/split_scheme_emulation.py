#!/usr/bin/env python
"""CPU emulation of the conditioner's matrix products under different operand splits (round 6: is a cheaper split than 3 x bf16 / 6 products
within reach of the parity bar?).  NSF cfg2, seed-0 weights, N rows of N(0,1); every layer's products are evaluated as the matrix instruction
does (exact products of the parts, f32 accumulator updated once per 32-deep block), the spline by the float32 oracle on the emulated parameters.
Reported against the float64 oracle, next to the float32 reference's own distance from it.

    python scripts/split_scheme_emulation.py [rows]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zuko_oracle as O  # noqa: E402

f32, f64 = np.float32, np.float64


def bf16(v):
    u = v.astype(f32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(f32)


def f16(v):
    return v.astype(np.float16).astype(f32)


def split(v, rnd, parts):
    out, r = [], v.astype(f32)
    for _ in range(parts):
        p = rnd(r)
        out.append(p)
        r = (r - p).astype(f32)
    return out


SCHEMES = {
    # name: (rounding, parts, [(activation part, weight part) ... smallest first])
    "bf16x3, 6 products (shipped)": (bf16, 3, [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]),
    "f16x2, 3 products": (f16, 2, [(1, 0), (0, 1), (0, 0)]),
    "f16x2, 4 products": (f16, 2, [(1, 1), (1, 0), (0, 1), (0, 0)]),
    "bf16x2, 3 products": (bf16, 2, [(1, 0), (0, 1), (0, 0)]),
    "f16x2, 3 products, scaled": (f16, 2, [(1, 0), (0, 1), (0, 0)]),
}


def pow2_to(v, target=15):
    """2^k with max|v| * 2^k in [2^(target-1), 2^target) (1 where v is all zero)."""
    m = np.abs(v).max(axis=-1, keepdims=True) if v.ndim > 1 else np.abs(v).max()
    e = np.frexp(np.where(m > 0, m, 1.0))[1]  # m = f * 2^e, f in [0.5, 1)
    return np.exp2((target - e).astype(f64)).astype(f64)


def linear_scaled(x, W, b, scheme):
    """f16 parts with a static per-layer weight scale and a dynamic per-sample activation scale (both powers of two: exact); the accumulator is
    descaled and the bias added in ONE fma at the end — the form a kernel would use."""
    rnd, parts, prods = SCHEMES[scheme]
    sw = f64(pow2_to(W.reshape(-1)))
    sx = pow2_to(x)  # [N, 1]
    xs = split((x.astype(f64) * sx).astype(f32), rnd, parts)
    ws = split((W.astype(f64) * sw).astype(f32), rnd, parts)
    N, K = x.shape
    acc = np.zeros((N, W.shape[0]), f32)
    for k0 in range(0, K, 32):
        sl = slice(k0, k0 + 32)
        for ia, iw in prods:
            acc = (acc.astype(f64) + xs[ia][:, sl].astype(f64) @ ws[iw][:, sl].astype(f64).T).astype(f32)
    return (acc.astype(f64) / (sx * sw) + b.astype(f64)).astype(f32)  # one rounding: fma(acc, d, bias)


def linear(x, W, b, scheme, wscale=1.0):
    if scheme.endswith("scaled"):
        return linear_scaled(x, W, b, scheme)
    rnd, parts, prods = SCHEMES[scheme]
    xs = split(x, rnd, parts)
    ws = split((W * f32(wscale)).astype(f32), rnd, parts)
    N, K = x.shape
    acc = np.broadcast_to(b.astype(f32) * f32(wscale), (N, W.shape[0])).astype(f32).copy()
    for k0 in range(0, K, 32):
        sl = slice(k0, k0 + 32)
        for ia, iw in prods:
            acc = (acc.astype(f64) + xs[ia][:, sl].astype(f64) @ ws[iw][:, sl].astype(f64).T).astype(f32)
    return (acc / f32(wscale)).astype(f32) if wscale != 1.0 else acc


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    import zuko_amd.flows as ZF

    torch.manual_seed(0)
    flow = ZF.NSF(features=64, context=0, transforms=8, bins=8, hidden_features=[256] * 3)
    regime = os.environ.get("REGIME", "init")
    xs_ = 1.0
    g = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for lazy in flow.transform.transforms:
            lins = [m for m in lazy.hyper if hasattr(m, "mask")]
            if regime == "trained":
                for l in lins[:-1]:
                    l.weight.mul_(30.0 ** (1.0 / 3.0) * 1.5)
                lins[-1].weight.mul_(0.05)
                xs_ = 2.0
            elif regime == "wide-range":
                for l in lins:
                    u = torch.rand(l.weight.shape, generator=g) * 24.0 - 20.0
                    l.weight.mul_(torch.exp2(u))
            elif regime == "denormal":
                lins[0].weight.mul_(2.0 ** -120)
                lins[0].bias.mul_(2.0 ** -120)
                lins[1].weight.mul_(2.0 ** 100)
    sd = {k: v for k, v in flow.state_dict().items() if v is not None}
    spec = O.spec_from_state_dict(sd, "ar", O.uni_rqs(8), 64)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    spec64 = O.spec_from_state_dict(sd64, "ar", O.uni_rqs(8), 64)
    x = torch.randn(N, 64, generator=torch.Generator().manual_seed(1)) * xs_
    print(f"regime {regime}, {N} rows")
    with torch.no_grad():
        z64, l64 = O.flow_forward(spec64, x.double())
        lp64 = O.diag_normal_log_prob(z64, spec64.loc, spec64.scale) + l64
        z32, l32 = O.flow_forward(spec, x)
        lp32 = O.diag_normal_log_prob(z32, spec.loc, spec.scale) + l32

    def report(name, z, l, lp):
        ez, el = (z.double() - z64).abs().max().item(), (l.double() - l64).abs().max().item()
        rl = ((lp.double() - lp64).abs() / lp64.abs()).max().item()
        rz32, rl32 = (z - z32).abs().max().item(), ((lp - lp32).abs() / lp32.abs()).max().item()
        strict = bool(torch.allclose(z, z32, rtol=1e-5, atol=1e-5)) and bool(torch.allclose(l, l32, rtol=1e-5, atol=1e-5))
        print(f"{name:32s} vs f64: z {ez:.2e} ladj {el:.2e} log_prob rel {rl:.2e} | vs f32 reference: z {rz32:.2e} log_prob rel {rl32:.2e} ladj {(l - l32).abs().max().item():.2e} allclose(1e-5,1e-5) z&ladj: {strict}")
        return ez, el, rl

    base = report("float32 reference (oracle)", z32, l32, lp32)
    for scheme in SCHEMES:
        with torch.no_grad():
            h = x.clone()
            ladj = torch.zeros(N)
            flips = 0
            for layer, layer64 in zip(spec.layers, spec64.layers):
                v = h.numpy()
                n = len(layer.weights)
                for i in range(n):
                    W = (layer.masks[i] * layer.weights[i]).numpy()
                    v = linear(v, W, layer.biases[i].numpy(), scheme)
                    if i + 1 < n:
                        v = np.maximum(v, 0)
                phi = torch.from_numpy(v).unflatten(-1, (-1, layer.uni.total))
                y, lj = O.univariate_forward(layer.uni, phi, h)
                h, ladj = y, ladj + lj.sum(-1)
            lp = O.diag_normal_log_prob(h, spec.loc, spec.scale) + ladj
        e = report(scheme, h, ladj, lp)
        print(f"{'':32s} error ratio to the f32 reference's own: z {e[0] / base[0]:.2f} ladj {e[1] / base[1]:.2f} log_prob {e[2] / base[2]:.2f}")


if __name__ == "__main__":
    main()
