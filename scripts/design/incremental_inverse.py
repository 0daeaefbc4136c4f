"""Design study (CPU, numpy, float64) for the next sampling kernel: INCREMENTAL inversion of a masked
autoregressive layer.

`AutoregressiveTransform._inverse` (zuko/transforms.py:994-1000) runs `passes` full conditioner evaluations; the
partial sweeps of csrc/fused_ar.hip re-evaluate, per sweep, the prefix of the conditioner the sweep's features depend
on (about D/5 forward passes in total for D = 64).  The conditioner is linear in everything that is already known,
so nothing has to be evaluated twice: keep every hidden unit's pre-activation as a running sum, and when feature f
becomes known
    1. add column f of W_1 (times x_f) into layer 1's sums,
    2. every hidden unit whose inputs are now all known is FINAL: apply the activation and add its column of the
       next layer's weights into that layer's sums, recursively,
    3. the parameters of the next feature in the order are final as well: invert it.
Each weight is used exactly once: one forward pass of multiply-adds per sample in total.  At tile granularity
(16 x 16 weight tiles over degree-sorted units, the layout of zuko_amd/fused.py) a step touches the diagonal tiles
only; everything left of the diagonal is accumulated once, in bulk, when its inputs become final.

Run:  python scripts/design/incremental_inverse.py      (asserts equality with the sweep loop and prints the counts)
"""

from __future__ import annotations

import numpy as np


def made_masks(D: int, hidden: list[int], total: int, rng: np.random.Generator):
    """Degree-based masks of a MADE with input order 0..D-1 (strictly-lower connectivity to outputs)."""
    deg_in = np.arange(D)
    masks, degs = [], [deg_in]
    for h in hidden:
        d = np.sort(rng.integers(0, max(D - 1, 1), size=h)) if False else (np.arange(h) % max(D - 1, 1))
        d = np.sort(d)
        masks.append(d[:, None] >= degs[-1][None, :])  # unit u sees inputs of degree <= deg(u)
        degs.append(d)
    out_deg = np.repeat(np.arange(D), total)
    masks.append(out_deg[:, None] > degs[-1][None, :])  # outputs of feature f see units of degree < f
    return masks, degs


def conditioner(masks, Ws, bs, x):
    h = x
    for l, (m, W, b) in enumerate(zip(masks, Ws, bs)):
        h = h @ (W * m).T + b
        if l + 1 < len(Ws):
            h = np.maximum(h, 0.0)
    return h


def inverse_by_sweeps(masks, Ws, bs, y, total, passes):
    """The reference loop with an affine univariate map: x = (y - shift) / exp(scale)."""
    x = np.zeros_like(y)
    flops = 0
    for _ in range(passes):
        phi = conditioner(masks, Ws, bs, x).reshape(y.shape[0], -1, total)
        x = (y - phi[..., 0]) * np.exp(-phi[..., 1])
        flops += sum(int(m.sum()) for m in masks)
    return x, flops


def inverse_incremental(masks, degs, Ws, bs, y, total):
    n, D = y.shape
    L = len(Ws)
    Wm = [W * m for W, m in zip(Ws, masks)]
    pre = [np.tile(b, (n, 1)) for b in bs]            # running pre-activations of every layer (start at the bias)
    final = [np.zeros(W.shape[0], dtype=bool) for W in Ws[:-1]]
    x = np.zeros_like(y)
    macs = 0
    # hidden units that see no input at all are final from the start
    def finalize(layer, units):
        nonlocal macs
        if len(units) == 0:
            return
        act = np.maximum(pre[layer][:, units], 0.0)
        pre[layer + 1] += act @ Wm[layer + 1][:, units].T
        macs += int(masks[layer + 1][:, units].sum())
        final[layer][units] = True
        if layer + 1 < L - 1:
            # units of the next layer whose every input unit is final
            ready = ~final[layer + 1] & ~(masks[layer + 1] & ~final[layer][None, :]).any(axis=1)
            # ... and that cannot receive anything from features that are still unknown
            ready &= degs[layer + 2] <= known_deg
            finalize(layer + 1, np.nonzero(ready)[0])

    known_deg = -1
    for f in range(D):
        # parameters of feature f are final: all their inputs (units of degree < f) have been folded in
        phi = pre[-1].reshape(n, D, total)[:, f]
        x[:, f] = (y[:, f] - phi[:, 0]) * np.exp(-phi[:, 1])
        known_deg = f
        pre[0] += np.outer(x[:, f], Wm[0][:, f])
        macs += int(masks[0][:, f].sum())
        ready = ~final[0] & (degs[1] <= known_deg)
        finalize(0, np.nonzero(ready)[0])
    return x, macs


def main():
    rng = np.random.default_rng(0)
    D, hidden, total, n = 64, [256, 256, 256], 2, 32
    masks, degs = made_masks(D, hidden, total, rng)
    dims = [D] + hidden + [D * total]
    Ws = [rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i]) for i in range(len(dims) - 1)]
    bs = [rng.standard_normal(dims[i + 1]) * 0.1 for i in range(len(dims) - 1)]
    y = rng.standard_normal((n, D))
    x_ref, macs_ref = inverse_by_sweeps(masks, Ws, bs, y, total, D)
    x_inc, macs_inc = inverse_incremental(masks, degs, Ws, bs, y, total)
    phi = conditioner(masks, Ws, bs, x_inc).reshape(n, D, total)
    y_back = x_inc * np.exp(phi[..., 1]) + phi[..., 0]
    assert np.allclose(x_inc, x_ref, rtol=1e-9, atol=1e-9), np.abs(x_inc - x_ref).max()
    assert np.allclose(y_back, y, rtol=1e-9, atol=1e-9)
    one_forward = sum(int(m.sum()) for m in masks)
    assert macs_inc == one_forward, (macs_inc, one_forward)
    print(f"incremental == sweeps (max diff {np.abs(x_inc - x_ref).max():.2e}); multiply-adds per sample: "
          f"{D} full sweeps {macs_ref:,}, incremental {macs_inc:,} (= one forward pass, {macs_ref / macs_inc:.0f}x fewer)")


if __name__ == "__main__":
    main()
