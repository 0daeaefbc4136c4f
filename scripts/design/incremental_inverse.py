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


# ------------------------------------------------------------------------------------------------------------------
# Tile-level version on zuko's own masks: dependency classes aligned to MFMA k-steps
# ------------------------------------------------------------------------------------------------------------------

def class_aligned_layout(mask_in: np.ndarray):
    """Hidden units of one layer grouped by dependency class (number of flow inputs they may see), every class padded
    to whole k-steps of 4 rows.  Returns (perm with -1 padding, class of every k-step)."""
    cls = mask_in.sum(axis=1)
    perm, kclass = [], []
    for c in np.unique(cls):
        units = np.nonzero(cls == c)[0].tolist()
        while len(units) % 4:
            units.append(-1)
        perm += units
        kclass += [int(c)] * (len(units) // 4)
    while len(perm) % 16:
        perm += [-1] * 4
        kclass.append(10**9)  # padding k-step: never needed
    return np.array(perm), kclass


def tile_level(D=64, H=256, layers=3, total=2, n=8, seed=1):
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import torch

    from zuko_amd.nn import masked_mlp_masks

    order = torch.arange(D)
    adjacency = (order[:, None] > order).repeat_interleave(total, dim=0)
    masks = [m.numpy() for m in masked_mlp_masks(adjacency, [H] * layers)]
    rng = np.random.default_rng(seed)
    dims = [D] + [H] * layers + [D * total]
    Ws = [rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i]) for i in range(len(dims) - 1)]
    bs = [rng.standard_normal(dims[i + 1]) * 0.1 for i in range(len(dims) - 1)]
    y = rng.standard_normal((n, D))
    x_ref, _ = inverse_by_sweeps(masks, Ws, bs, y, total, D)

    # dependency of every hidden unit on the flow inputs -> class; class-aligned row layout per hidden layer
    dep = masks[0].astype(np.int64)
    layouts = []
    for l in range(layers):
        layouts.append(class_aligned_layout(dep > 0))
        if l + 1 < layers:
            dep = (masks[l + 1].astype(np.int64) @ (dep > 0).astype(np.int64))
    # padded, masked weights in the aligned layout
    def padded(W, m, rows, cols):
        Wm = W * m
        out = np.zeros((len(rows), len(cols)))
        r_ok, c_ok = rows >= 0, cols >= 0
        out[np.ix_(r_ok, c_ok)] = Wm[np.ix_(rows[r_ok], cols[c_ok])]
        return out
    in_cols = np.arange(D)
    Wp, bp = [], []
    prev = in_cols
    for l in range(layers):
        rows = layouts[l][0]
        Wp.append(padded(Ws[l], masks[l], rows, prev))
        b = np.zeros(len(rows)); b[rows >= 0] = bs[l][rows[rows >= 0]]
        bp.append(b)
        prev = rows
    Wp.append(padded(Ws[-1], masks[-1], np.arange(D * total), prev))
    bp.append(bs[-1])

    # incremental evaluation at k-step granularity: k-step (layer l, index s) is FINAL once class(s) - 1 <= last known order
    ksteps = [len(kc) for _, kc in layouts]
    pre = [np.tile(b, (n, 1)) for b in bp]
    done = [np.zeros(k, dtype=bool) for k in ksteps]
    x = np.zeros((n, D))
    diag_mfma = bulk_mfma = 0

    def push(layer, s):
        """k-step s of hidden `layer` is final: activate its 4 rows and add their columns into the next layer (all
        rows: in the kernel the rows of the current tile are the diagonal MFMA, the rest is pulled in bulk later)."""
        nonlocal diag_mfma, bulk_mfma
        rows = slice(4 * s, 4 * s + 4)
        act = np.maximum(pre[layer][:, rows], 0.0)
        nxt = Wp[layer + 1][:, rows]
        pre[layer + 1] += act @ nxt.T
        out_tiles = int(np.ceil(np.count_nonzero(np.abs(nxt).sum(axis=1)) / 16))  # 16-row tiles that hold non-zeros of this k-step
        diag_mfma += 1
        bulk_mfma += max(out_tiles - 1, 0)
        done[layer][s] = True

    for f in range(D):
        phi = pre[-1].reshape(n, D, total)[:, f]
        x[:, f] = (y[:, f] - phi[:, 0]) * np.exp(-phi[:, 1])
        pre[0] += np.outer(x[:, f], Wp[0][:, f])
        diag_mfma += 1
        if f % 4 == 3:  # the group's k-step of inputs is complete: its bulk part into the later tiles
            bulk_mfma += int(np.ceil(np.count_nonzero(np.abs(Wp[0][:, f - 3 : f + 1]).sum(axis=1)) / 16)) - 1
        for l in range(layers):
            for s, c in enumerate(layouts[l][1]):
                if not done[l][s] and c - 1 <= f and (l == 0 or all(done[l - 1][t] for t, ct in enumerate(layouts[l - 1][1]) if ct <= c)):
                    push(l, s)
    assert np.allclose(x, x_ref, rtol=1e-9, atol=1e-9), np.abs(x - x_ref).max()
    dense_fwd = sum((Wp[i].shape[0] // 16 if i < layers else -(-Wp[i].shape[0] // 16)) * (Wp[i].shape[1] // 4) for i in range(layers + 1))
    print(f"tile level (zuko masks, D={D}, H={H}x{layers}): k-steps per hidden layer {ksteps} ({[k // 4 for k in ksteps]} tiles); "
          f"x == sweeps (max diff {np.abs(x - x_ref).max():.1e}); dependent (diagonal) MFMAs {diag_mfma}, bulk MFMAs {bulk_mfma} "
          f"(a dense forward over the padded layout: {dense_fwd})")
    return diag_mfma, bulk_mfma


if __name__ == "__main__":
    tile_level()
