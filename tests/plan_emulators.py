r"""Numpy walks through the SAME weight streams / tables the fused kernels read (tile order, skip bits, chunk padding, feature
regrouping) — test infrastructure: the CPU tests use them to validate a plan without a GPU.  Nothing in zuko_amd/ imports this file."""
from __future__ import annotations

import numpy as np

from zuko_amd.fused import GROUP_HIDDEN, TILE, ArPlan  # (TILE = 16 in all three planners)
from zuko_amd.coupling_plan import CHUNK as CP_CHUNK, MAX_T, CouplingPlan
from zuko_amd.incremental import L1D, L1S, MAX_TILES, HalfStream, IncPlan


def simulate_ar(plan: ArPlan, weights: list[np.ndarray], biases: list[np.ndarray], masks: list[np.ndarray], inp: np.ndarray, act) -> np.ndarray:
    """Pure-numpy walk through the SAME stream / tables the kernel uses (tile order, skip bits,
    chunk padding, feature regrouping) for a [n, din] input; returns phi[n, features, total].
    Used by the CPU tests to validate the plan without a GPU."""
    stream = []
    for l, g in enumerate(plan.gather):
        w = (weights[l] * masks[l]).reshape(-1)
        s = np.where(g >= 0, w[np.maximum(g, 0)], 0.0)
        stream.append(s)
    stream = np.concatenate(stream).reshape(-1, 64, 4)  # [block][lane][r]
    bias_img = []
    for l, g in enumerate(plan.bias_gather):
        bias_img.append(np.where(g >= 0, biases[l][np.maximum(g, 0)], 0.0))
    n = inp.shape[0]
    MW = plan.max_width
    n_otg, n_itile = MW // TILE // GROUP_HIDDEN, MW // TILE
    cur = np.zeros((n, MW))
    cur[:, : plan.din] = inp
    lay = plan.layout
    for l in range(plan.n_layers - 1):
        b = plan.layer_block0[l]
        out = np.zeros((n, MW))
        for otg in range(n_otg):
            bits = int(plan.skip[l * n_otg + otg])
            for it in range(n_itile):
                if bits >> it & 1:
                    for t in range(GROUP_HIDDEN):
                        ot = otg * GROUP_HIDDEN + t
                        blk = stream[b]
                        b += 1
                        # lane (i, q), r: A[i][k = 4q + r]
                        A = blk.reshape(4, 16, 4).transpose(1, 0, 2).reshape(16, 16)  # [i][4q+r]
                        out[:, ot * 16 : ot * 16 + 16] += cur[:, it * 16 : it * 16 + 16] @ A.T
        cur = act(out + bias_img[l][None, :])
    b = plan.layer_block0[-1]
    phi = np.zeros((n, plan.features, lay.total))
    per_group = 4 * lay.fpl
    for g in range(plan.n_groups):
        acc = np.zeros((n, lay.nt, 16))
        bits = int(plan.skip[(plan.n_layers - 1) * n_otg + g])
        for it in range(n_itile):
            if bits >> it & 1:
                for t in range(lay.nt):
                    blk = stream[b]
                    b += 1
                    A = blk.reshape(4, 16, 4).transpose(1, 0, 2).reshape(16, 16)
                    acc[:, t, :] += cur[:, it * 16 : it * 16 + 16] @ A.T
        acc += bias_img[-1].reshape(plan.n_groups, lay.nt, 16)[g][None]
        for t in range(lay.nt):
            for i in range(16):
                m = 4 * t + (i & 3)
                fi, p = divmod(m, lay.total)
                if fi < lay.fpl:
                    f = plan.featmap[g * per_group + (i >> 2) * lay.fpl + fi]
                    if f >= 0:
                        phi[:, f, p] = acc[:, t, i]
    return phi


def simulate_coupling(plan: CouplingPlan, weights, biases, x: np.ndarray, ctx: np.ndarray | None, act, ls: float):
    """Numpy walk through the kernel's tables: returns (y [n, D], ladj [n])."""
    n = x.shape[0]
    wcat = np.concatenate([np.asarray(w).reshape(-1) for w in weights])
    bcat = np.concatenate([np.asarray(b).reshape(-1) for b in biases])
    stream = np.where(plan.gather >= 0, wcat[np.maximum(plan.gather, 0)], 0.0).reshape(-1, 64, 4)
    bias = np.where(plan.bias_gather >= 0, bcat[np.maximum(plan.bias_gather, 0)], 0.0)

    def tile_mat(blk):
        return blk.reshape(4, 16, 4).transpose(1, 0, 2).reshape(16, 16)

    cur = np.zeros((n, plan.nit * TILE))
    for i, src in enumerate(plan.amap):
        if src >= 0:
            cur[:, i] = x[:, src]
        elif src <= -2:
            cur[:, i] = ctx[:, -2 - src]
    pos = 0
    L = plan.n_layers
    for l in range(L - 1):
        n_in = plan.nit if l == 0 else plan.tiles[l - 1]
        out = np.zeros((n, MAX_T * TILE))
        for otg in range(-(-plan.tiles[l] // 4)):
            for t in range(4):
                out[:, (otg * 4 + t) * TILE : (otg * 4 + t + 1) * TILE] = bias[plan.bias_off[l] + (otg * 4 + t) * TILE : plan.bias_off[l] + (otg * 4 + t + 1) * TILE]
            for it in range(n_in):
                for t in range(4):
                    out[:, (otg * 4 + t) * TILE : (otg * 4 + t + 1) * TILE] += cur[:, it * TILE : (it + 1) * TILE] @ tile_mat(stream[pos]).T
                    pos += 1
        pos = -(-pos // CP_CHUNK) * CP_CHUNK
        cur = act(out)
        cur[:, plan.widths[l] :] = 0.0
    y = x.copy()
    ladj = np.zeros(n)
    for g in range(plan.n_groups):
        acc = np.zeros((n, TILE)) + bias[plan.bias_off[L - 1] + g * TILE : plan.bias_off[L - 1] + (g + 1) * TILE]
        for it in range(plan.tiles[-1]):
            acc += cur[:, it * TILE : (it + 1) * TILE] @ tile_mat(stream[pos]).T
            pos += 1
        for qq in range(4):
            for fi in range(2):
                f = plan.fmap[g * 8 + qq * 2 + fi]
                if f >= 0:
                    shift, scale = acc[:, 4 * qq + 2 * fi], acc[:, 4 * qq + 2 * fi + 1]
                    lsc = scale / (1 + np.abs(scale / ls))
                    y[:, f] = x[:, f] * np.exp(lsc) + shift
                    ladj += lsc
    return y, ladj


def _f16_parts(v: np.ndarray):
    h = v.astype(np.float32).astype(np.float16).astype(np.float32)
    return h, (v.astype(np.float32) - h).astype(np.float16).astype(np.float32)


def _pow2(amax) -> np.ndarray:
    """2^ea with amax 2^ea in [2^14, 2^15) (ea clamped to [-90, 90]; amax = 0: ea = 15) — inc_pair_convert of the kernel."""
    m, e = np.frexp(np.asarray(amax, dtype=np.float64))
    return np.exp2(np.clip(15 - np.where(np.asarray(amax) > 0, e, 0), -90, 90).astype(np.float64))


def simulate_inc(plan: IncPlan, weights, biases, masks, y: np.ndarray, ctx: np.ndarray | None, act, inv_fn, half: HalfStream | None = None, wexp=None):
    """Numpy walk through the SAME stream / tables the kernel uses.  y [n, features] values to invert, ctx [n, context] or None;
    `inv_fn(phi[n, total], yv[n]) -> (x[n], ladj[n])`.  Returns (x [n, features], ladj [n])."""
    n = y.shape[0]
    NH, L = plan.n_hidden, plan.n_hidden + 1
    wcat = np.concatenate([(np.asarray(w) * np.asarray(m)).reshape(-1) for w, m in zip(weights, masks)])
    bcat = np.concatenate([np.asarray(b).reshape(-1) for b in biases])
    bias = np.where(plan.bias_gather >= 0, bcat[np.maximum(plan.bias_gather, 0)], 0.0)
    if half is not None:  # the HALF stream: f32 images + pair blocks in two f16 parts of W 2^wexp[l]; walked below exactly as the kernel walks it
        stream = np.where(half.gather_f32 >= 0, wcat[np.maximum(half.gather_f32, 0)], 0.0).reshape(-1, 64, 4)
        blocks = {}
        for l in range(1, L):
            idx = half.blk_gather[l].reshape(-1, 64, 8)
            vals = np.where(idx >= 0, wcat[np.maximum(idx, 0)], 0.0) * 2.0 ** wexp[l]
            bh, bl = _f16_parts(vals)
            for k, ps in enumerate(half.blk_pos[l]):
                blocks[int(ps)] = (bh[k], bl[k], 2.0 ** -wexp[l])
        pairs = [[None] * ((MAX_TILES + 1) // 2) for _ in range(NH)]  # per layer and pair: (h [n, 64 lanes.. as [n, 32]], l, 1 / s)

        def block_mat(img):  # [64 lanes, 8] -> A[i][k]: k = kq-th group of 4 of tile 2 p (first 16) / of tile 2 p + 1 (last 16)
            a4 = img.reshape(4, 16, 8)  # [kq, i, e]
            lo = a4[:, :, :4].transpose(1, 0, 2).reshape(16, 16)
            hi = a4[:, :, 4:].transpose(1, 0, 2).reshape(16, 16)
            return np.concatenate([lo, hi], axis=1)  # [16, 32]

        def pull(pos_, layer_src, p_):
            bh, bl, wd = blocks[pos_]
            ph, pl, inv_s = pairs[layer_src][p_]
            A_h, A_l = block_mat(bh), block_mat(bl)
            t = ph @ A_l.T + pl @ A_h.T + ph @ A_h.T  # (smallest first; f32 accumulation in the kernel)
            return t * (inv_s * wd)[:, None]

        def finalize(layer, j_, hj):
            lo = hj if j_ % 2 == 0 else h[layer][:, (j_ - 1) * TILE : j_ * TILE]
            hi = np.zeros_like(hj) if j_ % 2 == 0 else hj
            both = np.concatenate([lo, hi], axis=1)
            s_ = _pow2(np.abs(both).max(axis=1))
            ph, pl = _f16_parts(both * s_[:, None])
            pairs[layer][j_ // 2] = (ph, pl, 1.0 / s_)
    else:
        stream = np.where(plan.gather >= 0, wcat[np.maximum(plan.gather, 0)], 0.0).reshape(-1, 64, 4)

    def tile_mat(blk):  # [64 lanes, 4] -> A[i][k = 4q + r]
        return blk.reshape(4, 16, 4).transpose(1, 0, 2).reshape(16, 16)

    xin = np.zeros((n, plan.nit * TILE))
    if ctx is not None:
        xin[:, plan.features : plan.features + ctx.shape[1]] = ctx
    h = [np.zeros((n, MAX_TILES * TILE)) for _ in range(NH)]
    ladj = np.zeros(n)
    pos = 0
    total = plan.layout.total
    for j in range(plan.n_groups):
        ns, nd = int(plan.prog[j, 0]), int(plan.prog[j, 1])
        stat = plan.prog[j, 2 : 2 + ns]
        dyn = plan.prog[j, 2 + MAX_TILES : 2 + MAX_TILES + nd]
        off = [bias[plan.bias_off[l] + j * TILE : plan.bias_off[l] + (j + 1) * TILE][None, :].repeat(n, 0).copy() for l in range(NH)]
        for i, it in enumerate(stat):
            off[0] += xin[:, it * TILE : (it + 1) * TILE] @ tile_mat(stream[pos + i]).T
        pos += L1S
        poff = np.zeros((n, plan.nt, TILE))
        for tt in range(plan.nt):
            b0 = plan.bias_off[NH] + (j * plan.nt + tt) * TILE
            poff[:, tt, :] = bias[b0 : b0 + TILE][None, :]
        if half is not None:
            npr = (j + 1) // 2
            for l in range(1, NH):
                for p_ in range(npr):
                    off[l] += pull(pos, l - 1, p_)
                    pos += 2
            for p_ in range(npr):
                for tt in range(plan.nt):
                    poff[:, tt, :] += pull(pos, NH - 1, p_)
                    pos += 2
        else:
            for l in range(1, NH):
                for t in range(j):
                    off[l] += h[l - 1][:, t * TILE : (t + 1) * TILE] @ tile_mat(stream[pos]).T
                    pos += 1
            for t in range(j):
                for tt in range(plan.nt):
                    poff[:, tt, :] += h[NH - 1][:, t * TILE : (t + 1) * TILE] @ tile_mat(stream[pos]).T
                    pos += 1
        wd = [tile_mat(stream[pos + i]) for i in range(nd)]
        pos += L1D
        wh = [tile_mat(stream[pos + i]) for i in range(NH - 1)]
        pos += NH - 1
        wl = [tile_mat(stream[pos + i]) for i in range(plan.nt)]
        pos += plan.nt
        for r in range(5):
            cur = off[0].copy()
            for i, it in enumerate(dyn):
                cur += xin[:, it * TILE : (it + 1) * TILE] @ wd[i].T
            hj = [act(cur)]
            for l in range(1, NH):
                hj.append(act(off[l] + hj[l - 1] @ wh[l - 1].T))
            if r == 4:
                for l in range(NH):
                    h[l][:, j * TILE : (j + 1) * TILE] = hj[l]
                    if half is not None:
                        finalize(l, j, hj[l])
                break
            p = poff.copy()
            for tt in range(plan.nt):
                p[:, tt, :] += hj[NH - 1] @ wl[tt].T
            f = int(plan.featmap[j * 4 + r])
            if f >= 0:
                phi = np.stack([p[:, pp // 4, 4 * r + (pp & 3)] for pp in range(total)], axis=1)  # lane q = r: rows 4 r .. 4 r + 3 of every tile
                xv, lj = inv_fn(phi, y[:, f])
                xin[:, f] = xv
                ladj += lj
    return xin[:, : plan.features], ladj
