r"""Host-side plan of the INCREMENTAL autoregressive inverse (csrc/inc_inverse.hip).

`AutoregressiveTransform._inverse` (zuko/transforms.py:994-1000) runs `passes` sweeps x <- meta(x).inv(y): after sweep p
the features of order <= p are final.  Sweep p only needs the part of the conditioner those features depend on, and —
the observation this plan is built on — most of that part was already computed, from final inputs, by earlier sweeps.

Features are taken in groups of four consecutive slots (slot = position in the order-sorted feature list) and the hidden
units of every layer are laid out in ALIGNED tiles: tile j holds units whose newest dependency lies in group j (a unit
whose newest dependency is the LAST slot of group j may also spill into tile j + 1 — it is first needed there).  Then

  * tile j of every hidden layer is final once group j is done, and group j never needs a tile > j;
  * per group the kernel PULLS the contributions of the final tiles t < j into the pre-activations of tile j and into the
    parameters of group j exactly once (every off-diagonal 16x16 weight tile is multiplied once per sample, as in the
    density pass), keeps the nine or so DIAGONAL weight tiles in registers, and iterates only those: four passes, each
    finishing one more feature of the group (inverse univariate map in the epilogue, x fed back through a wave-private LDS
    tile), plus one pass that finalises the hidden tile.

Per sample this is ~1.5x the multiply-adds of ONE density evaluation instead of ~14x (the partial sweeps) or 64x (the
reference loop).  Layouts that do not fit (a tile would need more than 16 units, more than 16 tiles per layer, residual
blocks, D > 128) return None and the caller falls back to the partial sweeps.

Everything here is integer bookkeeping on the CPU, done once per module; `simulate` walks the same tables in numpy and is
what the CPU tests check against the oracle.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .fused import UniLayout, _deps

TILE = 16
CHUNK = 24  # tiles per LDS-ring chunk (AR_CH of the kernels)
MAX_TILES = 17  # tiles per hidden layer = feature groups (static unroll depth of the kernel)
MAX_FEATURES = 68
L1S = 4  # first-layer stream slots per group for input tiles that are final (IN_L1S of the kernel)
L1D = 4  # ... and for the diagonal (still changing) input tiles (IN_L1D)


@dataclass
class IncPlan:
    layout: UniLayout            # kind / total / nt of the univariate map (one feature per lane)
    n_hidden: int                # hidden layers (1..3)
    features: int
    din: int                     # conditioner inputs (features + context)
    nit: int                     # 16-wide input tiles
    n_groups: int
    nt: int                      # last-layer tiles per group
    featmap: np.ndarray          # int32 [n_groups * 4]: feature index of every slot, -1 = padding
    perms: list                  # per hidden layer: int64 [MAX_TILES * 16] unit index or -1
    n_tiles: list                # per hidden layer: tiles in use
    gather: np.ndarray           # int32 [n_blocks * 256]: index into the concatenation of the layers' weights (-1 -> 0)
    n_blocks: int
    n_chunks: int
    bias_gather: np.ndarray      # int32: index into the concatenation of the layers' biases (-1 -> 0)
    bias_off: list               # offset of every layer's bias image
    prog: np.ndarray             # int32 [n_groups, 2 + 2 * MAX_TILES]: (n_static, n_dynamic, static in-tile ids..., dynamic ids...)
    w_offsets: list = field(default_factory=list)   # element offset of every layer inside the concatenated weights
    b_offsets: list = field(default_factory=list)
    mults: int = 0               # 16x16 tile multiplications per 16 samples (reporting)


def build_inc_plan(masks, features: int, order: np.ndarray, layout: UniLayout, chunk: int = CHUNK):
    """masks[l]: bool [out_l, in_l] (torch or numpy) of the linear layers; `order`: the transform's feature order buffer.
    Returns an IncPlan or None when the network does not fit the aligned layout.

    The first group may hold fewer than four slots: zuko's degree assignment (zuko/nn.py:289-291) gives the lowest degrees
    one unit more than the others (e.g. 5, 5, 5, 5, 4, 4, ... for 256 units over 63 degrees), which overfills an aligned
    tile 0; shifting the group boundaries by one slot (groups {0,1,2}, {3..6}, ...) makes the 16-unit tiles fit."""
    M = [np.asarray(m.detach().cpu().numpy() if hasattr(m, "detach") else m).astype(bool) for m in masks]
    for first in (4, 3, 2, 1):
        plan = _build(M, features, order, layout, chunk, first)
        if plan is not None:
            return plan
    return None


def _build(M, features: int, order: np.ndarray, layout: UniLayout, chunk: int, first: int):
    L = len(M)
    NH = L - 1
    if NH < 1 or NH > 3 or features > MAX_FEATURES or features < 2:
        return None
    total = layout.total
    if M[-1].shape[0] != features * total:
        return None
    din = M[0].shape[1]
    if din > 256 or any(m.shape[0] > MAX_TILES * TILE for m in M[:-1]):
        return None
    nit = -(-din // TILE)
    nt = -(-total // 4)
    order = np.asarray(order).astype(np.int64)
    # slots: features sorted by order (stable), padded to whole groups of four
    slots = np.argsort(order, kind="stable")
    first = min(first, features)
    n_groups = 1 + -(-(features - first) // 4)
    if n_groups > MAX_TILES:
        return None
    slot_of = np.empty(features, dtype=np.int64)
    slot_of[slots] = np.arange(features)
    # group of every slot, and the lane (0..3) it occupies inside its group
    grp = lambda s_: np.where(np.asarray(s_) < first, 0, 1 + (np.asarray(s_) - first) // 4)
    lane_in = lambda s_: np.where(np.asarray(s_) < first, np.asarray(s_), (np.asarray(s_) - first) % 4)
    featmap = -np.ones(n_groups * 4, dtype=np.int64)
    for s_ in range(features):
        featmap[int(grp(s_)) * 4 + int(lane_in(s_))] = slots[s_]

    deps = _deps(M)  # per layer: [units, din] boolean dependency sets
    # rank(u) = newest slot a unit depends on (-1: none / context only)
    def rank_of(dep_rows: np.ndarray) -> np.ndarray:
        r = -np.ones(dep_rows.shape[0], dtype=np.int64)
        f = dep_rows[:, :features]
        for u in range(dep_rows.shape[0]):
            idx = np.nonzero(f[u])[0]
            if idx.size:
                r[u] = slot_of[idx].max()
        return r

    # the autoregressive property the scheme relies on: the parameters of slot s depend on slots < s only
    out_rank = rank_of(deps[-1][::total])
    for ftr in range(features):
        if out_rank[ftr] >= slot_of[ftr]:
            return None

    # ---- aligned tiles of the hidden layers ----------------------------------------------------------------------
    # unit u may sit in tile lo[u] .. hi[u]: not before the group of its newest dependency (nor before a unit it reads),
    # not after the group of the first slot that can read it (slot rank + 1).
    perms, n_tiles = [], []
    prev_tile = None
    for l in range(NH):
        rk = rank_of(deps[l])
        n_u = M[l].shape[0]
        lo = grp(np.maximum(rk, 0))
        hi = np.minimum(grp(np.minimum(rk + 1, features - 1)), n_groups - 1)
        if prev_tile is not None:
            for u in range(n_u):
                srcs = np.nonzero(M[l][u])[0]
                if srcs.size:
                    lo[u] = max(lo[u], int(prev_tile[srcs].max()))
        if (lo > hi).any() or lo.max(initial=0) >= MAX_TILES:
            return None
        tiles = [[] for _ in range(MAX_TILES)]
        placed = np.zeros(n_u, dtype=bool)
        byrank = np.argsort(rk, kind="stable")
        for j in range(MAX_TILES):
            must = [u for u in byrank if not placed[u] and hi[u] == j]
            if len(must) > TILE:
                return None
            tiles[j] = list(must)
            placed[must] = True
            for u in byrank:
                if len(tiles[j]) == TILE:
                    break
                if not placed[u] and lo[u] <= j < hi[u]:
                    tiles[j].append(u)
                    placed[u] = True
        if not placed.all():
            return None
        tile_of = -np.ones(n_u, dtype=np.int64)
        perm = -np.ones(MAX_TILES * TILE, dtype=np.int64)
        for j in range(MAX_TILES):
            for i, u in enumerate(tiles[j]):
                perm[j * TILE + i] = u
                tile_of[u] = j
        perms.append(perm)
        n_tiles.append(max((j for j in range(MAX_TILES) if tiles[j]), default=0) + 1)
        prev_tile = tile_of
    perms_prev_tile = prev_tile
    # the parameters of group j may only read last-hidden-layer tiles <= j
    for ftr in range(features):
        gj = int(grp(slot_of[ftr]))
        srcs = np.nonzero(M[-1][ftr * total : (ftr + 1) * total].any(axis=0))[0]
        if srcs.size and perms_prev_tile[srcs].max() > gj:
            return None
    # hidden tile t of layer l may only read input features of groups <= t (checked through the masks: zero weights)
    # ---- first layer: static / dynamic input tiles per group -------------------------------------------------------
    feat_slot = -np.ones(nit * TILE, dtype=np.int64)  # slot of every input column, -1 = context / padding (always final)
    feat_slot[:features] = slot_of
    prog = np.zeros((n_groups, 2 + 2 * MAX_TILES), dtype=np.int32)
    l1_lists = []
    for j in range(n_groups):
        rows = perms[0][j * TILE : (j + 1) * TILE]
        rows = rows[rows >= 0]
        stat, dyn = [], []
        for it in range(nit):
            cols = np.arange(it * TILE, min((it + 1) * TILE, din))
            if rows.size == 0 or not M[0][np.ix_(rows, cols)].any():
                continue
            used_cols = cols[M[0][np.ix_(rows, cols)].any(axis=0)]
            first_slot = 0 if j == 0 else first + 4 * (j - 1)
            pending = feat_slot[used_cols] >= first_slot  # inputs that are not final when group j starts
            (dyn if pending.any() else stat).append(it)
        if len(stat) > L1S or len(dyn) > L1D:
            return None
        prog[j, 0], prog[j, 1] = len(stat), len(dyn)
        prog[j, 2 : 2 + len(stat)] = stat
        prog[j, 2 + MAX_TILES : 2 + MAX_TILES + len(dyn)] = dyn
        l1_lists.append((stat, dyn))

    # ---- weight stream in consumption order ------------------------------------------------------------------------
    w_offsets, b_offsets = [], []
    acc = 0
    for m in M:
        w_offsets.append(acc)
        acc += m.size
    acc = 0
    for m in M:
        b_offsets.append(acc)
        acc += m.shape[0]
    lane = np.arange(64)
    li, lq = lane % 16, lane // 16

    def image(l: int, rows: np.ndarray, cols: np.ndarray) -> np.ndarray:
        """1 KiB tile image: element (lane, r) <- W_l[rows[lane % 16], cols[4 (lane / 16) + r]] (-1 = zero)."""
        r = rows[li][:, None]
        c = cols[(4 * lq)[:, None] + np.arange(4)[None, :]]
        idx = w_offsets[l] + r * M[l].shape[1] + c
        idx[(r < 0) | (c < 0)] = -1
        return idx.reshape(-1)

    def in_cols(it: int) -> np.ndarray:
        c = np.arange(it * TILE, (it + 1) * TILE)
        c[c >= din] = -1
        return c

    def hid_rows(l: int, j: int) -> np.ndarray:
        return perms[l][j * TILE : (j + 1) * TILE]

    def last_rows(j: int, t: int) -> np.ndarray:
        rows = -np.ones(TILE, dtype=np.int64)
        for i in range(TILE):
            p = 4 * t + (i & 3)
            f = featmap[j * 4 + (i >> 2)]
            if p < total and f >= 0:
                rows[i] = f * total + p
        return rows

    blocks, mults = [], 0
    zero_block = -np.ones(256, dtype=np.int64)
    for j in range(n_groups):
        stat, dyn = l1_lists[j]
        for it in stat:                                   # first layer, inputs that are already final
            blocks.append(image(0, hid_rows(0, j), in_cols(it)))
        blocks += [zero_block] * (L1S - len(stat))        # (fixed slot count: chunk boundaries are static in the kernel)
        for l in range(1, NH):                            # hidden layers: final tiles t < j
            for t in range(j):
                blocks.append(image(l, hid_rows(l, j), hid_rows(l - 1, t)))
        for t in range(j):                                # last layer: final tiles t < j
            for tt in range(nt):
                blocks.append(image(L - 1, last_rows(j, tt), hid_rows(NH - 1, t)))
        for it in dyn:                                    # the diagonal tiles, kept in registers over the five passes
            blocks.append(image(0, hid_rows(0, j), in_cols(it)))
        blocks += [zero_block] * (L1D - len(dyn))
        for l in range(1, NH):
            blocks.append(image(l, hid_rows(l, j), hid_rows(l - 1, j)))
        for tt in range(nt):
            blocks.append(image(L - 1, last_rows(j, tt), hid_rows(NH - 1, j)))
        mults += len(stat) + (NH - 1) * j + nt * j + 5 * (len(dyn) + (NH - 1)) + 4 * nt
    pad = -(-len(blocks) // chunk) * chunk - len(blocks)
    blocks += [-np.ones(256, dtype=np.int64)] * pad
    gather = np.concatenate(blocks).astype(np.int32)

    # bias image: hidden layers [MAX_TILES * 16] each, then the last layer [n_groups * nt * 16]
    bias_gather, bias_off = [], []
    cur = 0
    for l in range(NH):
        b = np.where(perms[l] >= 0, b_offsets[l] + np.maximum(perms[l], 0), -1)
        bias_gather.append(b)
        bias_off.append(cur)
        cur += len(b)
    lastb = []
    for j in range(n_groups):
        for tt in range(nt):
            rows = last_rows(j, tt)
            lastb.append(np.where(rows >= 0, b_offsets[L - 1] + np.maximum(rows, 0), -1))
    bias_off.append(cur)
    bias_gather.append(np.concatenate(lastb))
    return IncPlan(
        layout=layout, n_hidden=NH, features=features, din=din, nit=nit, n_groups=n_groups, nt=nt, featmap=featmap.astype(np.int32), perms=perms,
        n_tiles=n_tiles, gather=gather, n_blocks=len(blocks), n_chunks=len(blocks) // chunk, bias_gather=np.concatenate(bias_gather).astype(np.int32),
        bias_off=bias_off, prog=prog, w_offsets=w_offsets, b_offsets=b_offsets, mults=mults,
    )


def simulate(plan: IncPlan, weights, biases, masks, y: np.ndarray, ctx: np.ndarray | None, act, inv_fn):
    """Numpy walk through the SAME stream / tables the kernel uses.  y [n, features] values to invert, ctx [n, context] or None;
    `inv_fn(phi[n, total], yv[n]) -> (x[n], ladj[n])`.  Returns (x [n, features], ladj [n])."""
    n = y.shape[0]
    NH, L = plan.n_hidden, plan.n_hidden + 1
    wcat = np.concatenate([(np.asarray(w) * np.asarray(m)).reshape(-1) for w, m in zip(weights, masks)])
    bcat = np.concatenate([np.asarray(b).reshape(-1) for b in biases])
    stream = np.where(plan.gather >= 0, wcat[np.maximum(plan.gather, 0)], 0.0).reshape(-1, 64, 4)
    bias = np.where(plan.bias_gather >= 0, bcat[np.maximum(plan.bias_gather, 0)], 0.0)

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
        for l in range(1, NH):
            for t in range(j):
                off[l] += h[l - 1][:, t * TILE : (t + 1) * TILE] @ tile_mat(stream[pos]).T
                pos += 1
        poff = np.zeros((n, plan.nt, TILE))
        for tt in range(plan.nt):
            b0 = plan.bias_off[NH] + (j * plan.nt + tt) * TILE
            poff[:, tt, :] = bias[b0 : b0 + TILE][None, :]
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


# --------------------------------------------------------------------------------------------------
# device-side state
# --------------------------------------------------------------------------------------------------


class IncAR:
    """Runs zk_ar_inverse_incremental for one MaskedAutoregressiveTransform on one device."""

    def __init__(self, plan: IncPlan, lins, device, act: int, bound: float, slope: float) -> None:
        import ctypes

        import torch

        self.plan, self.device, self.act, self.bound, self.slope = plan, device, act, bound, slope
        self.gather = torch.from_numpy(plan.gather).to(device)
        self.bias_gather = torch.from_numpy(plan.bias_gather).to(device)
        self.featmap = torch.from_numpy(plan.featmap.copy()).to(device)
        self.prog = torch.from_numpy(plan.prog.copy()).contiguous().to(device)
        self.mask_cat = torch.cat([l.mask.detach().reshape(-1).to(torch.uint8) for l in lins]).to(device)
        self.stream = torch.empty(plan.n_blocks * 256, dtype=torch.float32, device=device)
        self.bias = torch.empty(len(plan.bias_gather), dtype=torch.float32, device=device)
        self.bias_off = (ctypes.c_int * (plan.n_hidden + 1))(*[int(v) for v in plan.bias_off])
        self._stamp = None

    def refresh(self, lins) -> None:
        """(Re)build the weight stream / bias image if any parameter changed since the last call."""
        import torch

        from . import _C
        from .nn import _param_stamp
        from .ops import _ptr, _stream

        stamp = _param_stamp(lins)
        if stamp == self._stamp:
            return
        lib = _C.lib()
        wcat = torch.cat([l.weight.detach().reshape(-1) for l in lins])
        bcat = torch.cat([(l.bias.detach() if l.bias is not None else torch.zeros(l.weight.shape[0], device=self.device)).reshape(-1) for l in lins])
        _C.check(lib.zk_gather_f32(_ptr(wcat), _ptr(self.mask_cat), _ptr(self.gather), self.gather.numel(), _ptr(self.stream), _stream()), "zk_gather_f32")
        _C.check(lib.zk_gather_f32(_ptr(bcat), None, _ptr(self.bias_gather), self.bias_gather.numel(), _ptr(self.bias), _stream()), "zk_gather_f32")
        self._stamp = stamp

    def run(self, y, ctx, want_ladj: bool = False):
        """y [N, D] contiguous fp32 (values to invert), ctx [N, C] or None -> (x [N, D], ladj [N] or None)."""
        import torch

        from . import _C
        from .ops import _ptr, _stream

        p = self.plan
        N = y.shape[0]
        x = torch.empty((N, p.features), dtype=torch.float32, device=y.device)
        ladj = torch.empty(N, dtype=torch.float32, device=y.device) if want_ladj else None
        C = 0 if ctx is None else ctx.shape[1]
        a = _C.args("zk_ar_inc_args_v1", uni_kind=p.layout.kind, n_hidden=p.n_hidden, N=N, D=p.features, C=C, y=_ptr(y), ldy=y.stride(0), ctx=_ptr(ctx),
                    ldc=0 if ctx is None else ctx.stride(0), x=_ptr(x), ldx=p.features, ladj=_ptr(ladj), wstream=_ptr(self.stream), bias=_ptr(self.bias),
                    bias_floats=self.bias.numel(), bias_off=self.bias_off, featmap=_ptr(self.featmap), prog=_ptr(self.prog), n_groups=p.n_groups, n_chunks=p.n_chunks,
                    act=self.act, bound=self.bound, slope=self.slope)
        err = _C.lib().zk_ar_inverse_incremental(a, _stream())
        _C.check(err, "zk_ar_inverse_incremental")
        return x, ladj
