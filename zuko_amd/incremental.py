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

Everything here is integer bookkeeping on the CPU, done once per module; tests/plan_emulators.py walks the same tables in numpy and is
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


def inverse_half_enabled() -> bool:
    """Whether the incremental inverse runs its pull phase on the f16 matrix instruction (HALF instantiation of csrc/inc_inverse.hip).  OFF by default:
    measured on MI355X (profiles/r06/inverse.md) it moves the launch by +2 % (NSF) / -4 % (MAF) — the launch is bound by the serial chain of its five passes
    per group, not by the pulls.  ZUKO_AMD_INVERSE_HALF=1 enables it (tests/test_gpu_flows.py runs both forms)."""
    import os

    return os.environ.get("ZUKO_AMD_INVERSE_HALF", "0") == "1"


@dataclass
class HalfStream:
    """The HALF stream of an IncPlan (csrc/inc_inverse.hip, HALF instantiation): same groups, same first-layer and diagonal tiles (f32 images), the
    PULLS as 16 x 32 blocks (out tile of group j, PAIR (2 p, 2 p + 1) of final in tiles, tile 2 p + 1 zero while it is not final) of two f16 images."""

    gather_f32: np.ndarray       # int32 [n_images * 256] into the concatenated weights, -1 = zero (block images: -1, overwritten by the block gathers)
    blk_gather: list             # per linear layer l = 1 .. NH: int32 [n_blocks_l * 512] (lane-major, 8 per lane) into the concatenated weights
    blk_pos: list                # per linear layer l: int64 [n_blocks_l] image index of every block's first image
    n_images: int
    n_chunks: int


def half_stream(plan: IncPlan, masks, chunk: int = CHUNK) -> HalfStream:
    """Rebuilds the plan's stream in the HALF layout from the same tables (`masks` only for their shapes)."""
    NH, L, nt, n_groups, din = plan.n_hidden, plan.n_hidden + 1, plan.nt, plan.n_groups, plan.din
    shapes = [tuple(np.asarray(m.detach().cpu().numpy() if hasattr(m, "detach") else m).shape) for m in masks]
    total = plan.layout.total
    lane = np.arange(64)
    li, lq = lane % 16, lane // 16
    perms, featmap, w_offsets = plan.perms, plan.featmap.astype(np.int64), plan.w_offsets

    def image(l: int, rows: np.ndarray, cols: np.ndarray) -> np.ndarray:  # [64, 4]: element (lane, r) <- W_l[rows[lane % 16], cols[4 (lane / 16) + r]]
        r = rows[li][:, None]
        c = cols[(4 * lq)[:, None] + np.arange(4)[None, :]]
        idx = w_offsets[l] + r * shapes[l][1] + c
        idx[(r < 0) | (c < 0)] = -1
        return idx

    def in_cols(it: int) -> np.ndarray:
        c = np.arange(it * TILE, (it + 1) * TILE)
        c[c >= din] = -1
        return c

    hid_rows = lambda l, j: perms[l][j * TILE : (j + 1) * TILE]
    none = -np.ones(TILE, dtype=np.int64)

    def last_rows(j: int, t: int) -> np.ndarray:
        rows = -np.ones(TILE, dtype=np.int64)
        for i in range(TILE):
            pp = 4 * t + (i & 3)
            f = featmap[j * 4 + (i >> 2)]
            if pp < total and f >= 0:
                rows[i] = f * total + pp
        return rows

    images: list = []                                      # per image: [256] indices, or None for a block image
    blk = {l: [] for l in range(1, L)}
    pos = {l: [] for l in range(1, L)}
    zero = -np.ones(256, dtype=np.int64)

    def pair_block(l: int, rows: np.ndarray, src_layer: int, p_: int, j: int) -> None:
        lo = image(l, rows, hid_rows(src_layer, 2 * p_))
        hi = image(l, rows, hid_rows(src_layer, 2 * p_ + 1) if 2 * p_ + 1 < j else none)  # (the pair's second tile joins once it is final)
        blk[l].append(np.concatenate([lo, hi], axis=1).reshape(-1))
        pos[l].append(len(images))
        images.extend([None, None])

    for j in range(n_groups):
        ns, nd = int(plan.prog[j, 0]), int(plan.prog[j, 1])
        stat = plan.prog[j, 2 : 2 + ns]
        dyn = plan.prog[j, 2 + MAX_TILES : 2 + MAX_TILES + nd]
        npr = (j + 1) // 2
        for it in stat:
            images.append(image(0, hid_rows(0, j), in_cols(int(it))).reshape(-1))
        images += [zero] * (L1S - ns)
        for l in range(1, NH):
            for p_ in range(npr):
                pair_block(l, hid_rows(l, j), l - 1, p_, j)
        for p_ in range(npr):
            for tt in range(nt):
                pair_block(L - 1, last_rows(j, tt), NH - 1, p_, j)
        for it in dyn:
            images.append(image(0, hid_rows(0, j), in_cols(int(it))).reshape(-1))
        images += [zero] * (L1D - nd)
        for l in range(1, NH):
            images.append(image(l, hid_rows(l, j), hid_rows(l - 1, j)).reshape(-1))
        for tt in range(nt):
            images.append(image(L - 1, last_rows(j, tt), hid_rows(NH - 1, j)).reshape(-1))
    n_chunks = -(-len(images) // chunk)
    images += [zero] * (n_chunks * chunk - len(images))
    g32 = np.concatenate([zero if im is None else im for im in images]).astype(np.int32)
    return HalfStream(gather_f32=g32, blk_gather=[None] + [np.concatenate(blk[l]).astype(np.int32) if blk[l] else np.zeros(0, np.int32) for l in range(1, L)],
                      blk_pos=[None] + [np.asarray(pos[l], dtype=np.int64) for l in range(1, L)], n_images=len(images), n_chunks=n_chunks)


# --------------------------------------------------------------------------------------------------
# device-side state
# --------------------------------------------------------------------------------------------------


class IncAR:
    """Runs zk_ar_inverse_incremental for one MaskedAutoregressiveTransform on one device."""

    def __init__(self, plan: IncPlan, lins, device, act: int, bound: float, slope: float, eps: float | None = None) -> None:
        import ctypes

        import torch

        self.plan, self.device, self.act, self.bound, self.slope = plan, device, act, bound, slope
        self.eps = eps  # (Bernstein map: continuation margin)
        self._gl = None  # (SOS map: ctypes arrays of the Gauss-Legendre nodes / weights, kept alive here)
        self.gather = torch.from_numpy(plan.gather).to(device)
        self.bias_gather = torch.from_numpy(plan.bias_gather).to(device)
        self.featmap = torch.from_numpy(plan.featmap.copy()).to(device)
        self.prog = torch.from_numpy(plan.prog.copy()).contiguous().to(device)
        self.mask_cat = torch.cat([l.mask.detach().reshape(-1).to(torch.uint8) for l in lins]).to(device)
        self.stream = torch.empty(plan.n_blocks * 256, dtype=torch.float32, device=device)
        self.bias = torch.empty(len(plan.bias_gather), dtype=torch.float32, device=device)
        self.bias_off = (ctypes.c_int * (plan.n_hidden + 1))(*[int(v) for v in plan.bias_off])
        self._stamp = None
        # the HALF stream (pulls on the f16 matrix instruction: csrc/inc_inverse.hip), built when the weights allow it (fused.half_scales) and
        # zuko_amd.matmul_precision() is "f16x2"
        hs = half_stream(plan, [l.mask for l in lins])
        self.half = hs
        self.h_gather = torch.from_numpy(hs.gather_f32).to(device)
        self.h_blk = [None] + [torch.from_numpy(g).to(device) for g in hs.blk_gather[1:]]
        self.h_pos = [None] + [torch.from_numpy(np.stack([p, p + 1], axis=1).reshape(-1)).to(device) for p in hs.blk_pos[1:]]  # both images of every block
        self.h_stream = torch.empty(hs.n_images * 256, dtype=torch.float32, device=device)
        self.h_ok = False
        self.h_descale = None
        self._h_stamp = None

    def refresh(self, lins) -> None:
        """(Re)build the weight stream / bias image if any parameter changed since the last call."""
        import torch

        from . import _C
        from .nn import _param_stamp
        from .ops import _ptr, _stream

        from . import fused

        stamp = _param_stamp(lins)
        want_half = fused.matmul_precision() == "f16x2" and inverse_half_enabled() and self._h_stamp != stamp  # (also when the mode was switched after the f32 stream was built)
        if stamp == self._stamp and not want_half:
            return
        lib = _C.lib()
        wcat = torch.cat([l.weight.detach().reshape(-1) for l in lins])
        if stamp != self._stamp:
            bcat = torch.cat([(l.bias.detach() if l.bias is not None else torch.zeros(l.weight.shape[0], device=self.device)).reshape(-1) for l in lins])
            _C.check(lib.zk_gather_f32(_ptr(wcat), _ptr(self.mask_cat), _ptr(self.gather), self.gather.numel(), _ptr(self.stream), _stream()), "zk_gather_f32")
            _C.check(lib.zk_gather_f32(_ptr(bcat), None, _ptr(self.bias_gather), self.bias_gather.numel(), _ptr(self.bias), _stream()), "zk_gather_f32")
            self._stamp = stamp
        if want_half:
            self._refresh_half(lins, wcat, stamp)

    def _refresh_half(self, lins, wcat, stamp) -> None:
        import torch

        from . import _C, fused
        from .ops import _ptr, _stream

        self._h_stamp, self.h_ok = stamp, False
        if fused.matmul_precision() != "f16x2" or not inverse_half_enabled():
            return
        scales = fused.half_scales(lins)  # (one synchronisation per weight version; the first layer stays on the f32 instruction whatever its weights)
        if not all(ok for ok, _ in scales[1:]):
            return
        lib = _C.lib()
        _C.check(lib.zk_gather_f32(_ptr(wcat), _ptr(self.mask_cat), _ptr(self.h_gather), self.h_gather.numel(), _ptr(self.h_stream), _stream()), "zk_gather_f32")
        images = self.h_stream.view(-1, 256)
        for l in range(1, len(lins)):
            nb = self.h_blk[l].numel() // 512
            if nb == 0:
                continue
            tmp = torch.empty(nb * 512, dtype=torch.float32, device=self.device)
            _C.check(lib.zk_gather_split_f16(_ptr(wcat), _ptr(self.mask_cat), _ptr(self.h_blk[l]), nb, _ptr(tmp), 2.0 ** scales[l][1], _stream()), "zk_gather_split_f16")
            images.index_copy_(0, self.h_pos[l], tmp.view(-1, 256))
        self.h_descale = [1.0] + [2.0 ** -e for _, e in scales[1:]] + [1.0] * (4 - len(scales))
        self.h_ok = True

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
        from . import fused

        half = self.h_ok and self._h_stamp == self._stamp and fused.matmul_precision() == "f16x2" and inverse_half_enabled() and p.layout.kind <= 3
        extra = dict(half=1, wdescale1=self.h_descale[1], wdescale2=self.h_descale[2], wdescale3=self.h_descale[3]) if half else {}
        if p.layout.kind in (5, 6):  # the polynomial maps: bisection inverse in the launch (zuko/transforms.py:608-617: n = ceil(log2(2 B / 1e-6)) steps)
            import ctypes
            import math

            eps = 1e-6 if self.eps is None else float(self.eps)
            extra["n_bisect"] = math.ceil(math.log2(2 * self.bound / (1e-6 if p.layout.kind == 5 else eps)))
            if p.layout.kind == 5:
                if self._gl is None:
                    from .ops import _leggauss01

                    self._gl = _leggauss01(5)
                extra.update(gl_nodes01=ctypes.cast(self._gl[0], ctypes.c_void_p), gl_weights01=ctypes.cast(self._gl[1], ctypes.c_void_p))
            else:
                extra["eps"] = eps
        a = _C.args("zk_ar_inc_args_v1", uni_kind=p.layout.kind, n_hidden=p.n_hidden, N=N, D=p.features, C=C, y=_ptr(y), ldy=y.stride(0), ctx=_ptr(ctx),
                    ldc=0 if ctx is None else ctx.stride(0), x=_ptr(x), ldx=p.features, ladj=_ptr(ladj), wstream=_ptr(self.h_stream if half else self.stream), bias=_ptr(self.bias),
                    bias_floats=self.bias.numel(), bias_off=self.bias_off, featmap=_ptr(self.featmap), prog=_ptr(self.prog), n_groups=p.n_groups,
                    n_chunks=self.half.n_chunks if half else p.n_chunks, act=self.act, bound=self.bound, slope=self.slope, **extra)
        err = _C.lib().zk_ar_inverse_incremental(a, _stream())
        _C.check(err, "zk_ar_inverse_incremental")
        return x, ladj
