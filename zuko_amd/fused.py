r"""Host-side planning for the fused autoregressive kernel (csrc/fused_ar.hip).

The kernel evaluates, for a tile of samples, the whole masked conditioner
(zuko/nn.py:217-218, :279-315) followed by the univariate transform + log|det J|
(zuko/transforms.py:554-567 / :436-446) with the activations held in MFMA accumulator
registers from the first layer to the spline epilogue.  This module turns a `MaskedMLP` into what
that kernel consumes:

* a permutation of every hidden layer's units by the size of their dependency set (the "degree"
  of MADE), applied to the rows of W_l and the columns of W_{l+1}: results are unchanged up to
  summation order, but the masks become block lower-triangular so that whole 16x16 weight tiles
  vanish and are skipped;
* the WEIGHT STREAM: the surviving 16x16 tiles, each stored as the 1 KiB image the 64 lanes of a
  wavefront read back as their `v_mfma_f32_16x16x4_f32` A operand (lane (i, q) <- W[row i][col 4q..4q+3]),
  in exactly the order the kernel consumes them, every layer padded to a whole number of
  24-tile chunks (the granularity of the LDS ring the tiles are DMA'd through);
* per tile-group skip bitmasks, the bias image, and the feature map of the last layer, whose
  output rows are regrouped so that ONE lane ends up holding all `total` parameters of a feature.

Everything here is integer bookkeeping on the CPU (done once per module); the per-call work —
gathering mask*W into the stream — is a HIP kernel (zk_gather_f32).
"""

from __future__ import annotations

from dataclasses import dataclass

import ctypes
import math
import os

import numpy as np
import torch
from torch import Tensor

TILE = 16
CHUNK = 24  # tiles per LDS-ring chunk; multiple of every group size used (4 and 6)
GROUP_HIDDEN = 4  # out tiles processed together in the hidden layers
MAX_WIDTH = 256  # generic kernel: activations live in 16 tiles x 4 registers per lane, two wavefronts per SIMD
MAX_WIDTH_WIDE = 512  # static-shape kernels only (one wavefront per SIMD, 32 + 32 activation tiles): zuko_amd/static_ar.py


@dataclass
class UniLayout:
    """How the last layer's rows are regrouped for a univariate transform with `total` params."""

    kind: int  # 0 = affine, 1 = rqs
    total: int
    fpl: int  # features per lane
    nt: int  # 16-row output tiles per group of 4*fpl features
    bins: int = 0


def uni_layout(kind: str, total: int, bins: int = 0) -> UniLayout | None:
    """C-ABI kind codes: 0 affine, 1 / 2 / 3 RQS with 8 / 4 / 16 bins, 4 circular RQS with 8 bins, 5 shifted SOS polynomial (3 polynomials of
    degree 4: SOSPF's defaults), 6 bounded Bernstein polynomial of degree 16 (BPF's default) — 5 and 6: forward, operand-split static-shape kernels only."""
    if kind == "sos" and total == 16:
        return UniLayout(5, 16, 1, 4)
    if kind == "bern" and total == 17:
        return UniLayout(6, 17, 1, 5)
    if kind == "affine" and total == 2:
        return UniLayout(0, 2, 2, 1)
    if kind == "rqs" and total == 3 * bins - 1 and bins in (8, 4, 16):
        return UniLayout({8: 1, 4: 2, 16: 3}[bins], total, 1, (total + 3) // 4, bins)
    if kind == "crqs" and bins == 8 and total == 23:
        return UniLayout(4, 23, 1, 6, 8)
    return None


def layout_supports(layout: UniLayout, features: int) -> bool:
    """Kinds 2-4 are built only for the LDS-staged epilogue: rows of x must be float4-addressable and
    the [128 x D] tile must fit beside the weight ring."""
    return layout.kind in (0, 1) or (features % 4 == 0 and features <= 128)


@dataclass
class ArPlan:
    layout: UniLayout
    n_layers: int  # linear layers (hidden + 1)
    din: int  # input width (features + context)
    features: int
    n_groups: int  # groups of 4*fpl features in the last layer
    n_chunks: int
    gather: list  # per layer: int32 index into W_l.flatten() (-1 -> 0), length = blocks*256
    layer_block0: list  # first stream block of each layer
    bias_gather: list  # per layer: int32 index into b_l (-1 -> 0)
    bias_off: list  # offset of each layer's bias image in the bias buffer
    skip: np.ndarray  # uint32 [(n_layers-1)*4 + n_groups]
    featmap: np.ndarray  # int32 [n_groups*4*fpl] original feature index or -1
    n_blocks: int
    dense_tiles: int  # tile pairs a dense evaluation would touch (for reporting)
    kept_tiles: int
    chunk: int = CHUNK
    otg_blocks_end: list = None
    group_chunk0: list = None  # (align_groups plans) first chunk / chunk count of every last-layer group
    group_nchunks: list = None
    # second stream for the static-shape kernel (csrc/fused_ar_static.hip): inside a kept (out-group, in-tile) block only the
    # 16x16 tiles that hold non-zero weights (the strictly upper tiles of the diagonal 64x64 blocks are dropped)
    fine_gather: list = None       # per layer, like `gather`
    fine_layer_block0: list = None
    fine_n_blocks: int = 0
    fine_n_chunks: int = 0
    fine_tilemask: np.ndarray = None  # uint8 [n_layers - 1, max_width / 64, max_width / 16]: bit t = out tile 4 otg + t multiplies in tile it
    fine_kept_tiles: int = 0       # tiles of the per-tile stream (without chunk padding)
    max_width: int = MAX_WIDTH     # 256: the generic kernel can run the plan; 512: static-shape kernels only
    widths: list = None            # hidden widths (units per hidden layer)


def _deps(masks: list[np.ndarray]) -> list[np.ndarray]:
    """Input-dependency sets of every layer's units (boolean products of the masks)."""
    out = []
    cur = masks[0].astype(bool)
    out.append(cur)
    for m in masks[1:]:
        cur = (m.astype(np.int64) @ cur.astype(np.int64)) > 0
        out.append(cur)
    return out


def build_plan(masks: list[Tensor], features: int, layout: UniLayout, chunk: int = CHUNK, align_groups: bool = False, max_width: int = MAX_WIDTH) -> ArPlan | None:
    """masks[l]: bool [out_l, in_l] of the conditioner's linear layers (last: features*total rows).
    `chunk`: tiles per LDS-ring chunk each layer is padded to.  `align_groups`: additionally start
    every last-layer group on a chunk boundary and record per-group / per-out-group stream positions,
    so that a sweep of the inverse can stream only the prefix of the network it needs."""
    M = [m.detach().cpu().numpy().astype(bool) for m in masks]
    L = len(M)
    if L < 2:
        return None
    din = M[0].shape[1]
    widths = [m.shape[0] for m in M[:-1]]
    if din > max_width or any(w > max_width for w in widths) or M[-1].shape[0] != features * layout.total:
        return None
    n_otg, n_itile = max_width // TILE // GROUP_HIDDEN, max_width // TILE
    deps = _deps(M)

    # unit permutations (stable sort by dependency count) and padded index lists (-1 = padding)
    def padded(idx: np.ndarray, mult: int) -> np.ndarray:
        n = -(-len(idx) // mult) * mult
        return np.concatenate([idx, -np.ones(n - len(idx), dtype=np.int64)])

    cols = padded(np.arange(din), TILE)  # layer-0 input order is the natural one
    perms = []
    for l in range(L - 1):
        key = deps[l].sum(axis=1)
        perms.append(padded(np.argsort(key, kind="stable"), TILE))

    fkey = deps[-1][:: layout.total].sum(axis=1)
    per_group = 4 * layout.fpl
    featmap = padded(np.argsort(fkey, kind="stable"), per_group).astype(np.int32)
    n_groups = len(featmap) // per_group

    gather, layer_block0, bias_gather, bias_off, skip = [], [], [], [], []
    block_cursor = 0
    bias_cursor = 0
    dense_tiles = kept_tiles = fine_kept = 0

    lane = np.arange(64)
    li, lq = lane % 16, lane // 16

    def block_index(rows: np.ndarray, ccols: np.ndarray, in_width: int) -> np.ndarray:
        """1 KiB tile image: element (lane, r) <- W[rows[lane%16], ccols[4*(lane//16) + r]]."""
        r = rows[li][:, None]  # [64, 1]
        c = ccols[(4 * lq)[:, None] + np.arange(4)[None, :]]  # [64, 4]
        idx = r * in_width + c
        idx[(r < 0) | (c < 0)] = -1
        return idx.reshape(-1)

    import os

    dense = os.environ.get("ZUKO_AMD_AR_DENSE", "0") == "1"  # profiling aid: keep every tile (no skipping)

    def tile_nonzero(mask: np.ndarray, rows: np.ndarray, ccols: np.ndarray) -> bool:
        rr, cc = rows[rows >= 0], ccols[ccols >= 0]
        if dense:
            return bool(rr.size and cc.size)
        return bool(rr.size and cc.size and mask[np.ix_(rr, cc)].any())

    def finish_layer(blocks: list[np.ndarray]) -> None:
        nonlocal block_cursor
        n = len(blocks)
        pad = -(-n // chunk) * chunk - n
        blocks = blocks + [-np.ones(256, dtype=np.int64)] * pad
        gather.append(np.concatenate(blocks).astype(np.int32) if blocks else np.zeros(0, np.int32))
        layer_block0.append(block_cursor)
        block_cursor += len(blocks)

    fine_gather, fine_layer_block0 = [], []
    fine_cursor = 0
    tilemask = np.zeros((L - 1, n_otg, n_itile), dtype=np.uint8)

    def finish_fine_layer(blocks: list[np.ndarray]) -> None:
        nonlocal fine_cursor
        n = len(blocks)
        blocks = blocks + [-np.ones(256, dtype=np.int64)] * (-(-n // chunk) * chunk - n)
        fine_gather.append(np.concatenate(blocks).astype(np.int32) if blocks else np.zeros(0, np.int32))
        fine_layer_block0.append(fine_cursor)
        fine_cursor += len(blocks)

    otg_blocks_end: list[list[int]] = []  # [layer][otg] blocks of the layer consumed once out-group otg is done
    group_chunk0: list[int] = []
    group_nchunks: list[int] = []
    in_cols = cols
    for l in range(L - 1):
        otg_blocks_end.append([])
        rows_all = perms[l]
        n_ot = len(rows_all) // TILE
        n_it = len(in_cols) // TILE
        blocks = []
        fblocks = []
        for otg in range(n_otg):
            bits = 0
            ots = [otg * GROUP_HIDDEN + t for t in range(GROUP_HIDDEN)]
            if ots[0] < n_ot:
                for it in range(n_it):
                    cc = in_cols[it * TILE : (it + 1) * TILE]
                    nz = False
                    for ot in ots:
                        if ot < n_ot:
                            dense_tiles += 1
                            if tile_nonzero(M[l], rows_all[ot * TILE : (ot + 1) * TILE], cc):
                                nz = True
                    if nz:
                        bits |= 1 << it
                        for t, ot in enumerate(ots):
                            rr = rows_all[ot * TILE : (ot + 1) * TILE] if ot < n_ot else -np.ones(TILE, dtype=np.int64)
                            blk = block_index(rr, cc, M[l].shape[1])
                            blocks.append(blk)
                            kept_tiles += 1
                            if l == 0 or (ot < n_ot and tile_nonzero(M[l], rr, cc)):  # (layer 1 keeps whole blocks: its columns are features, not sorted units)
                                fblocks.append(blk)
                                fine_kept += 1
                                tilemask[l, otg, it] |= 1 << t
            skip.append(bits)
            otg_blocks_end[l].append(len(blocks))
        finish_layer(blocks)
        finish_fine_layer(fblocks)
        b = -np.ones(max_width, dtype=np.int64)
        b[: len(rows_all)] = rows_all
        bias_gather.append(b.astype(np.int32))
        bias_off.append(bias_cursor)
        bias_cursor += max_width
        in_cols = rows_all

    # last layer: group g, tile t, tile row i  <->  feature slot / parameter
    n_it = len(in_cols) // TILE
    blocks = []
    bias_last = []
    for g in range(n_groups):
        tiles_rows = []
        for t in range(layout.nt):
            rows = -np.ones(TILE, dtype=np.int64)
            for i in range(TILE):
                m = 4 * t + (i & 3)
                fi, p = divmod(m, layout.total)
                if fi < layout.fpl:
                    f = featmap[g * per_group + (i >> 2) * layout.fpl + fi]
                    if f >= 0:
                        rows[i] = f * layout.total + p
            tiles_rows.append(rows)
            bias_last.append(rows)
        bits = 0
        g_start = len(blocks)
        for it in range(n_it):
            cc = in_cols[it * TILE : (it + 1) * TILE]
            dense_tiles += layout.nt
            if any(tile_nonzero(M[-1], rows, cc) for rows in tiles_rows):
                bits |= 1 << it
                for rows in tiles_rows:
                    blocks.append(block_index(rows, cc, M[-1].shape[1]))
                    kept_tiles += 1
                    fine_kept += 1
        skip.append(bits)
        if align_groups:
            n = len(blocks) - g_start
            blocks += [-np.ones(256, dtype=np.int64)] * (-(-n // chunk) * chunk - n)
            group_chunk0.append((block_cursor + g_start) // chunk)
            group_nchunks.append((len(blocks) - g_start) // chunk)
    finish_layer(blocks)
    if not align_groups:
        finish_fine_layer(blocks)
    bias_gather.append(np.concatenate(bias_last).astype(np.int32))
    bias_off.append(bias_cursor)

    return ArPlan(
        layout=layout,
        n_layers=L,
        din=din,
        features=features,
        n_groups=n_groups,
        n_chunks=block_cursor // chunk,
        gather=gather,
        layer_block0=layer_block0,
        bias_gather=bias_gather,
        bias_off=bias_off,
        skip=np.asarray(skip, dtype=np.uint32),
        featmap=featmap,
        n_blocks=block_cursor,
        dense_tiles=dense_tiles,
        kept_tiles=kept_tiles,
        chunk=chunk,
        otg_blocks_end=otg_blocks_end,
        group_chunk0=group_chunk0,
        group_nchunks=group_nchunks,
        fine_gather=None if align_groups else fine_gather,
        fine_layer_block0=None if align_groups else fine_layer_block0,
        fine_n_blocks=0 if align_groups else fine_cursor,
        fine_n_chunks=0 if align_groups else fine_cursor // chunk,
        fine_tilemask=tilemask,
        fine_kept_tiles=0 if align_groups else fine_kept,
        max_width=max_width,
        widths=widths,
    )


def partial_schedule(plan: ArPlan, g_lo: int, g_hi: int):
    """Chunk schedule + per-hidden-layer out-group limits for evaluating ONLY the last-layer groups
    [g_lo, g_hi) (plan built with align_groups=True): walk the skip masks backwards to find which
    hidden tiles those groups depend on, and keep the matching prefix of every layer's stream."""
    L = plan.n_layers
    need = 0
    for g in range(g_lo, g_hi):
        need |= int(plan.skip[(L - 1) * 4 + g])
    olim = [0] * (L - 1)
    for l in range(L - 2, -1, -1):
        otgs = sorted({it // GROUP_HIDDEN for it in range(16) if need >> it & 1})
        olim[l] = max(otgs) if otgs else -1
        nxt = 0
        for otg in range(olim[l] + 1):  # the stream is consumed as a prefix: every out-group up to the limit
            nxt |= int(plan.skip[l * 4 + otg])
        need = nxt
    sched = []
    for l in range(L - 1):
        nblk = plan.otg_blocks_end[l][olim[l]] if olim[l] >= 0 else 0
        c0 = plan.layer_block0[l] // plan.chunk
        sched += list(range(c0, c0 + -(-nblk // plan.chunk)))
    for g in range(g_lo, g_hi):
        sched += list(range(plan.group_chunk0[g], plan.group_chunk0[g] + plan.group_nchunks[g]))
    return sched, olim


# --------------------------------------------------------------------------------------------------
# device-side state: plan tables + the gathered weight stream, refreshed when parameters change
# --------------------------------------------------------------------------------------------------


GS_BLOCKS_PER_CHUNK = 8  # csrc/fused_ar_gsplit.hip: GS_CH = 24 one-KiB images = 8 blocks of three


def gsplit_gather(plan: ArPlan):
    """Gather indices of the GENERIC operand-split kernel's stream (csrc/fused_ar_gsplit.hip: zk_ar_forward_split), or None.

    The kernel walks the plan's own skip words: hidden layer l, out-group otg (4 out tiles), in-PAIR ip live when either in tile 2 ip /
    2 ip + 1 has its skip bit set -> 4 blocks (out tile 4 otg + t, t = 0..3); last layer: group g, live in-pair ip -> NT blocks.  A block
    is the 16 x 32 weight block as static_ar.split_tables describes it: lane L's 8 values are [tile(ot, 2 ip)[L, 0:4] | tile(ot, 2 ip + 1)
    [L, 0:4]] of the plan's f32 tiles (plan.gather), a tile the plan does not hold being zeros.  Returns (gathers, offsets, n_chunks):
    gathers[l] int32 [blocks_l * 512] into W_l.flatten() (-1 = zero), every layer padded to whole chunks of 8 blocks; offsets[l] the
    layer's first float in the stream; n_chunks >= 1."""
    if plan.max_width > MAX_WIDTH or plan.layout.kind > 4 or plan.group_chunk0:
        return None  # (wider plans, the polynomial maps and group-aligned plans have no generic split kernel)
    L, n_otg, n_it = plan.n_layers, MAX_WIDTH // TILE // GROUP_HIDDEN, MAX_WIDTH // TILE
    zero = -np.ones((64, 4), dtype=np.int64)

    def finish(blocks):
        pad = -len(blocks) % GS_BLOCKS_PER_CHUNK
        blocks = blocks + [-np.ones((64, 8), dtype=np.int64)] * pad
        return np.stack(blocks).astype(np.int32).reshape(-1) if blocks else np.zeros(0, np.int32)

    gathers = []
    for l in range(L - 1):
        tiles = plan.gather[l].reshape(-1, 64, 4)
        at, k = {}, 0
        for otg in range(n_otg):
            bits = int(plan.skip[l * n_otg + otg])
            for it in range(n_it):
                if bits >> it & 1:
                    for t in range(GROUP_HIDDEN):
                        at[(otg * GROUP_HIDDEN + t, it)] = k
                        k += 1
        blocks = []
        for otg in range(n_otg):
            bits = int(plan.skip[l * n_otg + otg])
            for ip in range(n_it // 2):
                if bits >> (2 * ip) & 3:
                    for t in range(GROUP_HIDDEN):
                        t0, t1 = at.get((otg * GROUP_HIDDEN + t, 2 * ip)), at.get((otg * GROUP_HIDDEN + t, 2 * ip + 1))
                        blocks.append(np.concatenate([zero if t0 is None else tiles[t0], zero if t1 is None else tiles[t1]], axis=1))
        gathers.append(finish(blocks))
    nt = plan.layout.nt
    tiles = plan.gather[L - 1].reshape(-1, 64, 4)
    blocks, k = [], 0
    for g in range(plan.n_groups):
        bits = int(plan.skip[(L - 1) * n_otg + g])
        at = {}
        for it in range(n_it):
            if bits >> it & 1:
                at[it] = k
                k += nt
        for ip in range(n_it // 2):
            if bits >> (2 * ip) & 3:
                for t in range(nt):
                    t0, t1 = at.get(2 * ip), at.get(2 * ip + 1)
                    blocks.append(np.concatenate([zero if t0 is None else tiles[t0 + t], zero if t1 is None else tiles[t1 + t]], axis=1))
    gathers.append(finish(blocks))
    offsets, cursor = [], 0
    for g in gathers:
        offsets.append(cursor * 768)
        cursor += len(g) // 512
    return gathers, offsets, max(1, cursor // GS_BLOCKS_PER_CHUNK)


# --------------------------------------------------------------------------------------------------
# precision of the conditioner's matrix products on the fused inference kernels
# --------------------------------------------------------------------------------------------------

_PRECISIONS = {"f16x2": "f16x2", "high": "f16x2", "bf16x3": "bf16x3", "highest": "bf16x3"}
_precision = None


def matmul_precision() -> str:
    """"f16x2" (default) or "bf16x3": how the fused autoregressive INFERENCE kernels evaluate the conditioner's f32 products — see set_matmul_precision."""
    global _precision
    if _precision is None:
        _precision = _PRECISIONS.get(os.environ.get("ZUKO_AMD_MATMUL", "f16x2").lower(), "f16x2")
    return _precision


def set_matmul_precision(mode: str) -> None:
    """The operand split of the conditioner's matrix products in the fused autoregressive inference kernels (torch.set_float32_matmul_precision's
    role; the reference computes them in f32, zuko/nn.py:217-218 — gfx950's f32 matrix instruction runs at 1/16 of the 16-bit rate):

      "f16x2" / "high" (default)  every operand as TWO f16 numbers scaled into f16's range by exact powers of two, three partial products
                                  (csrc/fused_ar_half_impl.h): 22 significant bits per operand, the error of an f32 dot product measured end to
                                  end (tests/test_gpu_flows.py, scripts/split_scheme_emulation.py).  Used when a conditioner's weights allow
                                  it (FusedAR._half_eligible), else the next form;
      "bf16x3" / "highest"        every operand as THREE bf16 numbers, six partial products (csrc/fused_ar_split_impl.h): 24 bits, any range.

    Training launches, the inverse and every other kernel are not affected.  Environment: ZUKO_AMD_MATMUL; ZUKO_AMD_EXACT_F32=1 selects the f32
    matrix instruction for everything."""
    global _precision
    if mode.lower() not in _PRECISIONS:
        raise ValueError(f"zuko_amd.set_matmul_precision: unknown mode '{mode}' (one of {sorted(_PRECISIONS)})")
    _precision = _PRECISIONS[mode.lower()]


HALF_SPREAD_LOG2 = 14        # a weight more than 2^14 below its layer's largest magnitude counts as "small" ...
HALF_SMALL_FRACTION = 0.01   # ... and a layer with more than 1 % of them keeps the three-part kernel
HALF_ROW_LOG2 = 12           # so does a layer with an out unit whose largest weight is more than 2^12 below the layer's
HALF_MIN_LOG2, HALF_MAX_LOG2 = -20, 40  # or whose largest magnitude is outside [2^-20, 2^40]: the weight scale 2^e then has e in [-25, 35], and with the kernel's
                                        # per-sample exponent in [-90, 90] the descale factor 2^-(e + ea) of an accumulator stays a normal f32 number


def half_scales(linears):
    """Per linear layer of a conditioner: (eligible, e) with 2^e the power of two its masked weights are stored with in the two-part kernels' stream
    (largest magnitude in [2^14, 2^15)).  One device synchronisation (a handful of floats) per call — made when the weights changed, inference only."""
    stats = []
    for m in linears:
        w = m.weight.detach()
        a = (w * m.mask).abs() if getattr(m, "mask", None) is not None else w.abs()
        big = a.amax()
        nz = a > 0
        small = (nz & (a < big * 2.0 ** -HALF_SPREAD_LOG2)).sum() / nz.sum().clamp_min(1)
        rowmax = a.amax(dim=1)
        minrow = torch.where(rowmax > 0, rowmax, big).amin()
        stats.append(torch.stack([big.float(), small.float(), minrow.float()]))
    out = []
    for big, small, minrow in torch.stack(stats).tolist():
        if big == 0.0:
            out.append((True, 0))
            continue
        ok = math.isfinite(big) and 2.0 ** HALF_MIN_LOG2 <= big <= 2.0 ** HALF_MAX_LOG2 and small <= HALF_SMALL_FRACTION and minrow >= big * 2.0 ** -HALF_ROW_LOG2
        out.append((bool(ok), 15 - math.frexp(big)[1] if math.isfinite(big) else 0))
    return out


def chunk_of(variant: int = 0) -> int:
    """Tiles per chunk the stream is padded to (= AR_CH of csrc/fused_ar.hip: a 3 x 24-tile LDS ring)."""
    return CHUNK


def default_variant() -> int:
    return 0  # reserved argument of the C ABI


class FusedAR:
    """Runs zk_ar_forward for one MaskedAutoregressiveTransform on one device."""

    def __init__(self, plan: ArPlan, device: torch.device, act: int, bound: float, slope: float, variant: int = 0) -> None:
        self.plan = plan
        self.variant = variant
        self.device = device
        self.act = act
        self.bound = bound
        self.slope = slope
        self.skip = torch.from_numpy(plan.skip.view(np.int32).copy()).to(device)
        self.featmap = torch.from_numpy(plan.featmap.copy()).to(device)
        self.gather = [torch.from_numpy(g).to(device) for g in plan.gather]
        self.bias_gather = [torch.from_numpy(g).to(device) for g in plan.bias_gather]
        self.stream = torch.empty(plan.n_blocks * 256, dtype=torch.float32, device=device)
        self.bias_floats = plan.bias_off[-1] + len(plan.bias_gather[-1])
        self.bias = torch.empty(self.bias_floats, dtype=torch.float32, device=device)
        self._stamp = None
        self._fine_stamp = None
        self.generic_ok = plan.max_width <= MAX_WIDTH and plan.layout.kind <= 4  # (wider plans and the polynomial maps exist only as static-shape kernels)
        self.eps = 1e-6     # Bernstein continuation margin (zuko/transforms.py:594); set by the caller when the transform was built with another
        self._gl = None     # SOS quadrature: ctypes arrays of the Gauss-Legendre nodes / weights on [0, 1] (kept alive here)
        self.static = None          # (StaticKernel, rev) of zuko_amd/static_ar.py once one has been found / compiled
        self._static_tried_rows = -1
        self.fine_gather = self.fine_stream = self.fine_offsets = None
        self.fine_n_chunks = 0
        # the two-part (f16 x 2) twin of an operand-split kernel: inference launches when the weights allow it (set_matmul_precision)
        self.half = None            # StaticKernel
        self.half_gather = self.half_stream = self.half_offsets = None
        self.half_n_chunks = 0
        self.half_ok = False        # the weights of _half_stamp are eligible and their stream is built
        self.half_descale = None    # per linear layer 2^-e
        self._half_stamp = None
        self._acquire_static(None)  # kernels already on disk (prebuilt or compiled earlier) are used whatever the batch size
        # the generic operand-split kernel (csrc/fused_ar_gsplit.hip): what run() launches while there is no static-shape kernel for the plan
        # (its tables are built on first use).  ZUKO_AMD_GSPLIT=0: keep such plans on the f32 matrix instruction; =force: even when there is one
        self.gs_mode = os.environ.get("ZUKO_AMD_GSPLIT", "1")  # (read when the state object is made — once per transform and device; zuko_amd.invalidate(module) re-reads it)
        self.gs = None              # (gathers on the device, offsets, n_chunks, stream) | False: the plan has no generic split kernel
        self._gs_stamp = self._seen_stamp = None

    @property
    def static_variant(self) -> int:
        """0: generic kernel; 1 / 2: a static-shape kernel with its primary / alternative first-layer pattern."""
        return 0 if self.static is None else 1 + self.static[1]

    def _acquire_static(self, rows) -> None:
        """Look the plan's static-shape kernel up (zuko_amd/static_ar.py); `rows` >= the JIT threshold allows compiling it."""
        from . import static_ar

        # (an f32 static kernel found on disk can still be replaced by the operand-split one once a batch is large enough to compile it)
        upgradable = self.static is not None and not self.static[0].meta.get("split") and static_ar.split_enabled() and rows is not None
        self._acquire_half(rows)
        if (self.static is not None and not upgradable) or self.plan.fine_gather is None or (rows is not None and rows <= self._static_tried_rows):
            return
        if rows is not None:
            self._static_tried_rows = rows if rows < static_ar.jit_min_rows() else 1 << 62
        found = static_ar.lookup(self.plan, self.plan.layout.kind, self.act, rows)
        if found is not None and (self.static is None or found[0] is not self.static[0]):
            self.static = found
            if found[0].meta.get("split"):
                t, gathers = static_ar.split_tables(self.plan, self.plan.layout.kind, self.act)
                self.fine_gather = [torch.from_numpy(g).to(self.device) for g in gathers]
                self.fine_offsets = [b * 256 for b in t["BASE"]] + [t["LAST_BASE"] * 256]
                self.fine_n_chunks = t["NCHUNK"]
                stream_images = t["STREAM_IMAGES"]
            else:
                self.fine_gather = [torch.from_numpy(g).to(self.device) for g in self.plan.fine_gather]
                self.fine_offsets = [b * 256 for b in self.plan.fine_layer_block0]
                self.fine_n_chunks = self.plan.fine_n_chunks
                stream_images = self.fine_n_chunks * 24
            self.fine_stream = torch.zeros(stream_images * 256, dtype=torch.float32, device=self.device)
            self._fine_stamp = None

    def _acquire_half(self, rows) -> None:
        """The plan's two-part kernel (zuko_amd/static_ar.py: lookup_half), looked up once per batch size class like the static kernel."""
        from . import static_ar

        if self.half is not None or self.plan.fine_gather is None or not static_ar.half_enabled() or (rows is not None and rows <= getattr(self, "_half_tried_rows", -1)):
            return
        if rows is not None:
            self._half_tried_rows = rows if rows < static_ar.jit_min_rows() else 1 << 62
        kern = static_ar.lookup_half(self.plan, self.plan.layout.kind, self.act, rows)
        if kern is not None:
            t, gathers = static_ar.half_tables(self.plan, self.plan.layout.kind, self.act)
            self.half = kern
            self.half_gather = [torch.from_numpy(g).to(self.device) for g in gathers]
            self.half_offsets = [b * 256 for b in t["BASE"]] + [t["LAST_BASE"] * 256]
            self.half_n_chunks = t["NCHUNK"]
            self.half_stream = torch.zeros(t["STREAM_IMAGES"] * 256, dtype=torch.float32, device=self.device)
            self._half_stamp = None

    def _refresh_half(self, linears, stamp) -> None:
        """(Re)build the two-part kernel's stream for the current weights: eligibility + per-layer scales (one synchronisation), then one gather per layer."""
        from . import _C
        from .ops import _ptr, _stream

        scales = half_scales(linears)
        self._half_stamp = stamp
        self.half_ok = all(ok for ok, _ in scales)
        if not self.half_ok:
            return
        self.half_descale = [2.0 ** -e for _, e in scales]
        for l, (m, (_, e)) in enumerate(zip(linears, scales)):
            w = m.weight.detach()
            if not w.is_contiguous():
                w = w.contiguous()
            nb = self.half_gather[l].numel() // 512
            if nb:
                _C.check(_C.lib().zk_gather_split_f16(_ptr(w), _ptr(m.mask.contiguous().view(torch.uint8)), _ptr(self.half_gather[l]), nb, _ptr(self.half_stream[self.half_offsets[l] :]), 2.0 ** e,
                                                      _stream()), "zk_gather_split_f16")

    def _half_args(self, **io):
        p = self.plan
        from . import _C
        from .ops import _ptr

        d = list(self.half_descale) + [1.0] * (4 - len(self.half_descale))
        return _C.args("zk_ar_args_v1", launcher=self.half.launcher, rev=0, uni_kind=p.layout.kind, D=p.features, wstream=_ptr(self.half_stream), bias=_ptr(self.bias),
                       bias_floats=self.bias_floats, featmap=_ptr(self.featmap), n_layers=p.n_layers, n_groups=p.n_groups, n_chunks=self.half_n_chunks, act=self.act,
                       bound=self.bound, slope=self.slope, wdescale0=d[0], wdescale1=d[1], wdescale2=d[2], wdescale3=d[3], **io)

    def _half_serves(self, y: Tensor) -> bool:
        """Whether this inference launch goes to the two-part kernel: one is held, the mode allows it, the current weights are eligible and gathered."""
        return (self.half is not None and self.half_ok and self._half_stamp is not None and self._half_stamp == self._seen_stamp and matmul_precision() == "f16x2"
                and self.gs_mode != "force" and (not self.half.meta["XLDS"] or (y.stride(0) % 4 == 0 and y.data_ptr() % 16 == 0)))

    def ready(self, rows: int) -> bool:
        """Whether run() can be served: always for plans the generic kernel covers; for wider ones only with a static-shape kernel
        (compiled now if `rows` makes it worth it)."""
        from . import static_ar

        self._acquire_static(static_ar.effective_rows(self, rows))  # (no-op once the best kernel for this plan is held or `rows` has been tried)
        return self.generic_ok or self.static is not None

    def _gsplit(self):
        """The generic operand-split kernel's tables if run() is to use that kernel, else None."""
        from . import static_ar

        if self.gs_mode == "0" or (self.static is not None and self.gs_mode != "force") or not self.generic_ok or not static_ar.split_enabled():
            return None
        if self.gs is None:
            t = gsplit_gather(self.plan)
            if t is None:
                self.gs = False
            else:
                gathers, offsets, n_chunks = t
                self.gs = ([torch.from_numpy(g).to(self.device) for g in gathers], offsets, n_chunks,
                           torch.zeros(n_chunks * GS_BLOCKS_PER_CHUNK * 768, dtype=torch.float32, device=self.device))
        return self.gs or None

    def refresh(self, linears, fine_only: bool = False) -> None:
        """(Re)build the weight streams / bias image if any parameter changed since they were last gathered.  The generic (block)
        stream and the static-shape kernels' per-tile stream carry separate stamps; fine_only (training forward, which re-gathers
        every step) skips the generic one."""
        from . import _C
        from .ops import _ptr, _stream

        from .nn import _param_stamp

        stamp = _param_stamp(linears)
        self._seen_stamp = stamp
        want_generic = self.generic_ok and not (fine_only and self.static is not None) and stamp != self._stamp
        want_fine = self.static is not None and stamp != self._fine_stamp
        gs = self._gsplit()
        want_gs = gs is not None and stamp != self._gs_stamp
        if self.half is not None and not fine_only and stamp != self._half_stamp and matmul_precision() == "f16x2":
            self._refresh_half(linears, stamp)
        if not (want_generic or want_fine or want_gs):
            return
        items = []
        for l, m in enumerate(linears):
            w = m.weight.detach()
            if not w.is_contiguous():
                w = w.contiguous()
            mask = m.mask.contiguous().view(torch.uint8)
            if want_generic:
                items.append((w, mask, self.gather[l], self.gather[l].numel(), self.stream[self.plan.layer_block0[l] * 256 :], 0))
            if want_fine:
                fdst = self.fine_stream[self.fine_offsets[l] :]
                if self.static[0].meta.get("split"):
                    items.append((w, mask, self.fine_gather[l], self.fine_gather[l].numel() // 512, fdst, 1))
                else:
                    items.append((w, mask, self.fine_gather[l], self.fine_gather[l].numel(), fdst, 0))
            if want_gs and gs[0][l].numel():
                items.append((w, mask, gs[0][l], gs[0][l].numel() // 512, gs[3][gs[1][l] :], 1))
            nb = self.bias_gather[l].numel()
            bdst = self.bias[self.plan.bias_off[l] :]
            if m.bias is None:
                bdst[:nb].zero_()
            else:
                items.append((m.bias.detach().contiguous(), None, self.bias_gather[l], nb, bdst, 0))
        _C.gather_multi(items, _stream())  # (one launch per eight gathers: they are tiny, and training re-gathers every step)
        if want_generic:
            self._stamp = stamp
        if want_fine:
            self._fine_stamp = stamp
        if want_gs:
            self._gs_stamp = stamp

    def run(self, inp: Tensor, y: Tensor, ladj: Tensor | None, accumulate: bool) -> None:
        """inp [N, DINP] (cat(x, c), zero-padded to a multiple of 4 columns), y [N, D], ladj [N]."""
        from . import _C
        from .ops import _ptr, _stream

        p = self.plan
        N = inp.shape[0]
        if self._half_serves(y):
            a = self._half_args(N=N, DIN=inp.shape[1], x=_ptr(inp), ldx=inp.stride(0), y=_ptr(y), ldy=y.stride(0), ladj=_ptr(ladj), accumulate=int(accumulate))
            _C.check(_C.lib().zk_ar_forward_static(a, _stream()), "zk_ar_forward_static")
            return
        gs = self._gsplit()
        if gs is not None and self._gs_stamp is not None and self._gs_stamp == self._seen_stamp and (p.layout.kind <= 1 or (p.features % 4 == 0 and y.stride(0) % 4 == 0 and y.data_ptr() % 16 == 0)):
            # (4 / 16 bins and the circular map exist for the LDS-staged epilogue only, like the generic f32 kernel's)
            a = self._generic_args(N=N, DIN=inp.shape[1], x=_ptr(inp), ldx=inp.stride(0), y=_ptr(y), ldy=y.stride(0), ladj=_ptr(ladj), accumulate=int(accumulate))
            a.wstream, a.n_chunks = gs[3].data_ptr(), gs[2]
            _C.check(_C.lib().zk_ar_forward_split(a, _stream()), "zk_ar_forward_split")
            return
        if self.static is not None:
            kern, rev = self.static
            # (a static-shape kernel that stages rows through LDS needs them 16-byte addressable; the generic kernel has an instantiation for the other case)
            if not kern.meta["XLDS"] or (y.stride(0) % 4 == 0 and y.data_ptr() % 16 == 0):
                extra = {}
                if p.layout.kind == 5:
                    if self._gl is None:
                        from .ops import _leggauss01

                        self._gl = _leggauss01(5)
                    extra = dict(gl_nodes01=ctypes.cast(self._gl[0], ctypes.c_void_p), gl_weights01=ctypes.cast(self._gl[1], ctypes.c_void_p))
                elif p.layout.kind == 6:
                    extra = dict(eps=float(self.eps))
                a = _C.args("zk_ar_args_v1", launcher=kern.launcher, rev=rev, uni_kind=p.layout.kind, N=N, D=p.features, DIN=inp.shape[1], x=_ptr(inp), ldx=inp.stride(0),
                            y=_ptr(y), ldy=y.stride(0), ladj=_ptr(ladj), accumulate=int(accumulate), wstream=_ptr(self.fine_stream), bias=_ptr(self.bias),
                            bias_floats=self.bias_floats, featmap=_ptr(self.featmap), n_layers=p.n_layers, n_groups=p.n_groups, n_chunks=self.fine_n_chunks, act=self.act,
                            bound=self.bound, slope=self.slope, **extra)
                _C.check(_C.lib().zk_ar_forward_static(a, _stream()), "zk_ar_forward_static")
                return
        if not self.generic_ok:
            raise RuntimeError("zuko_amd: this conditioner is wider than the generic fused kernel covers and has no static-shape kernel (FusedAR.ready() was not consulted)")
        a = self._generic_args(N=N, DIN=inp.shape[1], x=_ptr(inp), ldx=inp.stride(0), y=_ptr(y), ldy=y.stride(0), ladj=_ptr(ladj), accumulate=int(accumulate))
        _C.check(_C.lib().zk_ar_forward(a, _stream()), "zk_ar_forward")

    def _generic_args(self, **io):
        """The argument block of the generic kernel's entry points (include/zuko_amd.h: zk_ar_args_v1): the plan's tables + `io`."""
        from . import _C
        from .ops import _ptr

        p = self.plan
        return _C.args("zk_ar_args_v1", uni_kind=p.layout.kind, D=p.features, wstream=_ptr(self.stream), bias=_ptr(self.bias), bias_floats=self.bias_floats, skip=_ptr(self.skip),
                       featmap=_ptr(self.featmap), n_layers=p.n_layers, n_groups=p.n_groups, n_chunks=p.n_chunks, act=self.act, bound=self.bound, slope=self.slope, **io)

    def run_diag(self, inp: Tensor, y: Tensor, ladj: Tensor, bins: Tensor, knots: Tensor) -> None:
        """As run(), through the diagnostic twin of the kernel: also fills bins [N, D] int32 and knots [N, D, K+1]."""
        from . import _C
        from .ops import _ptr, _stream

        p = self.plan
        if self._half_serves(y) and p.layout.kind in (1, 2, 3) and y.stride(0) % 4 == 0 and y.data_ptr() % 16 == 0:
            # the product launch is the two-part kernel: ITS diagnostic instantiation
            a = self._half_args(N=inp.shape[0], DIN=inp.shape[1], x=_ptr(inp), ldx=inp.stride(0), y=_ptr(y), ldy=y.stride(0), ladj=_ptr(ladj), accumulate=0, bin_out=_ptr(bins), knots_out=_ptr(knots))
            _C.check(_C.lib().zk_ar_forward_static(a, _stream()), "zk_ar_forward_static")
            return
        if self.static is not None and self.static[0].meta.get("split") and y.stride(0) % 4 == 0 and y.data_ptr() % 16 == 0:
            # the product launch is an operand-split kernel: its own diagnostic instantiation (the generic kernel differs from it by rounding)
            kern, rev = self.static
            a = _C.args("zk_ar_args_v1", launcher=kern.launcher, rev=rev, uni_kind=p.layout.kind, N=inp.shape[0], D=p.features, DIN=inp.shape[1], x=_ptr(inp), ldx=inp.stride(0),
                        y=_ptr(y), ldy=y.stride(0), ladj=_ptr(ladj), accumulate=0, wstream=_ptr(self.fine_stream), bias=_ptr(self.bias), bias_floats=self.bias_floats,
                        featmap=_ptr(self.featmap), n_layers=p.n_layers, n_groups=p.n_groups, n_chunks=self.fine_n_chunks, act=self.act, bound=self.bound, slope=self.slope,
                        bin_out=_ptr(bins), knots_out=_ptr(knots))
            _C.check(_C.lib().zk_ar_forward_static(a, _stream()), "zk_ar_forward_static")
            return
        a = self._generic_args(N=inp.shape[0], DIN=inp.shape[1], x=_ptr(inp), ldx=inp.stride(0), y=_ptr(y), ldy=y.stride(0), ladj=_ptr(ladj), bin_out=_ptr(bins), knots_out=_ptr(knots))
        gs = self._gsplit()
        if gs is not None and self._gs_stamp is not None and self._gs_stamp == self._seen_stamp and p.features % 4 == 0 and y.stride(0) % 4 == 0 and y.data_ptr() % 16 == 0:
            # the product launch is the generic operand-split kernel: ITS diagnostic instantiation
            a.wstream, a.n_chunks = gs[3].data_ptr(), gs[2]
            _C.check(_C.lib().zk_ar_forward_split(a, _stream()), "zk_ar_forward_split")
            return
        err = _C.lib().zk_ar_forward_diag(a, _stream())
        _C.check(err, "zk_ar_forward_diag")

    def run_inverse_sweep(self, buf: Tensor, y: Tensor) -> None:
        """One sweep x <- f^{-1}(y | x) in place: buf [N, DINP] holds cat(x, c, 0-pad), y [N, D]."""
        from . import _C
        from .ops import _ptr, _stream

        a = self._generic_args(N=buf.shape[0], DIN=buf.shape[1], x=_ptr(buf), ldx=buf.stride(0), y_in=_ptr(y), ldy=y.stride(0), x_out=_ptr(buf), ldo=buf.stride(0))
        err = _C.lib().zk_ar_inverse_sweep(a, _stream())
        _C.check(err, "zk_ar_inverse_sweep")

    # ---- partial (wavefront) inverse ---------------------------------------------------------------

    def set_sweeps(self, order: np.ndarray, passes: int) -> None:
        """Pre-compute, for every sweep s of the inverse, the last-layer groups that hold the features of
        order s and the chunk schedule / out-group limits of the network prefix they depend on
        (plan must be group-aligned)."""
        import ctypes

        p = self.plan
        per_group = 4 * p.layout.fpl
        slot_order = np.where(p.featmap >= 0, order[np.maximum(p.featmap, 0)], -1)
        self.sweeps = []
        flat: list[int] = []
        for s_ in range(passes):
            slots = np.nonzero(slot_order == s_)[0]
            if slots.size == 0:
                self.sweeps.append(None)
                continue
            g0, g1 = int(slots.min() // per_group), int(slots.max() // per_group) + 1
            sched, olim = partial_schedule(p, g0, g1)
            if not sched:
                # root features (order 0, no context): every parameter is a bias, nothing is streamed.  The kernel still
                # wants a valid ring schedule; chunk 0 is prefetched and never consumed (olim = -1, skip bits all clear).
                sched = [0]
            self.sweeps.append((len(flat), len(sched), (ctypes.c_int * len(olim))(*olim), g0, g1))
            flat += sched
        self.sched_dev = torch.tensor(flat, dtype=torch.int32, device=self.device)

    def run_inverse_partial(self, buf: Tensor, y: Tensor, sweep: int) -> None:
        from . import _C
        from .ops import _ptr, _stream
        import ctypes

        entry = self.sweeps[sweep]
        if entry is None:
            return
        off, n_sched, olim, g0, g1 = entry
        sched_ptr = ctypes.c_void_p(self.sched_dev.data_ptr() + 4 * off)
        a = self._generic_args(N=buf.shape[0], DIN=buf.shape[1], x=_ptr(buf), ldx=buf.stride(0), y_in=_ptr(y), ldy=y.stride(0), x_out=_ptr(buf), ldo=buf.stride(0),
                               sched=sched_ptr, n_sched=n_sched, olim=olim, g0=g0, g1=g1)
        err = _C.lib().zk_ar_inverse_partial(a, _stream())
        _C.check(err, "zk_ar_inverse_partial")
