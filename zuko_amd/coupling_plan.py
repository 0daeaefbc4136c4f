r"""Host-side plan of the fused coupling kernel (csrc/fused_coupling.hip).

One `GeneralCouplingTransform` (zuko/flows/coupling.py:25-139) = dense MLP on the pass-through half (+ context) followed by
the affine map of the other half.  The kernel streams the MLP's weights as 1 KiB MFMA A-operand images in consumption
order; this module builds that order (gather indices into the concatenated layer weights), the bias image and the two
index maps that implement `CouplingTransform`'s split / merge (zuko/transforms.py:1037-1048) inside the kernel.
(tests/plan_emulators.py walks the same tables in numpy for the CPU tests.)
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

TILE = 16
CHUNK = 24
HALF_CHUNK = 16  # images per ring chunk of the two-part kernel (csrc/fused_coupling.hip: CPH_CH)
RING = 3     # CP_NR: ring slots
MAX_T = 32    # CP_T: activation tiles (hidden width <= 512)
MAX_IT = 16   # CP_IT: input tiles (conditioner inputs <= 256)


@dataclass
class CouplingPlan:
    n_layers: int
    din: int
    nit: int
    widths: list
    tiles: list
    moved: int
    n_groups: int
    gather: np.ndarray       # int32 [n_blocks * 256] into the concatenated weights (-1 -> 0)
    n_blocks: int
    n_chunks: int
    bias_gather: np.ndarray  # int32 into the concatenated biases (-1 -> 0)
    bias_off: list
    amap: np.ndarray         # int32 [nit * 16]
    fmap: np.ndarray         # int32 [n_groups * 8]
    features: int
    context: int
    # operand-split stream (csrc/fused_coupling.hip: coupling_kernel_split; only for the static shape: 128 inputs, hidden [512] * k)
    split_gather: np.ndarray = None  # int32 [split_blocks * 512] (lane-major, 8 per lane) into the concatenated weights (-1 -> 0)
    split_chunks: int = 0
    split_layer_blocks: list = None  # blocks of every linear layer inside split_gather (the two-part stream scales each layer by its own power of two)


def build_coupling_plan(shapes, idx_a, idx_b, features: int, context: int, chunk: int = CHUNK):
    """shapes: [(out, in)] of the MLP's linear layers; idx_a / idx_b: columns of x that pass through / are transformed.
    Returns None when the network does not fit the kernel (widths > 512, inputs > 256, fewer than two layers)."""
    L = len(shapes)
    idx_a, idx_b = np.asarray(idx_a, dtype=np.int64), np.asarray(idx_b, dtype=np.int64)
    if L < 2 or L > 8:
        return None
    din = shapes[0][1]
    moved = len(idx_b)
    if din != len(idx_a) + context or shapes[-1][0] != 2 * moved:
        return None
    widths = [s[0] for s in shapes[:-1]]
    if din > MAX_IT * TILE or any(w > MAX_T * TILE for w in widths) or any(shapes[l + 1][1] != widths[l] for l in range(L - 1)):
        return None
    if (CHUNK * RING * 256 + 4 * 16 * (features + context + 8)) * 4 > 150 * 1024:
        return None
    nit = -(-din // TILE)
    tiles = [-(-w // TILE) for w in widths]
    n_groups = -(-moved // 8)
    w_off, b_off = [], []
    acc = 0
    for o, i in shapes:
        w_off.append(acc)
        acc += o * i
    acc = 0
    for o, _ in shapes:
        b_off.append(acc)
        acc += o
    lane = np.arange(64)
    li, lq = lane % 16, lane // 16

    def image(l, rows, cols):
        r = rows[li][:, None]
        c = cols[(4 * lq)[:, None] + np.arange(4)[None, :]]
        idx = w_off[l] + r * shapes[l][1] + c
        idx[(r < 0) | (c < 0)] = -1
        return idx.reshape(-1)

    def span(t, limit):
        v = np.arange(t * TILE, (t + 1) * TILE)
        v[v >= limit] = -1
        return v

    def last_rows(g):
        rows = -np.ones(TILE, dtype=np.int64)
        for i in range(TILE):
            fi, p = divmod(i & 3, 2)
            slot = g * 8 + (i >> 2) * 2 + fi
            if slot < moved:
                rows[i] = slot * 2 + p
        return rows

    blocks = []

    def pad():
        n = len(blocks)
        blocks.extend([-np.ones(256, dtype=np.int64)] * (-(-n // chunk) * chunk - n))

    for l in range(L - 1):
        n_in, in_w = (nit, din) if l == 0 else (tiles[l - 1], widths[l - 1])
        for otg in range(-(-tiles[l] // 4)):
            for it in range(n_in):
                for t in range(4):
                    blocks.append(image(l, span(otg * 4 + t, widths[l]), span(it, in_w)))
        pad()
    for g in range(n_groups):
        for it in range(tiles[-1]):
            blocks.append(image(L - 1, last_rows(g), span(it, widths[-1])))
    pad()
    gather = np.concatenate(blocks).astype(np.int32)

    bias_gather, bias_off = [], []
    cur = 0
    for l in range(L - 1):
        units = np.arange(MAX_T * TILE)
        bias_gather.append(np.where(units < widths[l], b_off[l] + units, -1))
        bias_off.append(cur)
        cur += MAX_T * TILE
    bias_off.append(cur)
    lastb = []
    for g in range(n_groups):
        rows = last_rows(g)
        lastb.append(np.where(rows >= 0, b_off[L - 1] + np.maximum(rows, 0), -1))
    bias_gather.append(np.concatenate(lastb))

    # operand-split twin of the stream for the static shape: a block = (out tile, PAIR of in tiles) as three bf16 images; hidden layers
    # in steps of (4 out tiles, 1 in pair), the last layer group by group, pair by pair.  Every layer and every group fills whole chunks.
    split_gather, split_chunks, split_layer_blocks = None, 0, None
    if nit == 8 and all(w == 512 for w in widths):
        def block(l, rows, ip, in_w):
            lo, hi = image(l, rows, span(2 * ip, in_w)).reshape(64, 4), image(l, rows, span(2 * ip + 1, in_w)).reshape(64, 4)
            return np.concatenate([lo, hi], axis=1).reshape(-1)

        sblocks, split_layer_blocks = [], []
        for l in range(L - 1):
            n_in, in_w = (nit, din) if l == 0 else (tiles[l - 1], widths[l - 1])
            n0 = len(sblocks)
            for otg in range(tiles[l] // 4):
                for ip in range(n_in // 2):
                    for t in range(4):
                        sblocks.append(block(l, span(otg * 4 + t, widths[l]), ip, in_w))
            assert (3 * len(sblocks)) % chunk == 0
            split_layer_blocks.append(len(sblocks) - n0)
        n0 = len(sblocks)
        for g in range(n_groups):
            for ip in range(tiles[-1] // 2):
                sblocks.append(block(L - 1, last_rows(g), ip, widths[-1]))
        assert (3 * len(sblocks)) % chunk == 0
        split_layer_blocks.append(len(sblocks) - n0)
        split_gather, split_chunks = np.concatenate(sblocks).astype(np.int32), 3 * len(sblocks) // chunk

    amap = -np.ones(nit * TILE, dtype=np.int64)
    amap[: len(idx_a)] = idx_a
    for c in range(context):
        amap[len(idx_a) + c] = -(2 + c)
    fmap = -np.ones(n_groups * 8, dtype=np.int64)
    fmap[:moved] = idx_b
    return CouplingPlan(
        n_layers=L, din=din, nit=nit, widths=widths, tiles=tiles, moved=moved, n_groups=n_groups, gather=gather, n_blocks=len(blocks),
        n_chunks=len(blocks) // chunk, bias_gather=np.concatenate(bias_gather).astype(np.int32), bias_off=bias_off, amap=amap.astype(np.int32),
        fmap=fmap.astype(np.int32), features=features, context=context, split_gather=split_gather, split_chunks=split_chunks, split_layer_blocks=split_layer_blocks,
    )


class FusedCoupling:
    """Runs zk_coupling_forward for one GeneralCouplingTransform on one device."""

    def __init__(self, plan: CouplingPlan, device, act: int, slope: float) -> None:
        import ctypes

        import torch

        self.plan, self.device, self.act, self.slope = plan, device, act, slope
        self.gather = torch.from_numpy(plan.gather).to(device)
        self.bias_gather = torch.from_numpy(plan.bias_gather).to(device)
        self.amap = torch.from_numpy(plan.amap).to(device)
        self.fmap = torch.from_numpy(plan.fmap).to(device)
        import os

        # operand-split kernel (6 bf16 matrix products per f32 product) when the plan has its stream; ZUKO_AMD_EXACT_F32=1 keeps the f32 instruction
        self.split = plan.split_gather is not None and act == 1 and os.environ.get("ZUKO_AMD_EXACT_F32", "0") != "1"
        if self.split:
            self.gather = torch.from_numpy(plan.split_gather).to(device)
            self.stream = torch.empty(plan.split_chunks * CHUNK * 256, dtype=torch.float32, device=device)
        else:
            self.stream = torch.empty(plan.n_blocks * 256, dtype=torch.float32, device=device)
        # two-part (f16 x 2) twin of the split stream: the same blocks, two images each, 16-image chunks; used when the weights allow it (fused.half_scales)
        self.half_able = self.split and plan.n_layers <= 4 and all((2 * b) % HALF_CHUNK == 0 for b in plan.split_layer_blocks)
        self.half_stream = torch.empty(2 * (len(plan.split_gather) // 512) * 256, dtype=torch.float32, device=device) if self.half_able else None
        self.half_ok, self.half_descale, self._half_stamp = False, None, None
        self.bias = torch.empty(len(plan.bias_gather), dtype=torch.float32, device=device)
        nl = plan.n_layers
        self.bias_off = (ctypes.c_int * nl)(*[int(v) for v in plan.bias_off])
        self.tiles = (ctypes.c_int * (nl - 1))(*[int(v) for v in plan.tiles])
        self.widths = (ctypes.c_int * (nl - 1))(*[int(v) for v in plan.widths])
        self._stamp = None

    def refresh(self, lins) -> None:
        import torch

        from . import _C
        from .nn import _param_stamp
        from .ops import _ptr, _stream

        from . import fused

        stamp = _param_stamp(lins)
        want_half = self.half_able and fused.matmul_precision() == "f16x2" and self._half_stamp != stamp
        if stamp == self._stamp and not want_half:
            return
        lib = _C.lib()
        wcat = torch.cat([l.weight.detach().reshape(-1) for l in lins])
        if want_half:  # eligibility + per-layer powers of two (one synchronisation per weight version), then one gather per layer
            self._half_stamp, self.half_ok = stamp, False
            scales = fused.half_scales(lins)
            if all(ok for ok, _ in scales):
                b0 = 0
                for (ok, e), nb in zip(scales, self.plan.split_layer_blocks):
                    _C.check(lib.zk_gather_split_f16(_ptr(wcat), None, _ptr(self.gather[b0 * 512 :]), nb, _ptr(self.half_stream[b0 * 512 :]), 2.0 ** e, _stream()), "zk_gather_split_f16")
                    b0 += nb
                self.half_descale = [2.0 ** -e for _, e in scales] + [1.0] * (4 - len(scales))
                self.half_ok = True
        if stamp == self._stamp:
            return
        bcat = torch.cat([(l.bias.detach() if l.bias is not None else torch.zeros(l.weight.shape[0], device=self.device)).reshape(-1) for l in lins])
        if self.split:
            _C.check(lib.zk_gather_split_bf16(_ptr(wcat), None, _ptr(self.gather), self.gather.numel() // 512, _ptr(self.stream), _stream()), "zk_gather_split_bf16")
        else:
            _C.check(lib.zk_gather_f32(_ptr(wcat), None, _ptr(self.gather), self.gather.numel(), _ptr(self.stream), _stream()), "zk_gather_f32")
        _C.check(lib.zk_gather_f32(_ptr(bcat), None, _ptr(self.bias_gather), self.bias_gather.numel(), _ptr(self.bias), _stream()), "zk_gather_f32")
        self._stamp = stamp

    def run(self, x, ctx, inverse: bool = False):
        """x [N, D] fp32 (row stride arbitrary), ctx [N, C] or None -> (y [N, D], ladj [N]).  inverse=True: x holds the transform's
        output and the result is its preimage; ladj is log|det dy/dx| of the FORWARD map at that preimage either way."""
        import torch

        from . import _C
        from .ops import _ptr, _stream

        p = self.plan
        N = x.shape[0]
        y = torch.empty((N, p.features), dtype=torch.float32, device=x.device)
        ladj = torch.empty(N, dtype=torch.float32, device=x.device)
        fn = _C.lib().zk_coupling_inverse if inverse else _C.lib().zk_coupling_forward
        from . import fused

        if self.half_able and self.half_ok and self._half_stamp == self._stamp and fused.matmul_precision() == "f16x2":
            d = self.half_descale
            a = _C.args("zk_coupling_args_v1", N=N, D=p.features, C=p.context, **{"in": _ptr(x)}, ldx=x.stride(0), ctx=_ptr(ctx), ldc=0 if ctx is None else ctx.stride(0), out=_ptr(y),
                        ldy=p.features, ladj=_ptr(ladj), accumulate=0, wstream=_ptr(self.half_stream), bias=_ptr(self.bias), bias_floats=self.bias.numel(), bias_off=self.bias_off,
                        amap=_ptr(self.amap), nit=p.nit, fmap=_ptr(self.fmap), n_groups=p.n_groups, n_layers=p.n_layers, tiles=self.tiles, widths=self.widths,
                        n_chunks=2 * (len(p.split_gather) // 512) // HALF_CHUNK, act=self.act, slope=self.slope, static_ok=3, wdescale0=d[0], wdescale1=d[1], wdescale2=d[2], wdescale3=d[3])
            _C.check(fn(a, _stream()), "zk_coupling_inverse" if inverse else "zk_coupling_forward")
            return y, ladj
        a = _C.args("zk_coupling_args_v1", N=N, D=p.features, C=p.context, **{"in": _ptr(x)}, ldx=x.stride(0), ctx=_ptr(ctx), ldc=0 if ctx is None else ctx.stride(0), out=_ptr(y),
                    ldy=p.features, ladj=_ptr(ladj), accumulate=0, wstream=_ptr(self.stream), bias=_ptr(self.bias), bias_floats=self.bias.numel(), bias_off=self.bias_off,
                    amap=_ptr(self.amap), nit=p.nit, fmap=_ptr(self.fmap), n_groups=p.n_groups, n_layers=p.n_layers, tiles=self.tiles, widths=self.widths,
                    n_chunks=p.split_chunks if self.split else p.n_chunks, act=self.act, slope=self.slope, static_ok=2 if self.split else 1)
        err = fn(a, _stream())
        _C.check(err, "zk_coupling_inverse" if inverse else "zk_coupling_forward")
        return y, ladj
