r"""Training path of a coupling transform as ONE autograd node.

What it replaces: autograd through `GeneralCouplingTransform.meta` + `CouplingTransform.call_and_ladj` (zuko/flows/coupling.py:128-136,
zuko/transforms.py:1037-1073; the reference trains everything through plain autograd, tests/test_flows.py:22-29): index_select of the two
halves, the dense `MLP` (zuko/nn.py:15), the affine map, the merge — nine autograd nodes per transform, whose backward of the two index
operations alone (a sort-based `indexing_backward`) cost 1.7 ms of a 17.2 ms RealNVP cfg4 step.

Here (y, ladj) = CouplingFn(x, c, weights...) with
    forward   x_a gather -> L x zk_gemm_f16x2 (bias + ReLU in the epilogue, the maximum of every activation left on the device for the next
              layer's operand scale) -> zk_affine_forward -> y (x with the moved half overwritten)
    backward  zk_univariate_backward (adjoint of the affine map) -> L x zk_gemm_f16x2 on W^T with the ReLU gate in the epilogue ->
              zk_wgrad_multi (weight + bias gradients of all layers in two launches) -> g_x assembled in place.
The weights are re-split into f16 lane images once per call (zk_amax_f32 + zk_wsplit_f16: two launches for all layers, both orientations).
Covered: affine univariate with the default shapes, a plain (Linear, ReLU)* conditioner in float32 whose layer widths are multiples of 4,
every parameter trainable.  Anything else returns None and the caller keeps the layer-wise autograd path.  ZUKO_AMD_NO_COUPLING_FN=1 switches it off.
"""

from __future__ import annotations

import ctypes
import os
from functools import partial

import torch
from torch import Tensor
from torch.autograd.function import once_differentiable

from . import _C

AMAX_WORDS = 2048  # ZK_AMAX_WORDS of include/zuko_amd.h


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def image_words(units: int, k: int) -> int:
    """int32 words of the lane images of a (units x k) weight operand (zk_wsplit_f16)."""
    return -(-units // 128) * -(-k // 32) * 4096


def amax(items) -> None:
    """items: [(tensor [rows, cols] (row stride arbitrary), out [AMAX_WORDS] int32)]."""
    cls = _C.STRUCTS["zk_amax_desc_v1"]
    arr = (cls * len(items))()
    for d, (t, out) in zip(arr, items):
        d.struct_size, d.rows, d.cols, d.ld, d.src, d.out = ctypes.sizeof(cls), t.shape[0], t.shape[1], t.stride(0), t.data_ptr(), out.data_ptr()
    _C.check(_C.lib().zk_amax_f32(len(items), ctypes.cast(arr, ctypes.c_void_p), _stream()), "zk_amax_f32")


def wsplit(items) -> None:
    """items: [(weight [out, in] contiguous, transposed: bool, amax [AMAX_WORDS], dst int32 [image_words])]."""
    cls = _C.STRUCTS["zk_wsplit_desc_v1"]
    arr = (cls * len(items))()
    for d, (w, transposed, am, dst) in zip(arr, items):
        out_f, in_f = w.shape
        d.struct_size, d.src, d.amax, d.dst = ctypes.sizeof(cls), w.data_ptr(), am.data_ptr(), dst.data_ptr()
        if transposed:  # operand of the dgrad: units = inputs, k = outputs
            d.units, d.k, d.unit_stride, d.k_stride = in_f, out_f, 1, in_f
        else:
            d.units, d.k, d.unit_stride, d.k_stride = out_f, in_f, in_f, 1
    _C.check(_C.lib().zk_wsplit_f16(len(items), ctypes.cast(arr, ctypes.c_void_p), _stream()), "zk_wsplit_f16")


def gemm(a: Tensor, a_amax: Tensor, images: Tensor, w_amax: Tensor, n_out: int, bias, act: int, gate, c_amax) -> Tensor:
    """act(a W'^T + bias) (* (gate > 0)); a [M, K] fp32 with 16-byte aligned rows."""
    M, K = a.shape
    c = torch.empty((M, n_out), dtype=torch.float32, device=a.device)
    err = _C.lib().zk_gemm_f16x2(M, K, n_out, _ptr(a), a.stride(0), _ptr(a_amax), _ptr(images), _ptr(w_amax), _ptr(bias), act, _ptr(gate), 0 if gate is None else gate.stride(0), 1,
                                 _ptr(c), c.stride(0), _ptr(c_amax), _stream())
    _C.check(err, "zk_gemm_f16x2")
    return c


class CouplingFn(torch.autograd.Function):
    """(y, ladj) of one affine coupling transform.  Inputs after `c`: weight_0, bias_0, weight_1, ... of the conditioner."""

    @staticmethod
    def forward(ctx, lazy, plan, slope: float, x: Tensor, c, *params):
        from .autograd import _fwd_any

        idx_a, idx_b = lazy.split_indices()
        ws, bs = params[0::2], params[1::2]
        L = len(ws)
        N = x.shape[0]
        dev = x.device
        xa = x.index_select(1, idx_a)
        inp = xa if c is None else torch.cat((xa, c), dim=1)
        xb = x.index_select(1, idx_b)
        am = torch.zeros((3 * L + 1, AMAX_WORDS), dtype=torch.int32, device=dev)  # [input, h_1..h_{L-1}, (unused) | W_0..W_{L-1} | g_phi, g_{L-1}..g_1]
        wd = [w.detach() if w.is_contiguous() else w.detach().contiguous() for w in ws]
        amax([(inp, am[0])] + [(wd[l], am[L + l]) for l in range(L)])
        sizes = [image_words(*w.shape) for w in wd] + [image_words(w.shape[1], w.shape[0]) for w in wd]  # forward operands, then the dgrad operands W^T
        pool = torch.empty(sum(sizes), dtype=torch.int32, device=dev)
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        img_f = [pool[offs[l] : offs[l + 1]] for l in range(L)]
        img_t = [pool[offs[L + l] : offs[L + l + 1]] for l in range(L)]
        wsplit([(wd[l], False, am[L + l], img_f[l]) for l in range(L)] + [(wd[l], True, am[L + l], img_t[l]) for l in range(L)])
        hs = [inp]
        h = inp
        for l in range(L):
            last = l + 1 == L
            h = gemm(h, am[l], img_f[l], am[L + l], wd[l].shape[0], None if bs[l] is None else bs[l].detach(), 0 if last else 1, None, None if last else am[l + 1])
            hs.append(h)
        phi = hs[-1].view(N, xb.shape[1], 2)
        meta = (0, 5.0, slope, (1, 1), ())
        yb, ladj = _fwd_any(meta, xb, phi, True)
        y = x.clone()
        y.index_copy_(1, idx_b, yb)
        ctx.lazy, ctx.plan, ctx.meta, ctx.L, ctx.has_c = lazy, plan, meta, L, c is not None
        ctx.save_for_backward(xb, *hs, *img_t, am)
        return y, ladj

    @staticmethod
    @once_differentiable  # raw-pointer HIP kernels on detached data: double backward must raise, not return graph-less gradients
    def backward(ctx, gy, gl):
        from .autograd import _adj_any

        L, plan = ctx.L, ctx.plan
        saved = ctx.saved_tensors
        xb, hs, img_t, am = saved[0], saved[1 : L + 2], saved[L + 2 : 2 * L + 2], saved[2 * L + 2]
        idx_a, idx_b = ctx.lazy.split_indices()
        N, dev = xb.shape[0], xb.device
        phi = hs[L].view(N, xb.shape[1], 2)
        gyb = torch.zeros_like(xb) if gy is None else gy.index_select(1, idx_b)
        glc = torch.zeros(N, dtype=torch.float32, device=dev) if gl is None else gl.contiguous()
        gxb, gphi = _adj_any(ctx.meta, xb, phi, gyb, glc, True)
        g = gphi.view(N, -1)
        amax([(g, am[2 * L])])
        gs = [None] * L  # gradient of layer l's pre-activation output
        gs[L - 1] = g
        need_in = ctx.needs_input_grad[3] or (ctx.has_c and ctx.needs_input_grad[4])
        for l in range(L - 1, -1, -1):
            if l == 0 and not need_in:
                break
            # g_{l-1} = (g_l W_l) * relu'(h_l): W_l^T plays the weight, the saved activation h_l (hs[l], the layer's INPUT) the gate
            g = gemm(g, am[2 * L + (L - 1 - l)], img_t[l], am[L + l], hs[l].shape[1], None, 0, hs[l] if l > 0 else None, am[2 * L + (L - l)] if l > 0 else None)
            if l > 0:
                gs[l - 1] = g
        res = {}
        for l0 in range(0, L, 4):  # (zk_wgrad_multi: up to four layers per pair of launches)
            ls = range(l0, min(L, l0 + 4))
            # (maxima: of g_l — g_phi for the last layer — and of the layer's input h_l, all left on the device by the GEMM epilogues)
            res.update(plan.wgrad_multi([(l, gs[l], hs[l]) for l in ls], amax={l: (am[2 * L + (L - 1 - l)], am[l]) for l in ls}))
        grads = []
        for l in range(L):
            grads += list(res[l])
        gx = gc = None
        if need_in:
            na = idx_a.shape[0]
            if ctx.needs_input_grad[3]:
                gx = torch.zeros((N, int(idx_a.shape[0] + idx_b.shape[0])), dtype=torch.float32, device=dev) if gy is None else gy.clone()
                gx.index_copy_(1, idx_b, gxb)
                gx.index_add_(1, idx_a, g[:, :na])
            if ctx.has_c and ctx.needs_input_grad[4]:
                gc = g[:, na:].contiguous()
        elif ctx.needs_input_grad[3]:
            gx = torch.zeros((N, int(idx_a.shape[0] + idx_b.shape[0])), dtype=torch.float32, device=dev) if gy is None else gy.clone()
            gx.index_copy_(1, idx_b, gxb)
        return (None, None, None, gx, gc, *grads)


def coupling(lazy, x: Tensor, c):
    """(y, ladj) of `lazy` (a GeneralCouplingTransform) at x [N, features] (c [N, context] or None) under autograd through CouplingFn, or None when
    the layer / batch is not covered."""
    from . import train
    from .nn import Linear, _act_code
    from .transforms import MonotonicAffineTransform

    if os.environ.get("ZUKO_AMD_NO_COUPLING_FN", "0") == "1" or os.environ.get("ZUKO_AMD_EXACT_F32", "0") == "1":
        return None
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0) or (c is not None and (c.dim() != 2 or c.dtype != torch.float32 or c.shape[0] != x.shape[0])):
        return None
    u = lazy.univariate
    f, kw = (u.func, dict(u.keywords)) if isinstance(u, partial) else (u, {})
    slope = kw.pop("slope", 1e-3)
    if f is not MonotonicAffineTransform or kw or (isinstance(u, partial) and u.args) or [tuple(s) for s in lazy.shapes] != [(), ()]:
        return None
    mods = list(lazy.hyper)
    lins, acts = mods[0::2], mods[1::2]
    if len(mods) != 2 * len(lins) - 1 or not all(type(m) is Linear for m in lins) or any(_act_code(m) != 1 for m in acts) or len(lins) > 8:
        return None
    if not all(l.weight.dtype == torch.float32 and l.bias is not None and l.weight.requires_grad and l.bias.requires_grad and l.weight.shape[1] % 4 == 0 and l.weight.shape[0] % 4 == 0
               for l in lins):
        return None
    if x.shape[1] != lazy.mask.numel() or lins[0].weight.shape[1] != int(lazy.mask.sum()) + (0 if c is None else c.shape[1]):
        return None
    plan, _ = train.plan_for(lazy.hyper, x.device)
    if plan is None or not all(plan.cs_flag[i] is not None and plan.pairs[i].shape[0] > 0 for i in range(len(lins))):
        return None
    params = []
    for l in lins:
        params += [l.weight, l.bias]
    xc = x if x.is_contiguous() else x.contiguous()
    return CouplingFn.apply(lazy, plan, float(slope), xc, None if c is None else c.contiguous(), *params)
