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
Covered: affine univariate with the default shapes, a plain (Linear, activation)* conditioner in float32 — one of ReLU, ELU, Tanh, Sigmoid, LeakyReLU
(their derivatives are functions of their outputs) — whose layer widths are multiples of 4,
every parameter trainable.  Anything else returns None and the caller keeps the layer-wise autograd path.  ZUKO_AMD_NO_COUPLING_FN=1 switches it off.
"""

from __future__ import annotations

import ctypes
import os
import weakref
from functools import partial

import torch
from torch import Tensor
from torch.autograd.function import once_differentiable

from . import _C

AMAX_WORDS = 2048  # ZK_AMAX_WORDS of include/zuko_amd.h


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream() -> int:
    return _C.stream()


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


def gemm(a: Tensor, a_amax: Tensor, images: Tensor, w_amax: Tensor, n_out: int, bias, act: int, gate, c_amax, gate_act: int = 1) -> Tensor:
    """act(a W'^T + bias) (* act'_{gate_act}(gate), the derivative from the activation's output); a [M, K] fp32 with 16-byte aligned rows."""
    M, K = a.shape
    c = torch.empty((M, n_out), dtype=torch.float32, device=a.device)
    err = _C.lib().zk_gemm_f16x2(M, K, n_out, _ptr(a), a.stride(0), _ptr(a_amax), _ptr(images), _ptr(w_amax), _ptr(bias), act, _ptr(gate), 0 if gate is None else gate.stride(0), gate_act,
                                 _ptr(c), c.stride(0), _ptr(c_amax), _stream())
    _C.check(err, "zk_gemm_f16x2")
    return c


def _maps(lazy):
    """(idx_a, idx_b, half) as int32 device tensors for zk_coupling_split / zk_coupling_merge, cached per version of the mask."""
    idx_a, idx_b = lazy.split_indices()
    cached = lazy.__dict__.get("_train_maps")
    if cached is None or cached[0] is not idx_a:
        half = torch.empty(idx_a.shape[0] + idx_b.shape[0], dtype=torch.int32, device=idx_a.device)
        half[idx_b] = torch.arange(idx_b.shape[0], dtype=torch.int32, device=idx_a.device)
        half[idx_a] = -1 - torch.arange(idx_a.shape[0], dtype=torch.int32, device=idx_a.device)
        cached = (idx_a, idx_a.to(torch.int32), idx_b.to(torch.int32), half)
        lazy.__dict__["_train_maps"] = cached
    return cached[1], cached[2], cached[3]


class CouplingFn(torch.autograd.Function):
    """(y, ladj) of one affine coupling transform.  Inputs after `c`: weight_0, bias_0, weight_1, ... of the conditioner."""

    @staticmethod
    def forward(ctx, lazy, plan, slope: float, act: int, x: Tensor, c, *params):
        ia, ib, half = _maps(lazy)
        ws, bs = params[0::2], params[1::2]
        L = len(ws)
        N, D = x.shape
        dev = x.device
        na, nb, C = ia.shape[0], ib.shape[0], 0 if c is None else c.shape[1]
        am = torch.zeros((3 * L + 1, AMAX_WORDS), dtype=torch.int32, device=dev)  # [input, h_1..h_{L-1}, (unused) | W_0..W_{L-1} | g_phi, g_{L-1}..g_1]
        inp = torch.empty((N, na + C), dtype=torch.float32, device=dev)
        xb = torch.empty((N, nb), dtype=torch.float32, device=dev)
        _C.check(_C.lib().zk_coupling_split(N, D, C, x.data_ptr(), x.stride(0), None if c is None else c.data_ptr(), 0 if c is None else c.stride(0), ia.data_ptr(), na, ib.data_ptr(), nb,
                                            inp.data_ptr(), xb.data_ptr(), am[0].data_ptr(), _stream()), "zk_coupling_split")
        wd = [w.detach() if w.is_contiguous() else w.detach().contiguous() for w in ws]
        amax([(wd[l], am[L + l]) for l in range(L)])
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
            h = gemm(h, am[l], img_f[l], am[L + l], wd[l].shape[0], None if bs[l] is None else bs[l].detach(), 0 if last else act, None, None if last else am[l + 1])
            hs.append(h)
        phi = hs[-1].view(N, xb.shape[1], 2)
        meta = (0, 5.0, slope, (1, 1), ())
        # (zk_affine_forward straight on the packed parameters — shift = phi[.., 0], scale = phi[.., 1], strides (2 nb, 2) — instead of ops.affine_forward's
        #  general broadcasting front end: 55 us of host time per transform)
        yb = torch.empty_like(xb)
        ladj = torch.empty(N, dtype=torch.float32, device=dev)
        pp = phi.data_ptr()
        _C.check(_C.lib().zk_affine_forward(0, N, nb, slope, xb.data_ptr(), pp, 2 * nb, 2, pp + 4, 2 * nb, 2, yb.data_ptr(), ladj.data_ptr(), 1, _stream()), "zk_affine_forward")
        y = torch.empty_like(x)
        _C.check(_C.lib().zk_coupling_merge(N, D, x.data_ptr(), x.stride(0), yb.data_ptr(), nb, None, 0, half.data_ptr(), y.data_ptr(), _stream()), "zk_coupling_merge")
        ctx.lazy, ctx.plan, ctx.meta, ctx.L, ctx.has_c, ctx.act = lazy, plan, meta, L, c is not None, act
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
        ia, ib, half = _maps(ctx.lazy)
        N, dev = xb.shape[0], xb.device
        phi = hs[L].view(N, xb.shape[1], 2)
        gyb = torch.zeros_like(xb) if gy is None else gy.index_select(1, idx_b)
        glc = torch.zeros(N, dtype=torch.float32, device=dev) if gl is None else gl.contiguous()
        gxb, gphi = _adj_any(ctx.meta, xb, phi, gyb, glc, True)
        g = gphi.view(N, -1)
        amax([(g, am[2 * L])])
        gs = [None] * L  # gradient of layer l's pre-activation output
        gs[L - 1] = g
        need_in = ctx.needs_input_grad[4] or (ctx.has_c and ctx.needs_input_grad[5])
        for l in range(L - 1, -1, -1):
            if l == 0 and not need_in:
                break
            # g_{l-1} = (g_l W_l) * relu'(h_l): W_l^T plays the weight, the saved activation h_l (hs[l], the layer's INPUT) the gate
            g = gemm(g, am[2 * L + (L - 1 - l)], img_t[l], am[L + l], hs[l].shape[1], None, 0, hs[l] if l > 0 else None, am[2 * L + (L - l)] if l > 0 else None, gate_act=ctx.act)
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
        na, nb = ia.shape[0], ib.shape[0]
        if ctx.needs_input_grad[4]:  # g_x = g_y on the kept half (+ the conditioner's input gradient), the adjoint's g_xb on the moved half: one pass
            gx = torch.empty((N, na + nb), dtype=torch.float32, device=dev)
            gyc = None if gy is None else (gy if gy.stride(1) == 1 else gy.contiguous())
            _C.check(_C.lib().zk_coupling_merge(N, na + nb, None if gyc is None else gyc.data_ptr(), 0 if gyc is None else gyc.stride(0), gxb.data_ptr(), nb,
                                                g.data_ptr() if need_in else None, g.stride(0) if need_in else 0, half.data_ptr(), gx.data_ptr(), _stream()), "zk_coupling_merge")
        if need_in and ctx.has_c and ctx.needs_input_grad[5]:
            gc = g[:, na:].contiguous()
        return (None, None, None, None, gx, gc, *grads)


_STATE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()  # lazy -> {"key": .., "value": (plan, params, slope, activation code) or None}


def _static_verdict(lazy, device, n_ctx: int):
    """(plan, params, slope) when `lazy` is covered on `device` with n_ctx context columns, else None.  Cached per module and per what the verdict
    depends on (parameter identities / requires_grad flags / mask version): the checks walk the conditioner, and a training step calls this once
    per transform."""
    from . import train
    from .nn import Linear, _act_code
    from .transforms import MonotonicAffineTransform

    mods = list(lazy.hyper)
    lins = mods[0::2]
    key = (str(device), n_ctx, lazy.mask._version, lazy.mask.data_ptr(), id(lazy.univariate), len(mods),
           tuple((id(l.weight), id(l.bias), l.weight.requires_grad, l.bias is not None and l.bias.requires_grad, l.weight.data_ptr()) for l in lins if hasattr(l, "weight")))
    per = _STATE.setdefault(lazy, {})
    if per.get("key") == key:
        return per["value"]
    value = None
    u = lazy.univariate
    f, kw = (u.func, dict(u.keywords)) if isinstance(u, partial) else (u, {})
    slope = kw.pop("slope", 1e-3)
    acts = mods[1::2]
    ok = f is MonotonicAffineTransform and not kw and not (isinstance(u, partial) and u.args) and [tuple(s) for s in lazy.shapes] == [(), ()]
    ok = ok and len(mods) == 2 * len(lins) - 1 and all(type(m) is Linear for m in lins) and len({_act_code(m) for m in acts}) <= 1 and all(_act_code(m) in train.TRAIN_ACTS and _act_code(m) != 0 for m in acts) and len(lins) <= 8
    ok = ok and all(l.weight.dtype == torch.float32 and l.bias is not None and l.weight.requires_grad and l.bias.requires_grad and l.weight.shape[1] % 4 == 0 and l.weight.shape[0] % 4 == 0
                    for l in lins)
    if ok:
        kept = int(lazy.split_indices()[0].shape[0])  # (cached indices: mask.sum() would be a device -> host synchronisation per call)
        ok = lins[0].weight.shape[1] == kept + n_ctx
    if ok:
        plan, _ = train.plan_for(lazy.hyper, device)
        ok = plan is not None and all(plan.cs_flag[i] is not None and plan.pairs[i].shape[0] > 0 for i in range(len(lins)))
    if ok:
        params = []
        for l in lins:
            params += [l.weight, l.bias]
        value = (plan, tuple(params), float(slope), (_act_code(acts[0]) if acts else 0))
    per["key"], per["value"] = key, value
    return value


def coupling(lazy, x: Tensor, c):
    """(y, ladj) of `lazy` (a GeneralCouplingTransform) at x [N, features] (c [N, context] or None) under autograd through CouplingFn, or None when
    the layer / batch is not covered."""
    if os.environ.get("ZUKO_AMD_NO_COUPLING_FN", "0") == "1" or os.environ.get("ZUKO_AMD_EXACT_F32", "0") == "1":
        return None
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0) or (c is not None and (c.dim() != 2 or c.dtype != torch.float32 or c.shape[0] != x.shape[0])):
        return None
    if x.shape[1] != lazy.mask.shape[0]:
        return None
    st = _static_verdict(lazy, x.device, 0 if c is None else c.shape[1])
    if st is None:
        return None
    plan, params, slope, act = st
    xc = x if x.is_contiguous() else x.contiguous()
    return CouplingFn.apply(lazy, plan, slope, act, xc, None if c is None else c.contiguous(), *params)
