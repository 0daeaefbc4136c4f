r"""Autograd support: `loss = -flow(c).log_prob(x).mean(); loss.backward()`.

The reference relies on PyTorch autograd through every ATen op of the forward pass.  Here each
HIP forward kernel that can sit on a differentiable path gets a `torch.autograd.Function` whose
backward is a HIP kernel too (csrc/backward.hip: fused adjoint of softclip/softmax/cumsum/exp/bin
gather/rational-quadratic; affine; base density; activations; csrc/backward_poly.hip: SOS and Bernstein polynomials on
forward-mode dual numbers).  The dgrad / wgrad of plain conditioner stacks run on the tile-skipping MFMA kernels of
csrc/train.hip (zuko_amd/train.py); `LinearFn` below serves the module trees those plans do not cover (residual blocks, SiLU / GELU)
layer by layer on the same tile GEMMs (`_linear_backward`): no library GEMM anywhere on the path.

Scope: fp32; affine, RQS (4, 8 or 16 bins), SOS polynomial (SOSPF default size) and (bounded) Bernstein polynomial
(BPF default sizes), forward (`log_prob`) and inverse (`rsample`) directions, with MaskedMLP / MLP conditioners
(zuko_amd/train.py).  When gradients are required the layer-wise kernels are used (the fused inference kernel keeps no
intermediates).
"""

from __future__ import annotations

import torch
from torch.autograd.function import once_differentiable
from torch import Tensor

from . import _C


def needs_grad(*ts) -> bool:
    return torch.is_grad_enabled() and any(t is not None and torch.is_tensor(t) and t.requires_grad for t in ts)


def _ptr(t):
    import ctypes

    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream() -> int:
    return _C.stream()


def _require_f32(*ts) -> None:
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise NotImplementedError("zuko_amd autograd kernels are fp32-only")


def _sum_to(g: Tensor, shape) -> Tensor:
    return g if tuple(g.shape) == tuple(shape) else g.sum_to_size(shape)


def _packed(parts: list[Tensor], batch) -> Tensor:
    """The parameter pieces as ONE contiguous [*batch, total] tensor; zero-copy when they already are
    consecutive slices of such a buffer (what `unpack(phi, shapes)` produces)."""
    total = sum(p.shape[-1] for p in parts)
    first = parts[0]
    ok = tuple(first.shape[:-1]) == tuple(batch) and first.stride(-1) == 1
    off = 0
    for p in parts:
        ok = ok and tuple(p.shape[:-1]) == tuple(batch) and p.stride() == first.stride() and p.data_ptr() == first.data_ptr() + off * first.element_size()
        off += p.shape[-1]
    if ok:
        base = torch.as_strided(first, tuple(batch) + (total,), first.stride())
        if base.is_contiguous() and base.data_ptr() % 16 == 0:
            return base
    return torch.cat([p.expand(tuple(batch) + (p.shape[-1],)) for p in parts], dim=-1).contiguous()


# ---- univariate maps: one description object, three autograd Functions ------------------------------------------------
# meta = (kind, bound, slope, sizes, extra): kind 0 affine [shift | scale], 1 RQS [w(K) | h(K) | d(K-1)], 2 SOS
# [a(P*L1) | constant?] with extra = (P, L1, has_const), 3 Bernstein [theta(M)] with extra = (bounded,).


def _split(meta, phi):
    kind, _, _, sizes, extra = meta
    pieces = list(phi.split(list(sizes), -1))
    if kind == 0:
        return [pieces[0].squeeze(-1), pieces[1].squeeze(-1)]
    if kind == 2:
        P, L1, has_const = extra
        return [pieces[0].unflatten(-1, (P, L1))] + ([pieces[1].squeeze(-1)] if has_const else [])
    return pieces


def _fwd_any(meta, xe, phi, reduce):
    """(y, ladj) through the forward kernels (no autograd)."""
    from . import ops

    kind, bound, slope, _, extra = meta
    pc = _split(meta, phi)
    with torch.no_grad():
        if kind == 0:
            return ops.affine_forward(xe, pc[0], pc[1], slope, reduce)
        if kind == 1:
            return ops.rqs_forward(xe, pc[0], pc[1], pc[2], bound, slope, reduce)
        if kind == 2:
            return ops.sos_forward(xe, pc[0], pc[1] if extra[2] else None, slope, reduce)
        return ops.bernstein_forward(xe, pc[0], extra[0], bound, reduce, eps=extra[1] if len(extra) > 1 else ops.BERN_EPS)


def _inv_any(meta, ye, phi):
    from . import ops

    kind, bound, slope, _, extra = meta
    pc = _split(meta, phi)
    with torch.no_grad():
        if kind == 0:
            return ops.affine_inverse(ye, pc[0], pc[1], slope)
        if kind == 1:
            return ops.rqs_inverse(ye, pc[0], pc[1], pc[2], bound, slope)
        if kind == 2:
            return ops.sos_inverse(ye, pc[0], pc[1] if extra[2] else None, slope)
        return ops.bernstein_inverse(ye, pc[0], extra[0], bound, eps=extra[1] if len(extra) > 1 else ops.BERN_EPS)


def _adj_any(meta, xe, phi, gy, gl, reduce):
    """(gx, gphi) = VJP of (y, ladj) w.r.t. (x, packed parameters): csrc/backward.hip, csrc/backward_poly.hip."""
    kind, bound, slope, sizes, extra = meta
    D = xe.shape[-1] if xe.dim() else 1
    N = xe.numel() // max(D, 1)
    gx, gphi = torch.empty_like(xe), torch.empty_like(phi)
    lib = _C.lib()
    if kind in (0, 1):
        K = sizes[0] if kind == 1 else 0
        err = lib.zk_univariate_backward(kind, N, D, K, bound, slope, _ptr(xe), _ptr(phi), _ptr(gy), _ptr(gl), int(reduce), _ptr(gx), _ptr(gphi), _stream())
        _C.check(err, "zk_univariate_backward")
    elif kind == 2:
        from .ops import _leggauss01

        P, L1, has_const = extra
        nodes, weights = _leggauss01(L1)
        err = lib.zk_sos_backward(N, D, P, L1, slope, nodes, weights, int(has_const), _ptr(xe), _ptr(phi), _ptr(gy), _ptr(gl), int(reduce), _ptr(gx), _ptr(gphi), _stream())
        if err == 1:
            raise NotImplementedError("zuko_amd: the SOS adjoint kernel is built for 15 coefficients per element (SOSPF default: 3 polynomials of degree 4)")
        _C.check(err, "zk_sos_backward")
    else:
        M = sizes[0]
        from .ops import BERN_EPS

        err = lib.zk_bernstein_backward(N, D, M, int(extra[0]), bound, float(extra[1]) if len(extra) > 1 else BERN_EPS, _ptr(xe), _ptr(phi), _ptr(gy), _ptr(gl), int(reduce),
                                        _ptr(gx), _ptr(gphi), _stream())
        if err == 1:
            raise NotImplementedError("zuko_amd: the Bernstein adjoint kernel is built for the BPF defaults (bounded with 17 / unbounded with 16 unconstrained coefficients)")
        _C.check(err, "zk_bernstein_backward")
    return gx, gphi


def _meta_of(kind, bound, slope, params, extra=()):
    """(meta, parts): `parts` are the parameter tensors reshaped to [..., n_i] pieces of the packed layout."""
    if kind == 0:
        parts = [p.unsqueeze(-1) for p in params]
    elif kind == 2:
        parts = [params[0].flatten(-2)] + ([params[1].unsqueeze(-1)] if len(params) > 1 and params[1] is not None else [])
        extra = (params[0].shape[-2], params[0].shape[-1], len(parts) > 1)
    else:
        parts = list(params)
    return (kind, bound, slope, tuple(p.shape[-1] for p in parts), tuple(extra)), parts


def _unparts(meta, pieces, pshapes):
    kind = meta[0]
    out = []
    for g, shape in zip(pieces, pshapes):
        if kind == 0 or (kind == 2 and len(shape) < g.dim()):
            g = g.squeeze(-1)
        if kind == 2 and len(out) == 0:
            g = g.unflatten(-1, (meta[4][0], meta[4][1]))
        out.append(_sum_to(g, shape))
    return out


class UnivariateFn(torch.autograd.Function):
    """(y, ladj) of a univariate map from separate parameter tensors (see _meta_of for the kinds)."""

    @staticmethod
    def forward(ctx, kind: int, bound: float, slope: float, reduce: bool, x: Tensor, *params: Tensor):
        live = [p for p in params if p is not None]
        _require_f32(x, *live)
        meta, parts = _meta_of(kind, bound, slope, live)
        batch = torch.broadcast_shapes(x.shape, *[p.shape[:-1] for p in parts])
        xe = x.expand(batch).contiguous()
        phi = _packed([p.detach() for p in parts], batch)
        y, ladj = _fwd_any(meta, xe, phi, reduce)
        ctx.meta, ctx.reduce, ctx.xshape, ctx.pshapes, ctx.nparams = meta, reduce, x.shape, [p.shape for p in live], len(params)
        ctx.save_for_backward(xe, phi)
        return y, ladj

    @staticmethod
    @once_differentiable  # raw-pointer HIP kernels on detached data: double backward (create_graph=True) must raise, not return graph-less grads
    def backward(ctx, gy, gl):
        xe, phi = ctx.saved_tensors
        gy_c = None if gy is None else gy.expand(xe.shape).contiguous()
        gl_c = None if gl is None else (gl.expand(xe.shape[:-1]) if ctx.reduce else gl.expand(xe.shape)).contiguous()
        gx, gphi = _adj_any(ctx.meta, xe, phi, gy_c, gl_c, ctx.reduce)
        grads = _unparts(ctx.meta, gphi.split(list(ctx.meta[3]), -1), ctx.pshapes)
        grads += [None] * (ctx.nparams - len(grads))
        return (None, None, None, None, _sum_to(gx, ctx.xshape), *grads)


class BernsteinFn(torch.autograd.Function):
    """(y, ladj) of the (bounded) Bernstein polynomial (kind 3; `bounded` travels in meta.extra)."""

    @staticmethod
    def forward(ctx, bounded, bound: float, reduce: bool, x: Tensor, theta: Tensor):
        _require_f32(x, theta)
        meta, parts = _meta_of(3, bound, 0.0, [theta], tuple(bounded) if isinstance(bounded, tuple) else (bounded,))  # (bounded[, eps])
        batch = torch.broadcast_shapes(x.shape, theta.shape[:-1])
        xe = x.expand(batch).contiguous()
        phi = _packed([theta.detach()], batch)
        y, ladj = _fwd_any(meta, xe, phi, reduce)
        ctx.meta, ctx.reduce, ctx.xshape, ctx.tshape = meta, reduce, x.shape, theta.shape
        ctx.save_for_backward(xe, phi)
        return y, ladj

    @staticmethod
    @once_differentiable  # raw-pointer HIP kernels on detached data: double backward (create_graph=True) must raise, not return graph-less grads
    def backward(ctx, gy, gl):
        xe, phi = ctx.saved_tensors
        gy_c = None if gy is None else gy.expand(xe.shape).contiguous()
        gl_c = None if gl is None else (gl.expand(xe.shape[:-1]) if ctx.reduce else gl.expand(xe.shape)).contiguous()
        gx, gphi = _adj_any(ctx.meta, xe, phi, gy_c, gl_c, ctx.reduce)
        return (None, None, None, _sum_to(gx, ctx.xshape), _sum_to(gphi, ctx.tshape))


class UnivariatePackedFn(torch.autograd.Function):
    """As UnivariateFn, for parameters that ARE one packed phi[..., D, total] tensor (what the conditioner emits): phi
    is the differentiable input itself, so autograd neither splits nor re-concatenates it (the cat of the three spline
    pieces' gradients was a 386 MB copy per transform at batch 2^16).  `meta` as built by _meta_of."""

    @staticmethod
    def forward(ctx, meta, reduce: bool, x: Tensor, phi: Tensor):
        _require_f32(x, phi)
        xe = x.expand(phi.shape[:-1]).contiguous()
        ph = phi.detach()
        if not ph.is_contiguous() or ph.data_ptr() % 16 != 0:
            ph = ph.contiguous()
        y, ladj = _fwd_any(meta, xe, ph, reduce)
        ctx.meta, ctx.reduce, ctx.xshape = meta, reduce, x.shape
        ctx.save_for_backward(xe, ph)
        return y, ladj

    @staticmethod
    @once_differentiable  # raw-pointer HIP kernels on detached data: double backward (create_graph=True) must raise, not return graph-less grads
    def backward(ctx, gy, gl):
        xe, phi = ctx.saved_tensors
        gy_c = None if gy is None else gy.expand(xe.shape).contiguous()
        gl_c = None if gl is None else (gl.expand(xe.shape[:-1]) if ctx.reduce else gl.expand(xe.shape)).contiguous()
        gx, gphi = _adj_any(ctx.meta, xe, phi, gy_c, gl_c, ctx.reduce)
        return (None, None, _sum_to(gx, ctx.xshape), gphi)


class UnivariateInverseFn(torch.autograd.Function):
    """x = f^{-1}(y; phi), differentiable (gradients through `rsample`, which the reference obtains by autograd through its
    inverse formulas, zuko/transforms.py:443, :534-548, and through Bisection.backward, zuko/utils.py:185-209; asserted by
    its tests/test_flows.py:46-54).  By the inverse function theorem, with f' = exp(ladj(x)):
        dL/dy = g_x / f'(x),      dL/dphi = -(df/dphi)^T (g_x / f'(x)),
    i.e. the forward map's own adjoint kernel evaluated at the solution x with the seed -g_x / f'(x)."""

    @staticmethod
    def forward(ctx, kind: int, bound: float, slope: float, extra, y: Tensor, *params: Tensor):
        live = [p for p in params if p is not None]
        _require_f32(y, *live)
        meta, parts = _meta_of(kind, bound, slope, live, extra)
        batch = torch.broadcast_shapes(y.shape, *[p.shape[:-1] for p in parts])
        ye = y.expand(batch).contiguous()
        phi = _packed([p.detach() for p in parts], batch)
        x = _inv_any(meta, ye, phi)
        ctx.meta, ctx.yshape, ctx.pshapes, ctx.nparams = meta, y.shape, [p.shape for p in live], len(params)
        ctx.save_for_backward(x, phi)
        return x

    @staticmethod
    @once_differentiable  # raw-pointer HIP kernels on detached data: double backward (create_graph=True) must raise, not return graph-less grads
    def backward(ctx, gx):
        x, phi = ctx.saved_tensors
        _, ladj = _fwd_any(ctx.meta, x, phi, False)
        gxc = gx.expand(x.shape).contiguous()
        gy, seed = torch.empty_like(x), torch.empty_like(x)
        _C.check(_C.lib().zk_inverse_seed(x.numel(), _ptr(gxc), _ptr(ladj.contiguous()), _ptr(gy), _ptr(seed), _stream()), "zk_inverse_seed")
        _, gphi = _adj_any(ctx.meta, x, phi, seed, None, False)
        grads = _unparts(ctx.meta, gphi.split(list(ctx.meta[3]), -1), ctx.pshapes)
        grads += [None] * (ctx.nparams - len(grads))
        return (None, None, None, None, _sum_to(gy, ctx.yshape), *grads)


BACKWARD_ACTS = (0, 1, 2, 3, 6, 7)  # activations whose derivative is a function of their output


_ALL_PAIRS: dict = {}  # (out blocks, in blocks, device) -> int32 [n, 2]: every 128 x 128 block of a weight


def _linear_backward(g: Tensor, x: Tensor, weight: Tensor, mask, need_x: bool, need_w: bool, need_b: bool):
    """(gx, gW, gb) of y = x (mask * W)^T + b from g = d loss / dy [N, out], x [N, in] — what autograd derives from zuko/nn.py:217-218 — on the
    kernels of csrc/train.hip: gx = g (mask * W) through zk_gemm_f32_skip, gW = mask * (g^T x) through zk_wgrad_f32 (operand-split bf16 products,
    f32 accumulation, split-K with a fixed reduction order), gb through zk_colsum_f32.  No library GEMM."""
    lib = _C.lib()
    N, out_f = g.shape
    in_f = x.shape[1]
    dev = g.device
    gx = gw = gb = None
    if need_x:
        wm = weight.detach() if mask is None else weight.detach() * mask
        wt = wm.t().contiguous()  # [in, out]: the "weight" of the product g wt^T
        gx = torch.empty((N, in_f), dtype=torch.float32, device=dev)
        _C.check(lib.zk_gemm_f32_skip(N, out_f, in_f, _ptr(g), g.stride(0), _ptr(wt), None, None, 0, None, 0, 0, _ptr(gx), in_f, _stream()), "zk_gemm_f32_skip")
    if need_w:
        ob, ib = -(-out_f // 128), -(-in_f // 128)
        key = (ob, ib, str(dev))
        pairs = _ALL_PAIRS.get(key)
        if pairs is None:
            pairs = _ALL_PAIRS[key] = torch.cartesian_prod(torch.arange(ob), torch.arange(ib)).to(torch.int32).reshape(-1, 2).contiguous().to(dev)
        npairs = pairs.shape[0]
        ns = max(1, lib.zk_wgrad_slices(max(N, 1), npairs))
        partial = torch.empty(ns * npairs * 128 * 128, dtype=torch.float32, device=dev)
        gw = torch.zeros((out_f, in_f), dtype=torch.float32, device=dev)
        m8 = None if mask is None else mask.to(torch.uint8).contiguous()
        _C.check(lib.zk_wgrad_f32(N, out_f, in_f, _ptr(g), g.stride(0), _ptr(x), x.stride(0), _ptr(pairs), npairs, _ptr(partial), _ptr(m8), _ptr(gw), 0, None, None, _stream()), "zk_wgrad_f32")
    if need_b:
        ws = torch.empty(lib.zk_colsum_slices(max(N, 1)) * out_f, dtype=torch.float32, device=dev)
        gb = torch.empty(out_f, dtype=torch.float32, device=dev)
        if N == 0:
            gb.zero_()
        else:
            _C.check(lib.zk_colsum_f32(N, out_f, _ptr(g), g.stride(0), _ptr(ws), _ptr(gb), 0, _stream()), "zk_colsum_f32")
    return gx, gw, gb


class LinearFn(torch.autograd.Function):
    """y = act(x (mask * W)^T + b): forward = zk_linear, backward = act' kernel + the tile GEMMs of csrc/train.hip (_linear_backward)."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias, mask, act: int):
        from . import ops

        _require_f32(x, weight, bias)
        with torch.no_grad():
            y = ops.linear(x.detach(), weight.detach(), None if bias is None else bias.detach(), mask, act)
        ctx.act, ctx.has_bias = act, bias is not None
        ctx.save_for_backward(x, weight, mask if mask is not None else torch.empty(0, device=x.device), y)
        return y

    @staticmethod
    @once_differentiable  # raw-pointer HIP kernels on detached data: double backward (create_graph=True) must raise, not return graph-less grads
    def backward(ctx, gy):
        x, weight, mask, y = ctx.saved_tensors
        mask = mask if mask.numel() else None
        out_f, in_f = weight.shape
        g2 = gy.reshape(-1, out_f).contiguous()
        if ctx.act != 0:
            gin = torch.empty_like(g2)
            err = _C.lib().zk_act_backward(g2.numel(), _ptr(y.reshape(-1, out_f).contiguous()), _ptr(g2), ctx.act, _ptr(gin), _stream())
            _C.check(err, "zk_act_backward")
            g2 = gin
        x2 = x.reshape(-1, in_f)
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        gx, gw, gb = _linear_backward(g2, x2, weight, mask, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2])
        return (None if gx is None else gx.reshape(x.shape)), gw, gb, None, None


class DiagNormalLogProbFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z: Tensor, loc: Tensor, scale: Tensor, ladj):
        from . import ops

        _require_f32(z)
        with torch.no_grad():
            out = ops.diag_normal_log_prob(z.detach(), loc.detach(), scale.detach(), None if ladj is None else ladj.detach())
        ctx.has_ladj = ladj is not None
        ctx.save_for_backward(z, loc, scale)
        return out

    @staticmethod
    @once_differentiable  # raw-pointer HIP kernels on detached data: double backward (create_graph=True) must raise, not return graph-less grads
    def backward(ctx, g):
        z, loc, scale = ctx.saved_tensors
        D = z.shape[-1]
        z2 = z.reshape(-1, D).contiguous()
        gz = torch.empty_like(z2)
        gc = g.expand(z.shape[:-1]).reshape(-1).contiguous()
        err = _C.lib().zk_diag_normal_backward(z2.shape[0], D, _ptr(z2), _ptr(loc.contiguous()), _ptr(scale.contiguous()), _ptr(gc), _ptr(gz), _stream())
        _C.check(err, "zk_diag_normal_backward")
        return gz.reshape(z.shape), None, None, (g if ctx.has_ladj else None)
