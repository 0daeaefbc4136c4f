r"""Autograd support: `loss = -flow(c).log_prob(x).mean(); loss.backward()`.

The reference relies on PyTorch autograd through every ATen op of the forward pass.  Here each
HIP forward kernel that can sit on a differentiable path gets a `torch.autograd.Function` whose
backward is a HIP kernel too (csrc/backward.hip: fused adjoint of softclip/softmax/cumsum/exp/bin
gather/rational-quadratic; affine; base density; activations).  The dgrad / wgrad of a conditioner
layer are plain GEMMs and go to the vendor library through `torch.mm` (the task rules reserve
hand-written MFMA kernels for the fused hot ops and allow hipBLASLt/rocBLAS for plain library GEMMs);
the mask is re-applied to the weight gradient.

Scope: fp32, forward direction (`log_prob`) of affine and RQS (4, 8 or 16 bins) transforms with
MaskedMLP / MLP conditioners.  Gradients through the inverse (`rsample`), SOS and Bernstein are not
provided yet and raise.  When gradients are required the layer-wise kernels are used (the fused
inference kernel keeps no intermediates).
"""

from __future__ import annotations

import torch
from torch import Tensor

from . import _C


def needs_grad(*ts) -> bool:
    return torch.is_grad_enabled() and any(t is not None and torch.is_tensor(t) and t.requires_grad for t in ts)


def _ptr(t):
    import ctypes

    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


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


class UnivariateFn(torch.autograd.Function):
    """kind 0: affine(shift, scale); kind 1: RQS(widths, heights, derivatives).  Returns (y, ladj)."""

    @staticmethod
    def forward(ctx, kind: int, bound: float, slope: float, reduce: bool, x: Tensor, *params: Tensor):
        from . import ops

        _require_f32(x, *params)
        parts = [p.unsqueeze(-1) for p in params] if kind == 0 else list(params)
        batch = torch.broadcast_shapes(x.shape, *[p.shape[:-1] for p in parts])
        xe = x.expand(batch).contiguous()
        phi = _packed([p.detach() for p in parts], batch)
        sizes = [p.shape[-1] for p in parts]
        pieces = phi.split(sizes, -1)
        with torch.no_grad():
            if kind == 0:
                y, ladj = ops.affine_forward(xe, pieces[0].squeeze(-1), pieces[1].squeeze(-1), slope, reduce)
            else:
                y, ladj = ops.rqs_forward(xe, pieces[0], pieces[1], pieces[2], bound, slope, reduce)
        ctx.kind, ctx.bound, ctx.slope, ctx.reduce = kind, bound, slope, reduce
        ctx.sizes, ctx.xshape, ctx.pshapes = sizes, x.shape, [p.shape for p in params]
        ctx.save_for_backward(xe, phi)
        return y, ladj

    @staticmethod
    def backward(ctx, gy, gl):
        xe, phi = ctx.saved_tensors
        D = xe.shape[-1] if xe.dim() else 1
        N = xe.numel() // max(D, 1)
        gx = torch.empty_like(xe)
        gphi = torch.empty_like(phi)
        gy_c = None if gy is None else gy.expand(xe.shape).contiguous()
        if gl is None:
            gl_c = None
        elif ctx.reduce:
            gl_c = gl.expand(xe.shape[:-1]).contiguous()
        else:
            gl_c = gl.expand(xe.shape).contiguous()
        K = ctx.sizes[0] if ctx.kind == 1 else 0
        err = _C.lib().zk_univariate_backward(ctx.kind, N, D, K, ctx.bound, ctx.slope, _ptr(xe), _ptr(phi), _ptr(gy_c), _ptr(gl_c), int(ctx.reduce),
                                              _ptr(gx), _ptr(gphi), _stream())
        _C.check(err, "zk_univariate_backward")
        pieces = gphi.split(ctx.sizes, -1)
        if ctx.kind == 0:
            pieces = [p.squeeze(-1) for p in pieces]
        grads = [_sum_to(p, s) for p, s in zip(pieces, ctx.pshapes)]
        return (None, None, None, None, _sum_to(gx, ctx.xshape), *grads)


def _kernel_fwd(kind, bound, slope, reduce, xe, phi, sizes):
    from . import ops

    pieces = phi.split(sizes, -1)
    if kind == 0:
        return ops.affine_forward(xe, pieces[0].squeeze(-1), pieces[1].squeeze(-1), slope, reduce)
    return ops.rqs_forward(xe, pieces[0], pieces[1], pieces[2], bound, slope, reduce)


class UnivariatePackedFn(torch.autograd.Function):
    """As UnivariateFn, for parameters that ARE one packed phi[..., D, total] tensor (what the conditioner emits): phi
    is the differentiable input itself, so autograd neither splits nor re-concatenates it (the cat of the three spline
    pieces' gradients was a 386 MB copy per transform at batch 2^16)."""

    @staticmethod
    def forward(ctx, kind: int, bound: float, slope: float, reduce: bool, sizes, x: Tensor, phi: Tensor):
        _require_f32(x, phi)
        xe = x.expand(phi.shape[:-1]).contiguous()
        ph = phi.detach()
        if not ph.is_contiguous() or ph.data_ptr() % 16 != 0:
            ph = ph.contiguous()
        with torch.no_grad():
            y, ladj = _kernel_fwd(kind, bound, slope, reduce, xe, ph, list(sizes))
        ctx.kind, ctx.bound, ctx.slope, ctx.reduce, ctx.sizes, ctx.xshape = kind, bound, slope, reduce, list(sizes), x.shape
        ctx.save_for_backward(xe, ph)
        return y, ladj

    @staticmethod
    def backward(ctx, gy, gl):
        xe, phi = ctx.saved_tensors
        D = xe.shape[-1]
        N = xe.numel() // max(D, 1)
        gx, gphi = torch.empty_like(xe), torch.empty_like(phi)
        gy_c = None if gy is None else gy.expand(xe.shape).contiguous()
        gl_c = None if gl is None else (gl.expand(xe.shape[:-1]) if ctx.reduce else gl.expand(xe.shape)).contiguous()
        K = ctx.sizes[0] if ctx.kind == 1 else 0
        err = _C.lib().zk_univariate_backward(ctx.kind, N, D, K, ctx.bound, ctx.slope, _ptr(xe), _ptr(phi), _ptr(gy_c), _ptr(gl_c), int(ctx.reduce),
                                              _ptr(gx), _ptr(gphi), _stream())
        _C.check(err, "zk_univariate_backward")
        return (None, None, None, None, None, _sum_to(gx, ctx.xshape), gphi)


class UnivariateInverseFn(torch.autograd.Function):
    """x = f^{-1}(y; phi) for the affine map / the spline, differentiable (gradients through `rsample`, which the
    reference obtains by autograd through its inverse formulas, zuko/transforms.py:443, :534-548; asserted by its
    tests/test_flows.py:46-54).  By the inverse function theorem, with f' = exp(ladj(x)):
        dL/dy = g_x / f'(x),      dL/dphi = -(df/dphi)^T (g_x / f'(x)),
    i.e. the forward map's own adjoint kernel evaluated at the solution x with the seed -g_x / f'(x)."""

    @staticmethod
    def forward(ctx, kind: int, bound: float, slope: float, y: Tensor, *params: Tensor):
        from . import ops

        _require_f32(y, *params)
        parts = [p.unsqueeze(-1) for p in params] if kind == 0 else list(params)
        batch = torch.broadcast_shapes(y.shape, *[p.shape[:-1] for p in parts])
        ye = y.expand(batch).contiguous()
        phi = _packed([p.detach() for p in parts], batch)
        sizes = [p.shape[-1] for p in parts]
        pieces = phi.split(sizes, -1)
        with torch.no_grad():
            if kind == 0:
                x = ops.affine_inverse(ye, pieces[0].squeeze(-1), pieces[1].squeeze(-1), slope)
            else:
                x = ops.rqs_inverse(ye, pieces[0], pieces[1], pieces[2], bound, slope)
        ctx.kind, ctx.bound, ctx.slope, ctx.sizes, ctx.yshape, ctx.pshapes = kind, bound, slope, sizes, y.shape, [p.shape for p in params]
        ctx.save_for_backward(x, phi)
        return x

    @staticmethod
    def backward(ctx, gx):
        x, phi = ctx.saved_tensors
        D = x.shape[-1] if x.dim() else 1
        N = x.numel() // max(D, 1)
        with torch.no_grad():
            _, ladj = _kernel_fwd(ctx.kind, ctx.bound, ctx.slope, False, x, phi, ctx.sizes)
        gxc = gx.expand(x.shape).contiguous()
        gy, seed = torch.empty_like(x), torch.empty_like(x)
        _C.check(_C.lib().zk_inverse_seed(x.numel(), _ptr(gxc), _ptr(ladj.contiguous()), _ptr(gy), _ptr(seed), _stream()), "zk_inverse_seed")
        scratch, gphi = torch.empty_like(x), torch.empty_like(phi)
        K = ctx.sizes[0] if ctx.kind == 1 else 0
        err = _C.lib().zk_univariate_backward(ctx.kind, N, D, K, ctx.bound, ctx.slope, _ptr(x), _ptr(phi), _ptr(seed), None, 0, _ptr(scratch), _ptr(gphi), _stream())
        _C.check(err, "zk_univariate_backward")
        pieces = gphi.split(ctx.sizes, -1)
        if ctx.kind == 0:
            pieces = [p.squeeze(-1) for p in pieces]
        grads = [_sum_to(p, s) for p, s in zip(pieces, ctx.pshapes)]
        return (None, None, None, _sum_to(gy, ctx.yshape), *grads)


BACKWARD_ACTS = (0, 1, 2, 3, 6, 7)  # activations whose derivative is a function of their output


class LinearFn(torch.autograd.Function):
    """y = act(x (mask * W)^T + b): forward = zk_linear, backward = act' kernel + library GEMMs."""

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
        wm = weight if mask is None else weight * mask
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.mm(g2, wm).reshape(x.shape)
        if ctx.needs_input_grad[1]:
            gw = torch.mm(g2.t(), x2)
            if mask is not None:
                gw = gw * mask
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(dim=0)
        return gx, gw, gb, None, None


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
    def backward(ctx, g):
        z, loc, scale = ctx.saved_tensors
        D = z.shape[-1]
        z2 = z.reshape(-1, D).contiguous()
        gz = torch.empty_like(z2)
        gc = g.expand(z.shape[:-1]).reshape(-1).contiguous()
        err = _C.lib().zk_diag_normal_backward(z2.shape[0], D, _ptr(z2), _ptr(loc.contiguous()), _ptr(scale.contiguous()), _ptr(gc), _ptr(gz), _stream())
        _C.check(err, "zk_diag_normal_backward")
        return gz.reshape(z.shape), None, None, (g if ctx.has_ladj else None)
