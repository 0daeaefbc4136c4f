r"""Tensor-level wrappers around the C-ABI.  PyTorch is used for device memory, streams and
shape bookkeeping only; every arithmetic result comes out of a HIP kernel in libzuko_amd.so.

All functions require tensors on a HIP device (`tensor.is_cuda`) in float32 or float64 and raise
otherwise — there is no CPU or eager-PyTorch fallback.
"""

from __future__ import annotations

import ctypes
import math
from functools import lru_cache

import numpy as np
import torch
from torch import Tensor

from . import _C

ACTIVATIONS = {
    None: 0,
    torch.nn.Identity: 0,
    torch.nn.ReLU: 1,
    torch.nn.ELU: 2,
    torch.nn.Tanh: 3,
    torch.nn.SiLU: 4,
    torch.nn.GELU: 5,
    torch.nn.Sigmoid: 6,
    torch.nn.LeakyReLU: 7,
}


def _dtype_code(t: Tensor) -> int:
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.float64:
        return 1
    if t.dtype == torch.bfloat16:
        return 2  # storage type only (spline kernels, zk_linear_bf16): arithmetic and ladj are fp32
    raise TypeError(f"zuko_amd kernels support float32/float64 (and bfloat16 storage for splines / linear layers), got {t.dtype}")


def _require_device(*ts: Tensor) -> None:
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "zuko_amd: tensors must live on a HIP device (got a CPU tensor). "
                "This package has no CPU path; use the reference implementation for CPU work."
            )


def _bf16_trains_in_f32(fn):
    """bfloat16 operands under autograd.  The reference trains in whatever dtype the module is in (zuko tests/test_flows.py:17-29);
    the adjoint kernels here are float32, so a call that needs gradients runs on float32 copies of its bf16 operands (`.float()`
    is differentiable: gradients reach bf16 leaves through the casts, as torch.autocast would arrange it) and its transformed
    values (first result) are rounded back to bf16; log-determinants stay float32, as on the bf16 inference path."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kw):
        ts = [a for a in list(args) + list(kw.values()) if isinstance(a, Tensor)]
        if torch.is_grad_enabled() and any(t.dtype == torch.bfloat16 for t in ts) and any(t.requires_grad for t in ts):
            up = lambda a: a.float() if isinstance(a, Tensor) and a.dtype == torch.bfloat16 else a
            out = fn(*[up(a) for a in args], **{k: up(v) for k, v in kw.items()})
            if isinstance(out, tuple):
                return (out[0].to(torch.bfloat16),) + tuple(out[1:])
            return out.to(torch.bfloat16)
        return fn(*args, **kw)

    return wrapped


def _no_grad_only(*ts: Tensor) -> None:
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts):
        raise NotImplementedError(
            "zuko_amd: no backward kernel for this operation yet (autograd covers log_prob of affine / RQS "
            "flows, see zuko_amd/autograd.py); call it under torch.no_grad()."
        )


def _stream() -> int:
    return _C.stream()


def _ptr(t: Tensor | None):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _lead_collapse(t: Tensor) -> tuple[Tensor, int, int]:
    """View an expanded parameter tensor of shape (*lead, D, K) as strides (sN, sD) over a
    flattened leading index, copying only if the leading strides do not collapse."""
    if t.stride(-1) != 1 and t.shape[-1] != 1:
        t = t.contiguous()
    *lead, D, _ = t.shape
    sD = t.stride(-2) if D > 1 else 0
    dims = [(n, s) for n, s in zip(lead, t.stride()[: len(lead)]) if n > 1]
    if not dims:
        return t, 0, sD
    ok = True
    for (n0, s0), (n1, s1) in zip(dims[:-1], dims[1:]):
        if s0 != s1 * n1:
            ok = False
            break
    if ok:
        return t, dims[-1][1], sD
    t = t.contiguous()
    return t, t.stride(-3) if t.dim() >= 3 else 0, t.stride(-2) if D > 1 else 0


def _bshape(x: Tensor, *params: Tensor) -> torch.Size:
    """Broadcast of x against the parameters' batch shapes (everything but their last `k` dims)."""
    return torch.broadcast_shapes(x.shape, *[p for p in params])


class _Prepared:
    """x expanded to the broadcast shape as a contiguous [N, D] block + per-parameter strides."""

    def __init__(self, x: Tensor, params: list[tuple[Tensor, int]]):
        # params: (tensor, number of trailing "parameter" dims)
        _require_device(x, *[p for p, _ in params])
        _no_grad_only(x, *[p for p, _ in params])
        dt = x.dtype
        for p, _ in params:
            if p.dtype != dt or p.device != x.device:
                raise TypeError("zuko_amd: x and parameters must share dtype and device")
        batch = torch.broadcast_shapes(x.shape, *[p.shape[: p.dim() - k] for p, k in params])
        self.shape = batch
        if len(batch) == 0:
            shape2 = (1, 1)
        else:
            shape2 = (int(math.prod(batch[:-1])), int(batch[-1]))
        self.N, self.D = shape2
        self.x = x.expand(batch).contiguous()
        self.params = []
        for p, k in params:
            tail = p.shape[p.dim() - k :]
            flat = p.reshape(p.shape[: p.dim() - k] + (int(math.prod(tail)),)) if k != 1 else p
            pe = flat.expand(batch + flat.shape[-1:])
            if len(batch) == 0:
                pe = pe.reshape(1, 1, -1)
            elif len(batch) == 1:
                pe = pe.unsqueeze(0)
            t, sN, sD = _lead_collapse(pe)
            self.params.append((t, sN, sD))
        self.code = _dtype_code(x)

    def out(self) -> Tensor:
        return torch.empty(self.shape, dtype=self.x.dtype, device=self.x.device)

    def out_ladj(self, reduced: bool) -> Tensor:
        shape = self.shape[:-1] if reduced else self.shape
        dt = torch.float32 if self.x.dtype == torch.bfloat16 else self.x.dtype  # bf16 is a storage type: ladj stays fp32
        return torch.empty(shape, dtype=dt, device=self.x.device)


# ------------------------------------------------------------------------------------------------
# RQS
# ------------------------------------------------------------------------------------------------


def _packed_ok(x: Tensor, packed: Tensor | None, total: int) -> bool:
    return packed is not None and packed.dim() >= 2 and packed.shape[-1] == total and x.dim() >= 1 and tuple(torch.broadcast_shapes(x.shape, packed.shape[:-1])) == tuple(packed.shape[:-1])


@_bf16_trains_in_f32
def rqs_forward(x: Tensor, widths: Tensor, heights: Tensor, derivatives: Tensor, bound: float = 5.0, slope: float = 1e-3,
                reduce: bool = False, want_bins: bool = False, packed: Tensor | None = None):
    """(y, ladj[, k]) — see include/zuko_amd.h:zk_rqs_forward.  `packed`: the phi[..., D, 3K-1] tensor the three parameter
    views were cut from, when the caller has it (lets autograd differentiate phi itself instead of the three views)."""
    K = widths.shape[-1]
    if heights.shape[-1] != K or derivatives.shape[-1] != K - 1:
        raise ValueError("zuko_amd: widths/heights must have K entries and derivatives K-1")
    from . import autograd as AG

    if not want_bins and AG.needs_grad(x, widths, heights, derivatives):
        _require_device(x, widths, heights, derivatives)
        if K > 64:
            raise NotImplementedError("zuko_amd: spline backward is built for up to 64 bins")
        if _packed_ok(x, packed, 3 * K - 1):
            return AG.UnivariatePackedFn.apply((1, bound, slope, (K, K, K - 1), ()), reduce, x, packed)
        return AG.UnivariateFn.apply(1, bound, slope, reduce, x, widths, heights, derivatives)
    pr = _Prepared(x, [(widths, 1), (heights, 1), (derivatives, 1)])
    if reduce and len(pr.shape) == 0:
        raise ValueError("zuko_amd: cannot reduce ladj of a 0-d input")
    y, ladj = pr.out(), pr.out_ladj(reduce)
    bins = torch.empty(pr.shape, dtype=torch.int32, device=x.device) if want_bins else None
    (w, wn, wd), (h, hn, hd), (d, dn, dd) = pr.params
    err = _C.lib().zk_rqs_forward(pr.code, pr.N, pr.D, K, bound, slope, _ptr(pr.x), _ptr(w), wn, wd, _ptr(h), hn, hd, _ptr(d), dn, dd,
                                  _ptr(y), _ptr(ladj), int(reduce), _ptr(bins), _stream())
    _C.check(err, "zk_rqs_forward")
    return (y, ladj, bins) if want_bins else (y, ladj)


@_bf16_trains_in_f32
def rqs_inverse(y: Tensor, widths: Tensor, heights: Tensor, derivatives: Tensor, bound: float = 5.0, slope: float = 1e-3, want_bins: bool = False):
    K = widths.shape[-1]
    from . import autograd as AG

    if not want_bins and AG.needs_grad(y, widths, heights, derivatives):
        _require_device(y, widths, heights, derivatives)
        if K > 64:
            raise NotImplementedError("zuko_amd: spline backward is built for up to 64 bins")
        return AG.UnivariateInverseFn.apply(1, bound, slope, (), y, widths, heights, derivatives)
    pr = _Prepared(y, [(widths, 1), (heights, 1), (derivatives, 1)])
    x = pr.out()
    bins = torch.empty(pr.shape, dtype=torch.int32, device=y.device) if want_bins else None
    (w, wn, wd), (h, hn, hd), (d, dn, dd) = pr.params
    err = _C.lib().zk_rqs_inverse(pr.code, pr.N, pr.D, K, bound, slope, _ptr(pr.x), _ptr(w), wn, wd, _ptr(h), hn, hd, _ptr(d), dn, dd,
                                  _ptr(x), _ptr(bins), _stream())
    _C.check(err, "zk_rqs_inverse")
    return (x, bins) if want_bins else x


def rqs_diag(v: Tensor, widths: Tensor, heights: Tensor, derivatives: Tensor, bound: float = 5.0, slope: float = 1e-3, inverse: bool = False):
    """Diagnostic twin of rqs_forward / rqs_inverse (fp32): (out, ladj or None, k int32, knots [..., K+1]) where `knots`
    are the knots of the searched axis exactly as the product arithmetic compared them (zk_rqs_diag)."""
    K = widths.shape[-1]
    if v.dtype != torch.float32:
        raise TypeError("zuko_amd.rqs_diag: float32 only (the fp64 path searches IEEE knots: use rqs_forward(want_bins=True))")
    pr = _Prepared(v, [(widths, 1), (heights, 1), (derivatives, 1)])
    out = pr.out()
    ladj = None if inverse else pr.out_ladj(False)
    bins = torch.empty(pr.shape, dtype=torch.int32, device=v.device)
    knots = torch.empty(tuple(pr.shape) + (K + 1,), dtype=torch.float32, device=v.device)
    (w, wn, wd), (h, hn, hd), (d, dn, dd) = pr.params
    err = _C.lib().zk_rqs_diag(int(inverse), pr.N, pr.D, K, bound, slope, _ptr(pr.x), _ptr(w), wn, wd, _ptr(h), hn, hd, _ptr(d), dn, dd, _ptr(out), _ptr(ladj),
                               _ptr(bins), _ptr(knots), _stream())
    _C.check(err, "zk_rqs_diag")
    return out, ladj, bins, knots


def rqs_from_knots(v: Tensor, horizontal: Tensor, vertical: Tensor, slopes: Tensor, inverse: bool = False):
    """Test entry: evaluate from constrained knots; returns (out, ladj, k)."""
    K = horizontal.shape[-1] - 1
    pr = _Prepared(v, [(horizontal.contiguous(), 1), (vertical.contiguous(), 1), (slopes.contiguous(), 1)])
    out, ladj = pr.out(), pr.out_ladj(False)
    bins = torch.empty(pr.shape, dtype=torch.int32, device=v.device)
    (h, hn, hd), (ve, vn, vd), (s, sn, sd) = pr.params
    if (hn, hd) != (vn, vd) or (hn, hd) != (sn, sd):
        raise ValueError("zuko_amd: knots must share strides")
    err = _C.lib().zk_rqs_from_knots(pr.code, int(inverse), pr.N, pr.D, K, _ptr(pr.x), _ptr(h), _ptr(ve), _ptr(s), hn, hd, _ptr(out), _ptr(ladj),
                                     _ptr(bins), _stream())
    _C.check(err, "zk_rqs_from_knots")
    return out, ladj, bins


# ------------------------------------------------------------------------------------------------
# affine
# ------------------------------------------------------------------------------------------------


@_bf16_trains_in_f32
def affine_forward(x: Tensor, shift: Tensor, scale: Tensor, slope: float = 1e-3, reduce: bool = False, packed: Tensor | None = None):
    from . import autograd as AG

    if AG.needs_grad(x, shift, scale):
        _require_device(x, shift, scale)
        if _packed_ok(x, packed, 2):
            return AG.UnivariatePackedFn.apply((0, 5.0, slope, (1, 1), ()), reduce, x, packed)
        return AG.UnivariateFn.apply(0, 5.0, slope, reduce, x, shift, scale)
    pr = _Prepared(x, [(shift.unsqueeze(-1), 1), (scale.unsqueeze(-1), 1)])
    y, ladj = pr.out(), pr.out_ladj(reduce)
    (s, sn, sd), (c, cn, cd) = pr.params
    err = _C.lib().zk_affine_forward(pr.code, pr.N, pr.D, slope, _ptr(pr.x), _ptr(s), sn, sd, _ptr(c), cn, cd, _ptr(y), _ptr(ladj), int(reduce), _stream())
    _C.check(err, "zk_affine_forward")
    return y, ladj


@_bf16_trains_in_f32
def affine_inverse(y: Tensor, shift: Tensor, scale: Tensor, slope: float = 1e-3) -> Tensor:
    from . import autograd as AG

    if AG.needs_grad(y, shift, scale):
        _require_device(y, shift, scale)
        return AG.UnivariateInverseFn.apply(0, 5.0, slope, (), y, shift, scale)
    pr = _Prepared(y, [(shift.unsqueeze(-1), 1), (scale.unsqueeze(-1), 1)])
    x = pr.out()
    (s, sn, sd), (c, cn, cd) = pr.params
    err = _C.lib().zk_affine_inverse(pr.code, pr.N, pr.D, slope, _ptr(pr.x), _ptr(s), sn, sd, _ptr(c), cn, cd, _ptr(x), _stream())
    _C.check(err, "zk_affine_inverse")
    return x


# ------------------------------------------------------------------------------------------------
# SOS polynomial
# ------------------------------------------------------------------------------------------------


@lru_cache(maxsize=None)
def _leggauss01(n: int):
    """Gauss-Legendre nodes / weights mapped to [0, 1] (zuko/utils.py:328-347)."""
    nodes, weights = np.polynomial.legendre.leggauss(n)
    nodes = (nodes + 1) / 2
    weights = weights / 2
    return (ctypes.c_double * n)(*nodes.tolist()), (ctypes.c_double * n)(*weights.tolist())


SOS_BOUND = 10.0
SOS_EPS = 1e-6


@_bf16_trains_in_f32
def sos_forward(x: Tensor, a: Tensor, constant: Tensor | None = None, slope: float = 1e-3, reduce: bool = False):
    from . import autograd as AG

    if AG.needs_grad(x, a, constant):
        _require_device(x, a, constant)
        return AG.UnivariateFn.apply(2, SOS_BOUND, slope, reduce, x, a, constant)
    P, L1 = a.shape[-2:]
    plist = [(a, 2)] + ([(constant.unsqueeze(-1), 1)] if constant is not None else [])
    pr = _Prepared(x, plist)
    y, ladj = pr.out(), pr.out_ladj(reduce)
    nodes, weights = _leggauss01(L1)
    (at, an, ad) = pr.params[0]
    ct, cn, cd = pr.params[1] if constant is not None else (None, 0, 0)
    err = _C.lib().zk_sos_forward(pr.code, pr.N, pr.D, P, L1, slope, nodes, weights, _ptr(pr.x), _ptr(at), an, ad, _ptr(ct), cn, cd, _ptr(y), _ptr(ladj),
                                  int(reduce), _stream())
    _C.check(err, "zk_sos_forward")
    return y, ladj


@_bf16_trains_in_f32
def sos_inverse(y: Tensor, a: Tensor, constant: Tensor | None = None, slope: float = 1e-3) -> Tensor:
    from . import autograd as AG

    if AG.needs_grad(y, a, constant):
        _require_device(y, a, constant)
        return AG.UnivariateInverseFn.apply(2, SOS_BOUND, slope, (), y, a, constant)
    P, L1 = a.shape[-2:]
    plist = [(a, 2)] + ([(constant.unsqueeze(-1), 1)] if constant is not None else [])
    pr = _Prepared(y, plist)
    x = pr.out()
    nodes, weights = _leggauss01(L1)
    n_bisect = math.ceil(math.log2(2 * SOS_BOUND / SOS_EPS))  # transforms.py:615
    (at, an, ad) = pr.params[0]
    ct, cn, cd = pr.params[1] if constant is not None else (None, 0, 0)
    err = _C.lib().zk_sos_inverse(pr.code, pr.N, pr.D, P, L1, slope, nodes, weights, n_bisect, _ptr(pr.x), _ptr(at), an, ad, _ptr(ct), cn, cd, _ptr(x),
                                  _stream())
    _C.check(err, "zk_sos_inverse")
    return x


# ------------------------------------------------------------------------------------------------
# Bernstein polynomial
# ------------------------------------------------------------------------------------------------

BERN_EPS = 1e-6
BERN_NC_MAX = 72  # ZK_BERN_NCMAX of csrc/elementwise.hip


@_bf16_trains_in_f32
def bernstein_forward(x: Tensor, theta: Tensor, bounded: bool, bound: float = 5.0, reduce: bool = False, eps: float = BERN_EPS):
    """`eps`: margin (in [0, 1] coordinates) beyond which the polynomial is continued linearly (zuko/transforms.py:594, :742-760)."""
    from . import autograd as AG

    if AG.needs_grad(x, theta):
        _require_device(x, theta)
        return AG.BernsteinFn.apply((bool(bounded), float(eps)), bound, reduce, x, theta)
    M = theta.shape[-1]
    pr = _Prepared(x, [(theta, 1)])
    y, ladj = pr.out(), pr.out_ladj(reduce)
    (t, tn, td) = pr.params[0]
    err = _C.lib().zk_bernstein_forward(pr.code, pr.N, pr.D, M, int(bounded), bound, float(eps), _ptr(pr.x), _ptr(t), tn, td, _ptr(y), _ptr(ladj), int(reduce), _stream())
    _C.check(err, "zk_bernstein_forward")
    return y, ladj


@_bf16_trains_in_f32
def bernstein_inverse(y: Tensor, theta: Tensor, bounded: bool, bound: float = 5.0, eps: float = BERN_EPS) -> Tensor:
    """Bisection on [-B, B] to the precision `eps` of the reference (n = ceil(log2(2B / eps)) steps, zuko/transforms.py:609-617)."""
    from . import autograd as AG

    if AG.needs_grad(y, theta):
        _require_device(y, theta)
        return AG.UnivariateInverseFn.apply(3, bound, 0.0, (bool(bounded), float(eps)), y, theta)
    M = theta.shape[-1]
    pr = _Prepared(y, [(theta, 1)])
    x = pr.out()
    n_bisect = math.ceil(math.log2(2 * bound / eps))  # transforms.py:615
    (t, tn, td) = pr.params[0]
    err = _C.lib().zk_bernstein_inverse(pr.code, pr.N, pr.D, M, int(bounded), bound, float(eps), n_bisect, _ptr(pr.x), _ptr(t), tn, td, _ptr(x), _stream())
    _C.check(err, "zk_bernstein_inverse")
    return x


# ------------------------------------------------------------------------------------------------
# conditioner layer, base density, reduction
# ------------------------------------------------------------------------------------------------


@_bf16_trains_in_f32
def linear(x: Tensor, weight: Tensor, bias: Tensor | None = None, mask: Tensor | None = None, act: int = 0) -> Tensor:
    """act(x @ (mask * weight).T + bias) over the last dim of x (zuko/nn.py:217-218)."""
    _require_device(x, weight, bias, mask)
    from . import autograd as AG

    if AG.needs_grad(x, weight, bias):
        if act not in AG.BACKWARD_ACTS:
            raise NotImplementedError("zuko_amd: this activation cannot be fused when gradients are required")
        return AG.LinearFn.apply(x, weight, bias, mask, act)
    out_f, in_f = weight.shape
    if x.shape[-1] != in_f:
        raise ValueError(f"zuko_amd.linear: expected last dim {in_f}, got {x.shape[-1]}")
    if x.dtype == torch.bfloat16:
        w = weight if mask is None else weight * mask  # one pass over the parameters, not over the batch
        return linear_bf16(x, w, bias, None, act)
    x2 = x.reshape(-1, in_f)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    w = weight.contiguous()
    m = None
    if mask is not None:
        m = mask.contiguous()
        if m.dtype == torch.bool:
            m = m.view(torch.uint8)
        elif m.dtype != torch.uint8:
            raise TypeError("zuko_amd.linear: mask must be bool or uint8")
    b = None if bias is None else bias.contiguous()
    y = torch.empty((x2.shape[0], out_f), dtype=x.dtype, device=x.device)
    err = _C.lib().zk_linear(_dtype_code(x), x2.shape[0], in_f, out_f, _ptr(x2), x2.stride(0) if x2.shape[0] > 1 else in_f, _ptr(w), _ptr(m), _ptr(b), act,
                             _ptr(y), out_f, _stream())
    _C.check(err, "zk_linear")
    return y.reshape(x.shape[:-1] + (out_f,))


def linear_bf16(x: Tensor, weight_masked: Tensor, bias: Tensor | None, tile_live: Tensor | None, act: int = 0) -> Tensor:
    """bf16 act(x @ weight_masked.T + bias) with fp32 accumulation (zk_linear_bf16).  `tile_live`: int64
    [ceil(out/256)] bit masks (zuko_amd.nn.live_tile_masks), bit k clear where the 256 x 64 weight tile is entirely zero."""
    _require_device(x, weight_masked, bias, tile_live)
    _no_grad_only(x, weight_masked, bias)
    out_f, in_f = weight_masked.shape
    if x.dtype != torch.bfloat16 or weight_masked.dtype != torch.bfloat16 or (bias is not None and bias.dtype != torch.bfloat16):
        raise TypeError("zuko_amd.linear_bf16: x, weight and bias must be bfloat16")
    if in_f % 64 != 0:
        raise ValueError(f"zuko_amd.linear_bf16: in_features must be a multiple of 64 (got {in_f}); pad the conditioner input")
    if tile_live is not None and (tile_live.dtype != torch.int64 or tile_live.numel() != -(-out_f // 256) or in_f > 4096):
        raise ValueError("zuko_amd.linear_bf16: tile_live must be int64 [ceil(out / 256)] (and in_features <= 4096)")
    x2 = x.reshape(-1, in_f)
    if x2.stride(-1) != 1 or (x2.shape[0] > 1 and x2.stride(0) % 8 != 0) or x2.data_ptr() % 16 != 0:
        x2 = x2.contiguous()
    w = weight_masked.contiguous()
    b = None if bias is None else bias.contiguous()
    y = torch.empty((x2.shape[0], out_f), dtype=x.dtype, device=x.device)
    err = _C.lib().zk_linear_bf16(x2.shape[0], in_f, out_f, _ptr(x2), x2.stride(0) if x2.shape[0] > 1 else in_f, _ptr(w), _ptr(tile_live), _ptr(b), act,
                                  _ptr(y), out_f, _stream())
    _C.check(err, "zk_linear_bf16")
    return y.reshape(x.shape[:-1] + (out_f,))


def linear_bf16_rqs(h: Tensor, weight_panels: Tensor, bias_panels: Tensor | None, tile_live: Tensor | None, x: Tensor, K: int, bound: float = 5.0,
                    slope: float = 1e-3, lanes: bool = False):
    """(y bf16 [N, D], ladj fp32 [N]) of a bf16 autoregressive spline layer whose last conditioner layer and spline run
    in one kernel: h [N, in] last hidden activation, x [N, D] transform input, weights in feature panels — of 256 rows
    (zuko_amd.nn._Bf16Plan.spline_panels, zk_linear_bf16_rqs) or, with lanes=True, of 192 rows in the lane-owned order
    (spline_lane_panels, zk_linear_bf16_rqs_lanes)."""
    _require_device(h, weight_panels, bias_panels, tile_live, x)
    _no_grad_only(h, weight_panels, bias_panels, x)
    rows, in_f = weight_panels.shape
    panels = rows // (192 if lanes else 256)
    N, D = x.shape
    if h.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or weight_panels.dtype != torch.bfloat16:
        raise TypeError("zuko_amd.linear_bf16_rqs: bfloat16 tensors expected")
    h2 = h if (h.stride(-1) == 1 and h.stride(0) % 8 == 0 and h.data_ptr() % 16 == 0) else h.contiguous()
    x2 = x if x.stride(-1) == 1 else x.contiguous()
    y = torch.empty((N, D), dtype=torch.bfloat16, device=x.device)
    ladj = torch.empty(N, dtype=torch.float32, device=x.device)
    partial = torch.empty((2 * panels if lanes else panels, N), dtype=torch.float32, device=x.device)
    fn = _C.lib().zk_linear_bf16_rqs_lanes if lanes else _C.lib().zk_linear_bf16_rqs
    err = fn(N, in_f, panels, _ptr(h2), h2.stride(0) if N > 1 else in_f, _ptr(weight_panels), _ptr(tile_live), _ptr(bias_panels), K, D,
             bound, slope, _ptr(x2), x2.stride(0) if N > 1 else D, _ptr(y), D, _ptr(partial), _ptr(ladj), _stream())
    _C.check(err, "zk_linear_bf16_rqs_lanes" if lanes else "zk_linear_bf16_rqs")
    return y, ladj


def diag_normal_log_prob(z: Tensor, loc: Tensor, scale: Tensor, ladj: Tensor | None = None) -> Tensor:
    _require_device(z, loc, scale, ladj)
    from . import autograd as AG

    if AG.needs_grad(z, ladj):
        if z.dtype == torch.bfloat16:  # (bf16 module under autograd: float32 copies, gradients flow back through the casts)
            z, loc, scale = z.float(), loc.float(), scale.float()
            ladj = None if ladj is None else ladj.float()
        return AG.DiagNormalLogProbFn.apply(z, loc, scale, ladj)
    if z.dtype == torch.bfloat16:  # bf16 is a storage type here: the density is evaluated and returned in fp32
        z, loc, scale = z.float(), loc.float(), scale.float()
        ladj = None if ladj is None else ladj.float()
    D = z.shape[-1]
    z2 = z.reshape(-1, D).contiguous()
    out = torch.empty(z2.shape[0], dtype=z.dtype, device=z.device)
    la = None if ladj is None else ladj.expand(z.shape[:-1]).reshape(-1).contiguous()
    err = _C.lib().zk_diag_normal_log_prob(_dtype_code(z), z2.shape[0], D, _ptr(z2), _ptr(loc.contiguous()), _ptr(scale.contiguous()), _ptr(la), _ptr(out), _stream())
    _C.check(err, "zk_diag_normal_log_prob")
    return out.reshape(z.shape[:-1])


def sum_f64(v: Tensor, scale: float = 1.0) -> Tensor:
    """scale * sum(v) accumulated in float64; returns a 0-d float64 device tensor."""
    _require_device(v)
    v1 = v.reshape(-1).contiguous()
    ws = torch.empty(1024, dtype=torch.float64, device=v.device)
    out = torch.empty((), dtype=torch.float64, device=v.device)
    err = _C.lib().zk_sum_f64(_dtype_code(v), v1.numel(), _ptr(v1), scale, _ptr(ws), _ptr(out), _stream())
    _C.check(err, "zk_sum_f64")
    return out
