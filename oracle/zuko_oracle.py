r"""CPU oracle for the zuko transform hot path.  TEST INFRASTRUCTURE ONLY.

This file is a *functional restatement* of the arithmetic that probabilists/zuko
v1.6.0 performs for `flow(c).log_prob(x)` / `.inv` on the transforms named by
BASELINE.json:north_star.  It exists so that the HIP kernels in `zuko_amd/csrc`
have something to be compared against on a machine where `/root/reference` is not
mounted (the GPU box).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; the product package `zuko_amd`
never does, and has no CPU fallback of any kind.

Parity status: PINNED.  Every function below is checked, in the build container,
against the live reference (`PYTHONPATH=/root/reference`) by
`tests/golden/make_golden.py`, which also writes the committed fixtures under
`tests/golden/*.npz`; `tests/test_oracle_golden.py` re-checks the oracle against those
fixtures wherever the repo is checked out (bitwise for fp32/fp64 on the same torch
build, tolerance 1e-6 otherwise).

The reference is pure PyTorch, so the oracle is written with PyTorch *CPU* tensor
ops too: the same ATen kernels (MKL sgemm, Sleef exp/log) in the same order give
the same bits as the reference.  The structure is deliberately different from the
reference (plain functions over explicit arrays, no Transform/Module classes).

Each function cites the reference lines (relative to /root/reference) it follows.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Sequence

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

# --------------------------------------------------------------------------------------
# helpers (zuko/utils.py:212-244 broadcast, :596-622 unpack)
# --------------------------------------------------------------------------------------


def bcast(*ts: Tensor, ignore: int | Sequence[int] = 0) -> list[Tensor]:
    """Broadcast all but the trailing `ignore` dims (zuko/utils.py:212-244)."""
    if isinstance(ignore, int):
        ignore = [ignore] * len(ts)
    lead = [t.dim() - i for t, i in zip(ts, ignore)]
    common = torch.broadcast_shapes(*(t.shape[:n] for t, n in zip(ts, lead)))
    return [torch.broadcast_to(t, common + t.shape[n:]) for t, n in zip(ts, lead)]


def split_packed(phi: Tensor, shapes: Sequence[Sequence[int]]) -> tuple[Tensor, ...]:
    """Split the last dim of a packed parameter tensor (zuko/utils.py:596-622)."""
    sizes = [math.prod(s) for s in shapes]
    parts = phi.split(sizes, -1)
    return tuple(p.reshape(p.shape[:-1] + tuple(s)) for p, s in zip(parts, shapes))


# --------------------------------------------------------------------------------------
# monotonic affine (zuko/transforms.py:412-446)
# --------------------------------------------------------------------------------------


def affine_log_scale(scale: Tensor, slope: float = 1e-3) -> Tensor:
    """Soft-clipped log-scale, transforms.py:436."""
    return scale / (1 + abs(scale / math.log(slope)))


def affine_forward(shift: Tensor, scale: Tensor, x: Tensor, slope: float = 1e-3):
    """y = x*exp(a)+b and ladj = a (un-reduced). transforms.py:436-440, 445-446."""
    ls = affine_log_scale(scale, slope)
    y = x * ls.exp() + shift
    return y, ls.expand(x.shape)


def affine_inverse(shift: Tensor, scale: Tensor, y: Tensor, slope: float = 1e-3) -> Tensor:
    """x = (y-b)/exp(a). transforms.py:442-443."""
    ls = affine_log_scale(scale, slope)
    return (y - shift) / ls.exp()


# --------------------------------------------------------------------------------------
# monotonic rational-quadratic spline (zuko/transforms.py:449-567)
# --------------------------------------------------------------------------------------


def rqs_knots(widths: Tensor, heights: Tensor, derivatives: Tensor, bound: float = 5.0, slope: float = 1e-3):
    """Knot positions / slopes from unconstrained params. transforms.py:480-490.

    Returns (horizontal[*,K+1], vertical[*,K+1], slopes[*,K+1])."""
    ls = math.log(slope)
    w = widths / (1 + abs(2 * widths / ls))
    h = heights / (1 + abs(2 * heights / ls))
    d = derivatives / (1 + abs(derivatives / ls))
    w = F.pad(F.softmax(w, dim=-1), (1, 0), value=0)
    h = F.pad(F.softmax(h, dim=-1), (1, 0), value=0)
    d = F.pad(d, (1, 1), value=0)
    hor = bound * (2 * torch.cumsum(w, dim=-1) - 1)
    ver = bound * (2 * torch.cumsum(h, dim=-1) - 1)
    return hor, ver, torch.exp(d)


def rqs_bin_index(knots: Tensor, value: Tensor) -> Tensor:
    """k = #(knots < value) - 1, strict compare. transforms.py:521-526. int64, may be -1 or K."""
    return torch.sum(knots < value[..., None], dim=-1) - 1


def _rqs_gather(hor: Tensor, ver: Tensor, der: Tensor, k: Tensor):
    """Select the bin's corner values. transforms.py:499-519 (mask, k % K, gathers, s)."""
    K = hor.shape[-1] - 1
    inside = torch.logical_and(0 <= k, k < K)
    k = k % K
    kk = torch.stack((k, k + 1))
    kk, hs, vs, ds = bcast(kk[..., None], hor, ver, der, ignore=1)
    x0, x1 = hs.gather(-1, kk).squeeze(-1)
    y0, y1 = vs.gather(-1, kk).squeeze(-1)
    d0, d1 = ds.gather(-1, kk).squeeze(-1)
    s = (y1 - y0) / (x1 - x0)
    return inside, x0, x1, y0, y1, d0, d1, s


def rqs_forward_from_knots(hor: Tensor, ver: Tensor, der: Tensor, x: Tensor):
    """(y, ladj, k) given knots. transforms.py:554-567."""
    k = rqs_bin_index(hor, x)
    m, x0, x1, y0, y1, d0, d1, s = _rqs_gather(hor, ver, der, k)
    z = m * (x - x0) / (x1 - x0)
    y = y0 + (y1 - y0) * (s * z**2 + d0 * z * (1 - z)) / (s + (d0 + d1 - 2 * s) * z * (1 - z))
    jac = s**2 * (2 * s * z * (1 - z) + d0 * (1 - z) ** 2 + d1 * z**2) / (s + (d0 + d1 - 2 * s) * z * (1 - z)) ** 2
    return torch.where(m, y, x), m * jac.log(), k


def rqs_forward(widths, heights, derivatives, x, bound: float = 5.0, slope: float = 1e-3):
    """(y, ladj) from unconstrained params. transforms.py:469-490 + 554-567."""
    hor, ver, der = rqs_knots(widths, heights, derivatives, bound, slope)
    y, ladj, _ = rqs_forward_from_knots(hor, ver, der, x)
    return y, ladj


def rqs_inverse_from_knots(hor: Tensor, ver: Tensor, der: Tensor, y: Tensor):
    """(x, k) given knots; bin search runs on `vertical`. transforms.py:534-548."""
    k = rqs_bin_index(ver, y)
    m, x0, x1, y0, y1, d0, d1, s = _rqs_gather(hor, ver, der, k)
    y_ = m * (y - y0)
    a = (y1 - y0) * (s - d0) + y_ * (d0 + d1 - 2 * s)
    b = (y1 - y0) * d0 - y_ * (d0 + d1 - 2 * s)
    c = -s * y_
    z = 2 * c / (-b - (b**2 - 4 * a * c).sqrt())
    x = x0 + z * (x1 - x0)
    return torch.where(m, x, y), k


def rqs_inverse(widths, heights, derivatives, y, bound: float = 5.0, slope: float = 1e-3) -> Tensor:
    hor, ver, der = rqs_knots(widths, heights, derivatives, bound, slope)
    return rqs_inverse_from_knots(hor, ver, der, y)[0]


# --------------------------------------------------------------------------------------
# fixed-iteration bisection (zuko/utils.py:159-183) and Gauss-Legendre (utils.py:328-363)
# --------------------------------------------------------------------------------------


def bisect(f, y: Tensor, lo: float, hi: float, n: int) -> Tensor:
    """n halvings then midpoint; `f(c) < y` moves the lower end. utils.py:170-180."""
    a = torch.full_like(y, lo)
    b = torch.full_like(y, hi)
    for _ in range(n):
        c = (a + b) / 2
        below = f(c) < y
        a = torch.where(below, c, a)
        b = torch.where(below, b, c)
    return (a + b) / 2


def leggauss01(n: int, dtype, device="cpu"):
    """Nodes / weights on [0,1] from numpy.leggauss. utils.py:328-347."""
    nodes, weights = np.polynomial.legendre.leggauss(n)
    return (
        torch.as_tensor((nodes + 1) / 2, dtype=dtype, device=device),
        torch.as_tensor(weights / 2, dtype=dtype, device=device),
    )


def gl_quadrature(f, a: Tensor, b: Tensor, n: int) -> Tensor:
    """(b-a) * sum_i w_i f(lerp(a,b,node_i)). utils.py:349-363."""
    nodes, weights = leggauss01(n, a.dtype, a.device)
    pts = torch.lerp(a[..., None], b[..., None], nodes).movedim(-1, 0)
    return (b - a) * torch.tensordot(weights, f(pts), dims=1)


# --------------------------------------------------------------------------------------
# sum-of-squares polynomial (zuko/transforms.py:927-963 on top of :878-924 and :570-637)
# --------------------------------------------------------------------------------------

SOS_BOUND = 10.0  # MonotonicTransform default bound, transforms.py:593
SOS_EPS = 1e-6  # transforms.py:594


def sos_g(a: Tensor, x: Tensor, slope: float = 1e-3) -> Tensor:
    """Integrand mean_k (1 + sum_j a_kj (x/B)^j)^2 + slope. transforms.py:958-963."""
    u = x / SOS_BOUND
    pw = u[..., None] ** torch.arange(a.shape[-1], device=a.device)
    p = 1 + a @ pw[..., None]
    return p.squeeze(-1).square().mean(dim=-1) + slope


def sos_f(a: Tensor, x: Tensor, slope: float = 1e-3) -> Tensor:
    """f(x) = int_0^x g with n = L+1 nodes. transforms.py:911-918, :952."""
    return gl_quadrature(lambda t: sos_g(a, t, slope), torch.zeros_like(x), x, a.shape[-1])


def sos_forward(a: Tensor, x: Tensor, slope: float = 1e-3):
    """(f(x), log g(x)). transforms.py:923-924."""
    return sos_f(a, x, slope), sos_g(a, x, slope).log()


def sos_inverse(a: Tensor, y: Tensor, slope: float = 1e-3) -> Tensor:
    """25-step bisection on [-10, 10]. transforms.py:609-617."""
    n = math.ceil(math.log2(2 * SOS_BOUND / SOS_EPS))
    return bisect(lambda t: sos_f(a, t, slope), y, -SOS_BOUND, SOS_BOUND, n)


# --------------------------------------------------------------------------------------
# Bernstein polynomial (zuko/transforms.py:640-777) and bounded variant (:780-831)
# --------------------------------------------------------------------------------------

BERN_EPS = 1e-6


def bern_theta_unbounded(theta_unc: Tensor) -> Tensor:
    """softplus-diff cumsum minus shift, ends duplicated. transforms.py:703-727."""
    shift = math.log(2.0) * theta_unc.shape[-1] / 2
    first = theta_unc[..., :1]
    rest = theta_unc[..., 1:]
    rest = torch.cat((rest[..., :1], rest, rest[..., -1:]), dim=-1)
    diffs = torch.cat((first, F.softplus(rest)), dim=-1)
    return torch.cumsum(diffs, dim=-1) - shift


def bern_theta_bounded(theta_unc: Tensor, bound: float = 5.0) -> Tensor:
    """softmax-diff cumsum pinned to [-B, B] with unit end slopes. transforms.py:797-818."""
    lo = -bound * torch.ones_like(theta_unc[..., :1])
    edge = (2 * bound) / (theta_unc.shape[-1] + 4)
    diffs = F.softmax(theta_unc, dim=-1) * (2 * bound - 4 * edge)
    ones2 = edge * torch.ones_like(diffs[..., :2])
    return torch.cumsum(torch.cat((lo, ones2, diffs, ones2), dim=-1), dim=-1)


def _bern_basis(order: int, dtype, device):
    """Beta(i+1, M-i+1), i=0..M. transforms.py:729-734."""
    alpha = torch.arange(1, order + 2, dtype=dtype, device=device)
    beta = torch.arange(order + 1, 0, -1, dtype=dtype, device=device)
    return torch.distributions.Beta(alpha, beta, validate_args=False)


def _bern_poly(u: Tensor, theta: Tensor, basis) -> Tensor:
    """mean_i pdf_i(u) theta_i. transforms.py:736-740."""
    return torch.mean(basis.log_prob(u.unsqueeze(-1)).exp() * theta, dim=-1)


def _bern_tails(theta: Tensor, bounded: bool, bound: float, eps: float = None):
    eps = BERN_EPS if eps is None else eps
    """Offsets/slopes used for the linear extrapolation. transforms.py:685-701 / :820-831."""
    if bounded:
        off = (theta.new_tensor(-bound), theta.new_tensor(bound))
        slp = (theta.new_tensor(2 * bound), theta.new_tensor(2 * bound))
        return off, slp
    order = theta.shape[-1] - 1
    dtheta = order * (theta[..., 1:] - theta[..., :-1])
    basis = _bern_basis(order, theta.dtype, theta.device)
    dbasis = _bern_basis(order - 1, theta.dtype, theta.device)
    ends = [theta.new_tensor(eps), theta.new_tensor(1 - eps)]
    off = tuple(_bern_poly(e, theta, basis) for e in ends)
    slp = tuple(_bern_poly(e, dtheta, dbasis) for e in ends)
    return off, slp


def bern_f(theta: Tensor, x: Tensor, bounded: bool, bound: float = 5.0, eps: float = None) -> Tensor:
    """Polynomial inside (eps, 1-eps), straight lines outside. transforms.py:742-760 (eps: MonotonicTransform's kwarg, :594)."""
    eps = BERN_EPS if eps is None else eps
    basis = _bern_basis(theta.shape[-1] - 1, theta.dtype, theta.device)
    off, slp = _bern_tails(theta, bounded, bound, eps)
    u = (x + bound) / (2 * bound)
    lo = u <= eps
    hi = u >= 1 - eps
    safe = torch.where(lo | hi, 0.5 * torch.ones_like(u), u)
    y = _bern_poly(safe, theta, basis)
    y_lo = slp[0] * (u - eps) + off[0]
    y_hi = slp[1] * (u - 1 + eps) + off[1]
    y = torch.where(lo, y_lo, y)
    return torch.where(hi, y_hi, y)


def bern_constrain(theta_unc: Tensor, bounded: bool, bound: float = 5.0) -> Tensor:
    return bern_theta_bounded(theta_unc, bound) if bounded else bern_theta_unbounded(theta_unc)


def bern_forward(theta_unc: Tensor, x: Tensor, bounded: bool, bound: float = 5.0, eps: float = None):
    """(y, ladj); ladj is log of the autograd derivative, as transforms.py:623-637 does."""
    theta = bern_constrain(theta_unc.detach(), bounded, bound)
    with torch.enable_grad():
        xr = x.detach().clone().requires_grad_()
        y = bern_f(theta, xr, bounded, bound, eps)
        (jac,) = torch.autograd.grad(y, xr, torch.ones_like(y))
    return y.detach(), jac.log()


def bern_inverse(theta_unc: Tensor, y: Tensor, bounded: bool, bound: float = 5.0, eps: float = None) -> Tensor:
    """ceil(log2(2B / eps))-step (24 at the defaults) bisection on [-B, B] + closed-form tails. transforms.py:762-777, :609-617."""
    eps = BERN_EPS if eps is None else eps
    theta = bern_constrain(theta_unc, bounded, bound)
    off, slp = _bern_tails(theta, bounded, bound, eps)
    n = math.ceil(math.log2(2 * bound / eps))
    x = bisect(lambda t: bern_f(theta, t, bounded, bound, eps), y, -bound, bound, n)
    x_lo = ((y - off[0]) / slp[0] + eps) * 2 * bound - bound
    x_hi = ((y - off[1]) / slp[1] - eps + 1) * 2 * bound - bound
    x = torch.where(y <= off[0], x_lo, x)
    return torch.where(y >= off[1], x_hi, x)


# --------------------------------------------------------------------------------------
# conditioner networks (zuko/nn.py:202-318 masked; :13-22, :122-192 dense)
# --------------------------------------------------------------------------------------


def masked_mlp_masks(adjacency: Tensor, hidden: Sequence[int]) -> list[Tensor]:
    """Per-layer boolean masks [out,in] of the masked MLP. nn.py:265-295 (residual=False)."""
    uniq, inverse = torch.unique(adjacency, dim=0, return_inverse=True)
    precedence = uniq.double() @ uniq.double().t() == uniq.sum(dim=-1)
    masks: list[Tensor] = []
    idx = None
    for i, width in enumerate((*hidden, adjacency.shape[0])):
        m = uniq if i == 0 else precedence[:, idx]
        if (~m).all():
            raise ValueError("The adjacency matrix leads to a null Jacobian.")
        if i < len(hidden):
            reachable = m.sum(dim=-1).nonzero().squeeze(dim=-1)
            idx = reachable[torch.arange(width) % len(reachable)]
            m = m[idx]
        else:
            m = m[inverse]
        masks.append(m)
    return masks


def ar_adjacency(features: int, context: int, total: int, order: Tensor | None = None, passes: int | None = None):
    """Output-row x input adjacency of an autoregressive conditioner.
    flows/autoregressive.py:111-149.  Returns (adjacency[features*total, features+context], order, passes)."""
    if passes is None:
        passes = features
    if order is None:
        order = torch.arange(features)
    order = torch.as_tensor(order, dtype=int)
    passes = min(max(passes, 1), features)
    order = torch.div(order, math.ceil(features / passes), rounding_mode="floor")
    adj = order[:, None] > order
    if context > 0:
        adj = torch.cat((adj, torch.ones((features, context), dtype=bool)), dim=1)
    return torch.repeat_interleave(adj, repeats=total, dim=0), order, passes


def mlp_forward(x: Tensor, weights, biases, masks=None, act=torch.relu, plan=None) -> Tensor:
    """Linear stack; with masks: F.linear(x, mask*W, b) per layer (nn.py:217-218), else dense (nn.py:13-15).

    `plan` (residual=True networks, nn.py:297-309): a list of ("lin", i) / ("res", i, j) / ("act",) steps over
    the flat parameter lists; "res" is x + lin_j(act(lin_i(x))) (nn.py:195-199)."""
    def lin(i, v):
        W = weights[i] if masks is None else masks[i] * weights[i]
        return F.linear(v, W, biases[i])

    if plan is None:
        n = len(weights)
        for i in range(n):
            x = lin(i, x)
            if i + 1 < n:
                x = act(x)
        return x
    for step in plan:
        if step[0] == "lin":
            x = lin(step[1], x)
        elif step[0] == "res":
            x = x + lin(step[2], act(lin(step[1], x)))
        else:
            x = act(x)
    return x


# --------------------------------------------------------------------------------------
# flow descriptions: a list of layers, each a small dataclass, evaluated functionally
# --------------------------------------------------------------------------------------


@dataclass
class Univariate:
    """Which elementwise bijection a layer applies and how its packed params split."""

    kind: str  # 'affine' | 'rqs' | 'sos' | 'bernstein' | 'bbernstein'
    shapes: tuple  # e.g. ((8,), (8,), (7,))
    slope: float = 1e-3
    bound: float = 5.0

    @property
    def total(self) -> int:
        return sum(math.prod(s) for s in self.shapes)


def univariate_forward(u: Univariate, phi: Tensor, x: Tensor):
    """phi[..., D, total], x[..., D] -> (y[..., D], ladj[..., D])."""
    p = split_packed(phi, u.shapes)
    if u.kind == "affine":
        return affine_forward(p[0], p[1], x, u.slope)
    if u.kind == "rqs":
        return rqs_forward(p[0], p[1], p[2], x, u.bound, u.slope)
    if u.kind == "crqs":  # flows/spline.py:65-72: circular shift, then RQS on [-pi, pi]
        xs = torch.remainder(x, 2 * math.pi) - math.pi
        return rqs_forward(p[0], p[1], p[2], xs, math.pi, u.slope)
    if u.kind == "sos":  # flows/polynomial.py:23-29: SOS then + constant
        y, l = sos_forward(p[0], x, u.slope)
        return y + p[1], l
    if u.kind in ("bernstein", "bbernstein"):
        return bern_forward(p[0], x, u.kind == "bbernstein", u.bound)
    raise ValueError(u.kind)


def univariate_inverse(u: Univariate, phi: Tensor, y: Tensor) -> Tensor:
    p = split_packed(phi, u.shapes)
    if u.kind == "affine":
        return affine_inverse(p[0], p[1], y, u.slope)
    if u.kind == "rqs":
        return rqs_inverse(p[0], p[1], p[2], y, u.bound, u.slope)
    if u.kind == "crqs":
        xs = rqs_inverse(p[0], p[1], p[2], y, math.pi, u.slope)
        return torch.remainder(xs, 2 * math.pi) - math.pi
    if u.kind == "sos":
        return sos_inverse(p[0], y - p[1], u.slope)
    if u.kind in ("bernstein", "bbernstein"):
        return bern_inverse(p[0], y, u.kind == "bbernstein", u.bound)
    raise ValueError(u.kind)


@dataclass
class ARLayer:
    """One masked autoregressive transform (flows/autoregressive.py:24-218)."""

    uni: Univariate
    weights: list
    biases: list
    masks: list
    passes: int
    features: int
    plan: list | None = None  # module sequence for residual conditioners


@dataclass
class CouplingLayer:
    """One coupling transform (flows/coupling.py:25-139; transforms.py:1010-1073)."""

    uni: Univariate
    weights: list
    biases: list
    mask: Tensor  # True = passed through unchanged (x_a)


@dataclass
class SoftclipLayer:
    """x / (1 + |x / B|), SOSPF glue (transforms.py:286-316, flows/polynomial.py:73-76)."""

    bound: float = 11.0


def _ar_phi(layer: ARLayer, x: Tensor, c: Tensor | None) -> Tensor:
    """Conditioner call + unflatten. flows/autoregressive.py:207-213."""
    inp = x if c is None else torch.cat(bcast(x, c, ignore=1), dim=-1)
    phi = mlp_forward(inp, layer.weights, layer.biases, layer.masks, plan=layer.plan)
    return phi.unflatten(-1, (-1, layer.uni.total))


def ar_forward(layer: ARLayer, x: Tensor, c: Tensor | None = None):
    """(y[...,D], ladj[...]) — ladj summed over features. transforms.py:1005-1007, :210-214."""
    y, ladj = univariate_forward(layer.uni, _ar_phi(layer, x, c), x)
    return y, ladj.sum(dim=-1)


def ar_inverse(layer: ARLayer, y: Tensor, c: Tensor | None = None) -> Tensor:
    """`passes` fixed-point sweeps from x=0. transforms.py:994-1000."""
    x = torch.zeros_like(y)
    for _ in range(layer.passes):
        x = univariate_inverse(layer.uni, _ar_phi(layer, x, c), y)
    return x


def _coupling_phi(layer: CouplingLayer, xa: Tensor, c: Tensor | None) -> Tensor:
    inp = xa if c is None else torch.cat(bcast(xa, c, ignore=1), dim=-1)
    phi = mlp_forward(inp, layer.weights, layer.biases, None)
    return phi.unflatten(-1, (-1, layer.uni.total))


def coupling_forward(layer: CouplingLayer, x: Tensor, c: Tensor | None = None):
    """transforms.py:1068-1073 with split/merge :1040-1048."""
    ia = layer.mask.nonzero().squeeze(-1)
    ib = (~layer.mask).nonzero().squeeze(-1)
    xa, xb = x[..., ia], x[..., ib]
    yb, ladj = univariate_forward(layer.uni, _coupling_phi(layer, xa, c), xb)
    y = x.new_empty(x.shape)
    y[..., ia] = xa
    y[..., ib] = yb
    return y, ladj.sum(dim=-1)


def coupling_inverse(layer: CouplingLayer, y: Tensor, c: Tensor | None = None) -> Tensor:
    """transforms.py:1056-1060."""
    ia = layer.mask.nonzero().squeeze(-1)
    ib = (~layer.mask).nonzero().squeeze(-1)
    ya, yb = y[..., ia], y[..., ib]
    xb = univariate_inverse(layer.uni, _coupling_phi(layer, ya, c), yb)
    x = y.new_empty(y.shape)
    x[..., ia] = ya
    x[..., ib] = xb
    return x


def layer_forward(layer, x, c=None):
    if isinstance(layer, ARLayer):
        return ar_forward(layer, x, c)
    if isinstance(layer, CouplingLayer):
        return coupling_forward(layer, x, c)
    if isinstance(layer, SoftclipLayer):
        y = x / (1 + abs(x / layer.bound))
        return y, (-2 * torch.log1p(abs(x / layer.bound))).sum(dim=-1)
    raise TypeError(type(layer))


def layer_inverse(layer, y, c=None):
    if isinstance(layer, ARLayer):
        return ar_inverse(layer, y, c)
    if isinstance(layer, CouplingLayer):
        return coupling_inverse(layer, y, c)
    if isinstance(layer, SoftclipLayer):
        return y / (1 - abs(y / layer.bound))
    raise TypeError(type(layer))


def diag_normal_log_prob(z: Tensor, loc: Tensor, scale: Tensor) -> Tensor:
    """Independent(Normal).log_prob as torch/distributions/normal.py computes it,
    summed over the last dim (zuko/distributions.py:337-363)."""
    var = scale**2
    lp = -((z - loc) ** 2) / (2 * var) - scale.log() - math.log(math.sqrt(2 * math.pi))
    return lp.sum(dim=-1)


def box_uniform_log_prob(z: Tensor, lower: Tensor, upper: Tensor) -> Tensor:
    """Independent(Uniform(lower, upper)).log_prob as torch/distributions/uniform.py computes it
    (zuko/distributions.py:366-396): log(1[l <= z < u]) - log(u - l), summed over the last dim."""
    inside = lower.le(z).type_as(lower) * upper.gt(z).type_as(lower)
    return (torch.log(inside) - torch.log(upper - lower)).sum(dim=-1)


@dataclass
class FlowSpec:
    """A composed flow with a diagonal-normal (or box-uniform, NCSF) base (lazy.py:131-172, distributions.py:39-138)."""

    layers: list
    loc: Tensor
    scale: Tensor
    meta: dict = field(default_factory=dict)
    box: tuple | None = None  # (lower, upper) when the base is a BoxUniform


def flow_forward(spec: FlowSpec, x: Tensor, c: Tensor | None = None):
    """z, total ladj.  transforms.py:141-150."""
    acc = 0
    for layer in spec.layers:
        x, ladj = layer_forward(layer, x, c)
        acc = acc + ladj
    return x, acc


def flow_log_prob(spec: FlowSpec, x: Tensor, c: Tensor | None = None) -> Tensor:
    """base.log_prob(f(x)) + ladj.  distributions.py:115-119."""
    z, ladj = flow_forward(spec, x, c)
    if spec.box is not None:
        return box_uniform_log_prob(z, *spec.box) + ladj
    return diag_normal_log_prob(z, spec.loc, spec.scale) + ladj


def flow_inverse(spec: FlowSpec, z: Tensor, c: Tensor | None = None) -> Tensor:
    """x = f^{-1}(z): layers reversed. transforms.py:132-135."""
    for layer in reversed(spec.layers):
        z = layer_inverse(layer, z, c)
    return z


# --------------------------------------------------------------------------------------
# building a FlowSpec from a state_dict (key layout of the reference's module tree:
# transform.transforms.{i}.hyper.{0,2,4,..}.{weight,bias,mask}, .order / .mask, base.loc/.scale)
# --------------------------------------------------------------------------------------

UNI_AFFINE = Univariate("affine", ((), ()))


def uni_rqs(bins: int = 8, slope: float = 1e-3) -> Univariate:
    return Univariate("rqs", ((bins,), (bins,), (bins - 1,)), slope=slope)


def uni_crqs(bins: int = 8, slope: float = 1e-3) -> Univariate:
    return Univariate("crqs", ((bins,), (bins,), (bins - 1,)), slope=slope)


def uni_sos(degree: int = 4, polynomials: int = 3, slope: float = 1e-3) -> Univariate:
    return Univariate("sos", ((polynomials, degree + 1), ()), slope=slope)


def uni_bpf(degree: int = 16) -> Univariate:
    return Univariate("bbernstein", ((degree + 1,),))


def spec_from_state_dict(sd: dict, kind: str, uni: Univariate, features: int, passes: int | None = None, softclip: float | None = None) -> FlowSpec:
    """kind: 'ar' (MAF/NSF/SOSPF/BPF) or 'coupling' (NICE/RealNVP)."""
    idx = sorted({int(k.split(".")[2]) for k in sd if k.startswith("transform.transforms.") and ".hyper." in k})
    layers = []
    for n, i in enumerate(idx):
        pre = f"transform.transforms.{i}.hyper."
        mods = sorted({int(k[len(pre) :].split(".")[0]) for k in sd if k.startswith(pre)})
        W, b, M, plan = [], [], [], []
        residual = any(f"{pre}{j}.0.weight" in sd for j in mods)
        prev = None
        for j in mods:
            if prev is not None and j - prev == 2:
                plan.append(("act",))  # a parameter-free activation module sits between the two
            prev = j
            if f"{pre}{j}.weight" in sd:
                names = [f"{pre}{j}"]
                plan.append(("lin", len(W)))
            else:
                names = [f"{pre}{j}.0", f"{pre}{j}.2"]
                plan.append(("res", len(W), len(W) + 1))
            for nm in names:
                W.append(sd[nm + ".weight"])
                b.append(sd[nm + ".bias"])
                M.append(sd.get(nm + ".mask"))
        if kind == "ar":
            layers.append(ARLayer(uni, W, b, M, passes if passes is not None else features, features, plan if residual else None))
        else:
            layers.append(CouplingLayer(uni, W, b, sd[f"transform.transforms.{i}.mask"]))
        if softclip is not None and n + 1 < len(idx):
            layers.append(SoftclipLayer(softclip))
    if "base.loc" in sd:
        return FlowSpec(layers, sd["base.loc"], sd["base.scale"])
    return FlowSpec(layers, None, None, box=(sd["base.lower"], sd["base.upper"]))
