r"""Transform objects of the hot path, backed by HIP kernels.

API mirror of the classes `zuko.transforms` exposes for this path (constructor arguments,
`_call` / `_inverse` / `log_abs_det_jacobian` / `call_and_ladj` / `.inv`, broadcasting of
parameters against inputs), so that they can be handed to the reference's flow factories as
`univariate=` hooks or used through `zuko_amd.flows`.  Differences from the reference:

* parameters are kept UNCONSTRAINED inside the object; constraining (softclip / softmax / cumsum
  / exp) happens inside the kernel that also evaluates the transform (one launch per call);
* `call_and_ladj_reduced(x)` returns log|det J| already summed over the last axis — this is what
  `DependentTransform(t, 1)` asks for and lets the kernel do the feature reduction in-wave;
* tensors must be on a HIP device; there is no CPU path.

Reference lines are cited per class.
"""

from __future__ import annotations

from typing import Callable, Sequence

import torch
from torch import Tensor
from torch.distributions import Transform, constraints
from torch.distributions.utils import _sum_rightmost

from . import ops

__all__ = [
    "AdditiveTransform",
    "AutoregressiveTransform",
    "BernsteinTransform",
    "BoundedBernsteinTransform",
    "CircularShiftTransform",
    "ComposedTransform",
    "CouplingTransform",
    "DependentTransform",
    "MonotonicAffineTransform",
    "MonotonicRQSTransform",
    "SOSPolynomialTransform",
    "ShiftedSOSPolynomialTransform",
    "SoftclipTransform",
]


def _generic_call_and_ladj(self: Transform, x: Tensor):
    y = self(x)
    return y, self.log_abs_det_jacobian(x, y)


# zuko installs the same helper on torch's Transform base class (zuko/transforms.py:46-56) so that
# `t.inv.call_and_ladj` works for torch's own _InverseTransform; do likewise unless already present.
if not hasattr(Transform, "call_and_ladj"):
    Transform.call_and_ladj = _generic_call_and_ladj


class _Univariate(Transform):
    """Common surface of the elementwise monotone bijections R -> R."""

    domain = constraints.real
    codomain = constraints.real
    bijective = True
    sign = +1

    def _forward(self, x: Tensor, reduce: bool):
        raise NotImplementedError

    def _call(self, x: Tensor) -> Tensor:
        return self._forward(x, False)[0]

    def log_abs_det_jacobian(self, x: Tensor, y: Tensor) -> Tensor:
        return self._forward(x, False)[1]

    def call_and_ladj(self, x: Tensor):
        return self._forward(x, False)

    def call_and_ladj_reduced(self, x: Tensor):
        """(y, ladj summed over the last axis) in one kernel launch."""
        return self._forward(x, True)


class MonotonicAffineTransform(_Univariate):
    r"""y = exp(softclip(a)) * x + b.  Mirrors zuko/transforms.py:412-446."""

    def __init__(self, shift: Tensor, scale: Tensor, slope: float = 1e-3, **kwargs) -> None:
        super().__init__(**kwargs)
        self.shift = shift
        self.unconstrained_scale = scale
        self.slope = slope

    def _forward(self, x, reduce):
        return ops.affine_forward(x, self.shift, self.unconstrained_scale, self.slope, reduce, packed=getattr(self, "_packed", None))

    def _inverse(self, y: Tensor) -> Tensor:
        return ops.affine_inverse(y, self.shift, self.unconstrained_scale, self.slope)


class MonotonicRQSTransform(_Univariate):
    r"""Monotonic rational-quadratic spline on [-B, B], identity outside.
    Mirrors zuko/transforms.py:449-567 (widths[*, K], heights[*, K], derivatives[*, K-1])."""

    def __init__(self, widths: Tensor, heights: Tensor, derivatives: Tensor, bound: float = 5.0, slope: float = 1e-3, **kwargs) -> None:
        super().__init__(**kwargs)
        self.widths = widths
        self.heights = heights
        self.unconstrained_derivatives = derivatives
        self.bound = bound
        self.slope = slope

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(bins={self.bins})"

    @property
    def bins(self) -> int:
        return self.widths.shape[-1]

    def _forward(self, x, reduce):
        return ops.rqs_forward(x, self.widths, self.heights, self.unconstrained_derivatives, self.bound, self.slope, reduce, packed=getattr(self, "_packed", None))

    def _inverse(self, y: Tensor) -> Tensor:
        return ops.rqs_inverse(y, self.widths, self.heights, self.unconstrained_derivatives, self.bound, self.slope)

    def bin_index(self, x: Tensor) -> Tensor:
        """k = #(horizontal knots < x) - 1 as int32 (zuko/transforms.py:521-526)."""
        return ops.rqs_forward(x, self.widths, self.heights, self.unconstrained_derivatives, self.bound, self.slope, False, want_bins=True)[2]


class SOSPolynomialTransform(_Univariate):
    r"""f(x) = \int_0^x mean_k (1 + sum_j a_kj (u/10)^j)^2 + slope du, Gauss-Legendre with L+1 nodes;
    inverse by 25-step bisection on [-10, 10].  Mirrors zuko/transforms.py:927-963 (+ :878-924, :570-637)."""

    def __init__(self, a: Tensor, slope: float = 1e-3, **kwargs) -> None:
        super().__init__(**kwargs)
        self.a = a
        self.slope = slope
        self.constant = None

    def _forward(self, x, reduce):
        return ops.sos_forward(x, self.a, self.constant, self.slope, reduce)

    def _inverse(self, y: Tensor) -> Tensor:
        return ops.sos_inverse(y, self.a, self.constant, self.slope)


class ShiftedSOSPolynomialTransform(SOSPolynomialTransform):
    r"""SOS polynomial followed by `+ constant`, the univariate of SOSPF
    (zuko/flows/polynomial.py:23-29), evaluated in the same kernel."""

    def __init__(self, a: Tensor, constant: Tensor, slope: float = 1e-3, **kwargs) -> None:
        super().__init__(a, slope, **kwargs)
        self.constant = constant


class BernsteinTransform(_Univariate):
    r"""Monotone Bernstein polynomial with linear continuation outside [-B, B]; theta is the
    unconstrained [*, M] vector.  Mirrors zuko/transforms.py:640-777; the derivative the reference
    obtains through autograd (:623-637) is evaluated in closed form (de Casteljau)."""

    bounded = False

    def __init__(self, theta: Tensor, bound: float = 5.0, **kwargs) -> None:
        self.eps = float(kwargs.pop("eps", ops.BERN_EPS))  # MonotonicTransform's kwarg (zuko/transforms.py:594): continuation margin and bisection precision
        if not 0.0 < self.eps < 0.5:
            raise ValueError(f"zuko_amd: eps must lie in (0, 0.5), got {self.eps}")
        super().__init__(**kwargs)
        self.unconstrained_theta = theta
        self.bound = bound
        nc = theta.shape[-1] + (5 if self.bounded else 2)
        if nc > ops.BERN_NC_MAX:
            raise NotImplementedError(f"zuko_amd: Bernstein polynomials are built for up to {ops.BERN_NC_MAX} constrained coefficients, got {nc}")

    def _forward(self, x, reduce):
        return ops.bernstein_forward(x, self.unconstrained_theta, self.bounded, self.bound, reduce, eps=self.eps)

    def _inverse(self, y: Tensor) -> Tensor:
        return ops.bernstein_inverse(y, self.unconstrained_theta, self.bounded, self.bound, eps=self.eps)


class BoundedBernsteinTransform(BernsteinTransform):
    r"""Bernstein polynomial pinned to the identity at +-B.  Mirrors zuko/transforms.py:780-831."""

    bounded = True


# ------------------------------------------------------------------------------------------------
# glue transforms SOSPF composes in (zuko/transforms.py:286-316, 381-409); tiny, parameter-free
# ------------------------------------------------------------------------------------------------


class SoftclipTransform(Transform):
    r"""x / (1 + |x / B|).  Mirrors zuko/transforms.py:286-316 (left on device tensor ops, as
    SURVEY section 2 row 8 prescribes)."""

    bijective = True
    sign = +1

    def __init__(self, bound: float = 1.0, **kwargs) -> None:
        super().__init__(**kwargs)
        self.bound = bound
        self.domain = constraints.real
        self.codomain = constraints.interval(-bound, bound)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(bound={self.bound})"

    def _call(self, x):
        return x / (1 + abs(x / self.bound))

    def _inverse(self, y):
        return y / (1 - abs(y / self.bound))

    def log_abs_det_jacobian(self, x, y):
        return -2 * torch.log1p(abs(x / self.bound))


class CircularShiftTransform(Transform):
    r"""(x mod 2B) - B on [-B, B], zero log|det J|.  Mirrors zuko/transforms.py:319-351 (NCSF glue,
    one device `remainder`)."""

    bijective = True

    def __init__(self, bound: float = 1.0, **kwargs) -> None:
        super().__init__(**kwargs)
        self.bound = bound
        self.domain = constraints.interval(-bound, bound)
        self.codomain = constraints.interval(-bound, bound)

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(bound={self.bound})"

    def _call(self, x):
        return torch.remainder(x, 2 * self.bound) - self.bound

    def _inverse(self, y):
        return torch.remainder(y, 2 * self.bound) - self.bound

    def log_abs_det_jacobian(self, x, y):
        return torch.zeros_like(x)


class AdditiveTransform(Transform):
    r"""x + b.  Mirrors zuko/transforms.py:381-409."""

    domain = constraints.real
    codomain = constraints.real
    bijective = True
    sign = +1

    def __init__(self, shift: Tensor, **kwargs) -> None:
        super().__init__(**kwargs)
        self.shift = shift

    def _call(self, x):
        return x + self.shift

    def _inverse(self, y):
        return y - self.shift

    def log_abs_det_jacobian(self, x, y):
        return torch.zeros_like(x)


# ------------------------------------------------------------------------------------------------
# structural transforms
# ------------------------------------------------------------------------------------------------


class DependentTransform(Transform):
    r"""Treats the right-most `reinterpreted` dims of `base` as one event (sums ladj over them).
    Mirrors zuko/transforms.py:163-220."""

    def __init__(self, base: Transform, reinterpreted: int, **kwargs) -> None:
        super().__init__(**kwargs)
        self.base = base
        self.reinterpreted = reinterpreted

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self.base}, {self.reinterpreted})"

    @property
    def domain(self):
        return constraints.independent(self.base.domain, self.reinterpreted)

    @property
    def codomain(self):
        return constraints.independent(self.base.codomain, self.reinterpreted)

    @property
    def bijective(self) -> bool:
        return self.base.bijective

    def _call(self, x):
        return self.base(x)

    @property
    def inv(self) -> Transform:
        return DependentTransform(self.base.inv, self.reinterpreted)

    def _inverse(self, y):
        return self.base.inv(y)

    def log_abs_det_jacobian(self, x, y):
        return self.call_and_ladj(x)[1] if hasattr(self.base, "call_and_ladj_reduced") and self.reinterpreted == 1 else _sum_rightmost(
            self.base.log_abs_det_jacobian(x, y), self.reinterpreted
        )

    def call_and_ladj(self, x):
        if self.reinterpreted == 1 and hasattr(self.base, "call_and_ladj_reduced") and x.dim() >= 1:
            return self.base.call_and_ladj_reduced(x)
        y, ladj = self.base.call_and_ladj(x)
        return y, _sum_rightmost(ladj, self.reinterpreted)

    def forward_shape(self, shape):
        return self.base.forward_shape(shape)

    def inverse_shape(self, shape):
        return self.base.inverse_shape(shape)


class ComposedTransform(Transform):
    r"""f_n o ... o f_0 with accumulated log|det J|.  Mirrors zuko/transforms.py:59-160."""

    def __init__(self, *transforms: Transform, **kwargs) -> None:
        super().__init__(**kwargs)
        if not transforms:
            raise AssertionError("'transforms' cannot be empty")
        self.transforms = list(transforms)
        dim = 0
        for t in reversed(self.transforms):
            dim = t.domain.event_dim + max(dim - t.codomain.event_dim, 0)
        self.domain_dim = dim
        for t in self.transforms:
            dim += t.codomain.event_dim - t.domain.event_dim
        self.codomain_dim = dim

    def __repr__(self) -> str:
        body = "\n".join(f"  ({i}): " + repr(t).replace("\n", "\n  ") for i, t in enumerate(self.transforms))
        return f"{self.__class__.__name__}(\n{body}\n)"

    @staticmethod
    def _lift(c, extra: int):
        return constraints.independent(c, extra) if extra > 0 else c

    @property
    def domain(self):
        d = self.transforms[0].domain
        return self._lift(d, self.domain_dim - d.event_dim)

    @property
    def codomain(self):
        c = self.transforms[-1].codomain
        return self._lift(c, self.codomain_dim - c.event_dim)

    @property
    def bijective(self) -> bool:
        return all(t.bijective for t in self.transforms)

    def _call(self, x):
        for t in self.transforms:
            x = t(x)
        return x

    @property
    def inv(self) -> Transform:
        rev = ComposedTransform.__new__(ComposedTransform)
        Transform.__init__(rev)
        rev.transforms = [t.inv for t in reversed(self.transforms)]
        rev.domain_dim, rev.codomain_dim = self.codomain_dim, self.domain_dim
        return rev

    def _inverse(self, y):
        for t in reversed(self.transforms):
            y = t.inv(y)
        return y

    def log_abs_det_jacobian(self, x, y):
        return self.call_and_ladj(x)[1]

    def call_and_ladj(self, x):
        dim = self.domain_dim
        total = None
        owned = False  # `total` is a tensor this loop may write: allocated by a fused transform for this call, or the result of `total + ladj`
        for t in self.transforms:
            acc = getattr(t, "call_and_accumulate_ladj", None)
            y = acc(x, total) if (acc is not None and owned and dim == t.domain.event_dim) else None  # the kernel adds its log-determinant to `total`
            if y is not None:
                x = y
            else:
                x, ladj = t.call_and_ladj(x)
                ladj = _sum_rightmost(ladj, dim - t.domain.event_dim)
                owned = acc is not None if total is None else True
                total = ladj if total is None else total + ladj
            dim += t.codomain.event_dim - t.domain.event_dim
        return x, total

    def forward_shape(self, shape):
        for t in self.transforms:
            shape = t.forward_shape(shape)
        return shape

    def inverse_shape(self, shape):
        for t in reversed(self.transforms):
            shape = t.inverse_shape(shape)
        return shape


class AutoregressiveTransform(Transform):
    r"""y_i = f(x_i | x_<i) with `meta(x) -> Transform`; the inverse runs `passes` fixed-point sweeps
    from zeros.  Mirrors zuko/transforms.py:966-1007 (generic form; the fused conditioner+univariate
    kernel lives in zuko_amd.flows.autoregressive)."""

    domain = constraints.real_vector
    codomain = constraints.real_vector
    bijective = True

    def __init__(self, meta: Callable[[Tensor], Transform], passes: int, **kwargs) -> None:
        super().__init__(**kwargs)
        self.meta = meta
        self.passes = passes

    def _call(self, x):
        return self.meta(x)(x)

    def _inverse(self, y):
        x = torch.zeros_like(y)
        for _ in range(self.passes):
            x = self.meta(x).inv(y)
        return x

    def log_abs_det_jacobian(self, x, y):
        return self.meta(x).log_abs_det_jacobian(x, y)

    def call_and_ladj(self, x):
        return self.meta(x).call_and_ladj(x)


class CouplingTransform(Transform):
    r"""y_a = x_a, y_b = f(x_b | x_a); `mask` is True on the pass-through split.
    Mirrors zuko/transforms.py:1010-1073."""

    domain = constraints.real_vector
    codomain = constraints.real_vector
    bijective = True

    def __init__(self, meta: Callable[[Tensor], Transform], mask: Tensor, **kwargs) -> None:
        super().__init__(**kwargs)
        self.meta = meta
        self.idx_a = mask.nonzero().squeeze(-1)
        self.idx_b = (~mask).nonzero().squeeze(-1)

    def split(self, x):
        return x[..., self.idx_a], x[..., self.idx_b]

    def merge(self, a, b, shape):
        out = a.new_empty(shape)
        out[..., self.idx_a] = a
        out[..., self.idx_b] = b
        return out

    def _call(self, x):
        a, b = self.split(x)
        return self.merge(a, self.meta(a)(b), x.shape)

    def _inverse(self, y):
        a, b = self.split(y)
        return self.merge(a, self.meta(a).inv(b), y.shape)

    def log_abs_det_jacobian(self, x, y):
        a, b = self.split(x)
        return self.meta(a).log_abs_det_jacobian(b, self.split(y)[1])

    def call_and_ladj(self, x):
        a, b = self.split(x)
        yb, ladj = self.meta(a).call_and_ladj(b)
        return self.merge(a, yb, x.shape), ladj
