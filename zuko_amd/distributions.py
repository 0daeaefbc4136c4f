r"""`NormalizingFlow` and `DiagNormal`: the caller of the transform hot path.

Mirrors zuko/distributions.py:39-138 and :337-363.  `log_prob` feeds z and the accumulated
log|det J| to one HIP kernel that evaluates the diagonal-normal log-density, reduces over features
and adds the ladj (zk_diag_normal_log_prob) when the base is a `DiagNormal`.
"""

from __future__ import annotations

import torch
from torch import Size, Tensor
from torch.distributions import Distribution, Independent, Normal, Transform, Uniform
from torch.distributions.utils import _sum_rightmost

from . import ops

__all__ = ["BoxUniform", "DiagNormal", "NormalizingFlow"]

# the reference switches argument validation off globally (zuko/distributions.py:35-36); NaNs propagate
Distribution._validate_args = False
Distribution.arg_constraints = {}


class DiagNormal(Independent):
    """Independent(Normal(loc, scale), ndims)."""

    def __init__(self, loc: Tensor, scale: Tensor, ndims: int = 1) -> None:
        super().__init__(Normal(torch.as_tensor(loc), torch.as_tensor(scale)), ndims)

    def __repr__(self) -> str:
        return "Diag" + repr(self.base_dist)

    def expand(self, batch_shape: Size, new: Distribution | None = None) -> Distribution:
        new = self._get_checked_instance(DiagNormal, new)
        return super().expand(batch_shape, new)


class BoxUniform(Independent):
    """Independent(Uniform(lower, upper), ndims): the base of NCSF.  Mirrors zuko/distributions.py:366-396."""

    def __init__(self, lower: Tensor, upper: Tensor, ndims: int = 1) -> None:
        super().__init__(Uniform(torch.as_tensor(lower), torch.as_tensor(upper)), ndims)

    def __repr__(self) -> str:
        return "Box" + repr(self.base_dist)

    def expand(self, batch_shape: Size, new: Distribution | None = None) -> Distribution:
        new = self._get_checked_instance(BoxUniform, new)
        return super().expand(batch_shape, new)


class NormalizingFlow(Distribution):
    """p(X = x) = p(Z = f(x)) |det df/dx| for a transformation f and a base distribution p(Z)."""

    has_rsample = True

    def __init__(self, transform: Transform, base: Distribution) -> None:
        super().__init__()
        extra = transform.codomain.event_dim - len(base.event_shape)
        if extra > 0:
            base = Independent(base, extra)
        self.transform = transform
        self.base = base
        self.reinterpreted = max(-extra, 0)

    def __repr__(self) -> str:
        inner = f"(transform): {self.transform}\n(base): {self.base}".replace("\n", "\n  ")
        return f"{type(self).__name__}(\n  {inner}\n)"

    @property
    def batch_shape(self) -> Size:
        return self.base.batch_shape

    @property
    def event_shape(self) -> Size:
        return self.transform.inverse_shape(self.base.event_shape)

    def expand(self, batch_shape: Size, new: Distribution | None = None) -> Distribution:
        new = self._get_checked_instance(NormalizingFlow, new)
        new.transform = self.transform
        new.base = self.base.expand(batch_shape)
        new.reinterpreted = self.reinterpreted
        Distribution.__init__(new, validate_args=False)
        return new

    def _fusable_base(self):
        """(loc[D], scale[D]) if the base is a plain diagonal normal over the last axis."""
        b = self.base
        if isinstance(b, DiagNormal) and b.reinterpreted_batch_ndims == 1 and self.reinterpreted == 0:
            n = b.base_dist
            loc, scale = n.loc, n.scale
            D = loc.shape[-1]
            # an expanded base (context given) has stride-0 leading dims: take the underlying vector
            if loc.dim() > 1:
                if any(s != 0 for s in loc.stride()[:-1]) or any(s != 0 for s in scale.stride()[:-1]):
                    return None
                # (works for empty batch shapes too: view the last axis of the underlying vector)
                loc = loc.as_strided((D,), (loc.stride(-1),))
                scale = scale.as_strided((D,), (scale.stride(-1),))
            return loc, scale
        return None

    def log_prob(self, x: Tensor) -> Tensor:
        z, ladj = self.transform.call_and_ladj(x)
        ladj = _sum_rightmost(ladj, self.reinterpreted)
        fused = self._fusable_base()
        if fused is not None and torch.is_grad_enabled() and (fused[0].requires_grad or fused[1].requires_grad):
            fused = None  # trainable base: its gradient flows through torch's Normal.log_prob, as in the reference
        if fused is not None and z.is_cuda and torch.is_tensor(ladj) and ladj.shape == z.shape[:-1]:
            return ops.diag_normal_log_prob(z, fused[0], fused[1], ladj)
        return self.base.log_prob(z) + ladj

    def rsample(self, shape: Size = ()) -> Tensor:
        z = self.base.rsample(shape) if self.base.has_rsample else self.base.sample(shape)
        return self.transform.inv(z)

    def rsample_and_log_prob(self, shape: Size = ()):
        z = self.base.rsample(shape) if self.base.has_rsample else self.base.sample(shape)
        x, ladj = self.transform.inv.call_and_ladj(z)
        ladj = _sum_rightmost(ladj, self.reinterpreted)
        return x, self.base.log_prob(z) - ladj
