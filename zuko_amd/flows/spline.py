r"""Neural spline flow (autoregressive by default; `passes=2` gives the coupling variant).
Mirrors zuko/flows/spline.py:21-62."""

from __future__ import annotations

from functools import partial
from math import pi

import torch

from ..distributions import BoxUniform
from ..lazy import UnconditionalDistribution
from ..transforms import CircularShiftTransform, ComposedTransform, MonotonicRQSTransform
from .autoregressive import MAF

__all__ = ["NCSF", "NSF"]


class NSF(MAF):
    def __init__(self, features: int, context: int = 0, bins: int = 8, slope: float = 1e-3, **kwargs) -> None:
        super().__init__(
            features=features,
            context=context,
            univariate=partial(MonotonicRQSTransform, slope=slope),
            shapes=[(bins,), (bins,), (bins - 1,)],
            **kwargs,
        )


def CircularRQSTransform(*phi, slope: float = 1e-3):
    """Circular shift followed by an RQS on [-pi, pi] (zuko/flows/spline.py:65-72)."""
    return ComposedTransform(CircularShiftTransform(bound=pi), MonotonicRQSTransform(*phi, bound=pi, slope=slope))


class NCSF(MAF):
    r"""Neural circular spline flow: features live in [-pi, pi[, base = BoxUniform.
    Mirrors zuko/flows/spline.py:74-117."""

    def __init__(self, features: int, context: int = 0, bins: int = 8, slope: float = 1e-3, **kwargs) -> None:
        super().__init__(
            features=features,
            context=context,
            univariate=partial(CircularRQSTransform, slope=slope),
            shapes=[(bins,), (bins,), (bins - 1,)],
            **kwargs,
        )
        self.base = UnconditionalDistribution(
            BoxUniform,
            lower=torch.full((features,), -pi - 1e-5),
            upper=torch.full((features,), pi + 1e-5),
            buffer=True,
        )
