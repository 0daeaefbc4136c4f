r"""Neural spline flow (autoregressive by default; `passes=2` gives the coupling variant).
Mirrors zuko/flows/spline.py:21-62."""

from __future__ import annotations

from functools import partial

from ..transforms import MonotonicRQSTransform
from .autoregressive import MAF

__all__ = ["NSF"]


class NSF(MAF):
    def __init__(self, features: int, context: int = 0, bins: int = 8, slope: float = 1e-3, **kwargs) -> None:
        super().__init__(
            features=features,
            context=context,
            univariate=partial(MonotonicRQSTransform, slope=slope),
            shapes=[(bins,), (bins,), (bins - 1,)],
            **kwargs,
        )
