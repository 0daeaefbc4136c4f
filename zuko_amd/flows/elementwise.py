r"""Element-wise (features == 1 fallback) lazy transformation.

`MaskedAutoregressiveTransform` / `GeneralCouplingTransform` return this class when asked for a
single feature (zuko/flows/autoregressive.py:73-86, coupling.py).  Mirrors
zuko/flows/gaussianization.py:28-94: the parameters come from an MLP of the context, or are free
parameters when there is no context.
"""

from __future__ import annotations

from math import prod
from typing import Callable, Sequence

import torch
import torch.nn as nn
from torch import Size, Tensor
from torch.distributions import Transform

from ..lazy import LazyTransform
from ..nn import MLP
from ..transforms import DependentTransform, MonotonicAffineTransform
from ..utils import unpack


class ElementWiseTransform(LazyTransform):
    def __init__(
        self,
        features: int,
        context: int = 0,
        univariate: Callable[..., Transform] = MonotonicAffineTransform,
        shapes: Sequence[Size] = ((), ()),
        **kwargs,
    ) -> None:
        super().__init__()
        self.univariate = univariate
        self.shapes = shapes
        self.total = sum(prod(s) for s in shapes)
        self.features = features
        if context > 0:
            self.hyper = MLP(context, features * self.total, **kwargs)
        else:
            self.hyper = None
            self.phi = nn.ParameterList(torch.randn(features, *s) for s in shapes)

    def extra_repr(self) -> str:
        return f"(base): {getattr(self.univariate, '__name__', self.univariate)}"

    def forward(self, c: Tensor | None = None) -> Transform:
        if self.hyper is None:
            phi = tuple(self.phi)
        else:
            phi = unpack(self.hyper(c).unflatten(-1, (-1, self.total)), self.shapes)
        return DependentTransform(self.univariate(*phi), 1)
