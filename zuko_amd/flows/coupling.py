r"""Coupling transformations, NICE and RealNVP.

API / module-tree mirror of zuko/flows/coupling.py:25-200: dense `zuko_amd.nn.MLP` conditioner
(`hyper.{0,2,...}.{weight,bias}`), boolean buffer `mask` (True = pass-through half).
"""

from __future__ import annotations

from functools import partial
from math import prod
from typing import Callable, Sequence

import torch
from torch import BoolTensor, Size, Tensor
from torch.distributions import Transform

from ..distributions import DiagNormal
from ..lazy import Flow, LazyTransform, UnconditionalDistribution
from ..nn import MLP
from ..transforms import CouplingTransform, DependentTransform, MonotonicAffineTransform, MonotonicRQSTransform
from ..utils import broadcast, unpack
from .autoregressive import _univariate_name
from .elementwise import ElementWiseTransform

__all__ = ["NICE", "GeneralCouplingTransform", "RealNVP"]


class GeneralCouplingTransform(LazyTransform):
    r"""Lazy coupling transformation y_a = x_a, y_b = f(x_b | x_a, c)."""

    def __new__(cls, features: int | None = None, context: int = 0, mask=None, *args, **kwargs):
        if features is None or features > 1:
            return super().__new__(cls)
        return ElementWiseTransform(features, context, *args, **kwargs)

    def __init__(
        self,
        features: int,
        context: int = 0,
        mask: BoolTensor | None = None,
        univariate: Callable[..., Transform] = MonotonicAffineTransform,
        shapes: Sequence[Size] = ((), ()),
        **kwargs,
    ) -> None:
        super().__init__()
        self.univariate = univariate
        self.shapes = shapes
        self.total = sum(prod(s) for s in shapes)

        mask = (torch.arange(features) % 2 == 1) if mask is None else torch.as_tensor(mask, dtype=bool)
        assert mask.ndim == 1, "'mask' should be a vector."
        assert mask.shape[0] == features, f"'mask' should have {features} elements."
        kept = int(mask.sum())
        moved = features - kept
        assert kept > 0
        assert moved > 0
        self.register_buffer("mask", mask)
        self.hyper = MLP(kept + context, moved * self.total, **kwargs)

    def extra_repr(self) -> str:
        m = self.mask.int().tolist()
        text = str(m) if len(m) <= 10 else "[" + ", ".join(map(str, m[:5])) + ", ..., " + ", ".join(map(str, m[-5:])) + "]"
        return f"(base): {_univariate_name(self.univariate)}\n(mask): {text}"

    def meta(self, c: Tensor | None, x: Tensor) -> Transform:
        if c is not None:
            x = torch.cat(broadcast(x, c, ignore=1), dim=-1)
        phi = self.hyper(x).unflatten(-1, (-1, self.total))
        u = self.univariate(*unpack(phi, self.shapes))
        if type(u) in (MonotonicAffineTransform, MonotonicRQSTransform):
            u._packed = phi  # the views above were cut from this tensor: autograd can differentiate it directly
        return DependentTransform(u, 1)

    def forward(self, c: Tensor | None = None) -> Transform:
        return CouplingTransform(partial(self.meta, c), self.mask)


class NICE(Flow):
    r"""NICE / RealNVP: `transforms` coupling layers with alternating checkered (or random) masks,
    affine univariates by default.  Mirrors zuko/flows/coupling.py:142-196."""

    def __init__(self, features: int, context: int = 0, transforms: int = 3, randmask: bool = False, **kwargs) -> None:
        layers = []
        for i in range(transforms):
            positions = torch.randperm(features) if randmask else torch.arange(features)
            layers.append(GeneralCouplingTransform(features=features, context=context, mask=positions % 2 == i % 2, **kwargs))
        base = UnconditionalDistribution(DiagNormal, loc=torch.zeros(features), scale=torch.ones(features), buffer=True)
        super().__init__(layers, base)


class RealNVP(NICE):
    pass
