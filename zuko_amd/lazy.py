r"""Lazy (module -> object) layer: modules that build a Transform / Distribution from a context.

API mirror of zuko/lazy.py:25-172, 242-335 (LazyDistribution, LazyTransform, LazyInverse,
LazyComposedTransform, Flow, UnconditionalDistribution, UnconditionalTransform).  No arithmetic
happens here; module/attribute names are kept so state_dict keys line up with the reference
(`transform.transforms.{i}...`, `base.loc`, `base.scale`).
"""

from __future__ import annotations

import abc
from typing import Callable, Sequence

import torch.nn as nn
from torch import Tensor
from torch.distributions import Distribution, Transform

from .distributions import NormalizingFlow
from .transforms import ComposedTransform
from .utils import Partial

__all__ = [
    "Flow",
    "LazyComposedTransform",
    "LazyDistribution",
    "LazyInverse",
    "LazyTransform",
    "UnconditionalDistribution",
    "UnconditionalTransform",
]


class LazyDistribution(nn.Module, abc.ABC):
    """Module whose forward(c) returns the distribution p(X | c)."""

    @abc.abstractmethod
    def forward(self, c: Tensor | None = None) -> Distribution: ...


class LazyTransform(nn.Module, abc.ABC):
    """Module whose forward(c) returns the transformation y = f(x | c)."""

    @abc.abstractmethod
    def forward(self, c: Tensor | None = None) -> Transform: ...

    @property
    def inv(self) -> "LazyTransform":
        return LazyInverse(self)


class LazyInverse(LazyTransform):
    """forward(c) = transform(c).inv"""

    def __init__(self, transform: LazyTransform) -> None:
        super().__init__()
        self.transform = transform

    def forward(self, c: Tensor | None = None) -> Transform:
        return self.transform(c).inv

    @property
    def inv(self) -> LazyTransform:
        return self.transform


class LazyComposedTransform(LazyTransform):
    """forward(c) = ComposedTransform(t_0(c), ..., t_n(c))"""

    def __init__(self, *transforms: LazyTransform) -> None:
        super().__init__()
        self.transforms = nn.ModuleList(transforms)

    def __repr__(self) -> str:
        return repr(self.transforms).replace("ModuleList", type(self).__name__, 1)

    def forward(self, c: Tensor | None = None) -> Transform:
        return ComposedTransform(*[t(c) for t in self.transforms])


class Flow(LazyDistribution):
    """Lazy normalizing flow: a lazy transformation (or a sequence of them) and a lazy base."""

    def __init__(self, transform: LazyTransform | Sequence[LazyTransform], base: LazyDistribution) -> None:
        super().__init__()
        self.transform = transform if isinstance(transform, LazyTransform) else LazyComposedTransform(*transform)
        self.base = base

    def forward(self, c: Tensor | None = None) -> NormalizingFlow:
        base = self.base(c)
        if c is not None:
            base = base.expand(c.shape[:-1])
        return NormalizingFlow(self.transform(c), base)


class UnconditionalDistribution(Partial, LazyDistribution):
    """Context-free lazy distribution built from a constructor and its (tensor) arguments."""

    def __init__(self, f: Callable[..., Distribution], *args, buffer: bool = False, **kwargs) -> None:
        super().__init__(f, *args, buffer=buffer, **kwargs)

    def extra_repr(self) -> str:
        return "" if isinstance(self.f, nn.Module) else repr(self.forward())

    def forward(self, c: Tensor | None = None) -> Distribution:
        return Partial.forward(self)


class UnconditionalTransform(Partial, LazyTransform):
    """Context-free lazy transformation built from a constructor and its (tensor) arguments."""

    def __init__(self, f: Callable[..., Transform], *args, buffer: bool = False, **kwargs) -> None:
        super().__init__(f, *args, buffer=buffer, **kwargs)

    def extra_repr(self) -> str:
        return "" if isinstance(self.f, nn.Module) else repr(self.forward())

    def forward(self, c: Tensor | None = None) -> Transform:
        return Partial.forward(self)
