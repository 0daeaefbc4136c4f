r"""Polynomial flows: SOSPF (sum-of-squares) and BPF (bounded Bernstein).
Mirrors zuko/flows/polynomial.py:23-117."""

from __future__ import annotations

from functools import partial

from ..lazy import UnconditionalTransform
from ..transforms import BoundedBernsteinTransform, ShiftedSOSPolynomialTransform, SoftclipTransform
from .autoregressive import MAF

__all__ = ["BPF", "SOSPF"]


class SOSPF(MAF):
    r"""Autoregressive SOS-polynomial layers (shifted by a learned constant) with a
    Softclip(bound=11) between consecutive layers."""

    def __init__(self, features: int, context: int = 0, degree: int = 4, polynomials: int = 3, slope: float = 1e-3, **kwargs) -> None:
        super().__init__(
            features=features,
            context=context,
            univariate=partial(ShiftedSOSPolynomialTransform, slope=slope),
            shapes=[(polynomials, degree + 1), ()],
            **kwargs,
        )
        layers = self.transform.transforms
        for i in range(len(layers) - 1, 0, -1):
            layers.insert(i, UnconditionalTransform(SoftclipTransform, bound=11.0))


class BPF(MAF):
    r"""Autoregressive bounded-Bernstein-polynomial layers."""

    def __init__(self, features: int, context: int = 0, degree: int = 16, **kwargs) -> None:
        super().__init__(
            features=features,
            context=context,
            univariate=BoundedBernsteinTransform,
            shapes=[(degree + 1,)],
            **kwargs,
        )
