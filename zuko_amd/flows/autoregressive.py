r"""Masked autoregressive transformations and MAF.

API / module-tree mirror of zuko/flows/autoregressive.py:24-316.  The conditioner is a
`zuko_amd.nn.MaskedMLP` (keys `hyper.{0,2,4,...}.{weight,bias,mask}`), the feature order is the
buffer `order`, and `forward(c)` returns a Transform whose `call_and_ladj` runs conditioner +
univariate transform + feature-sum of log|det J| on the GPU.
"""

from __future__ import annotations

from functools import partial
from math import ceil, prod
from typing import Callable, Sequence

import torch
from torch import BoolTensor, LongTensor, Size, Tensor
from torch.distributions import Transform

from ..distributions import DiagNormal
from ..lazy import Flow, LazyTransform, UnconditionalDistribution
from ..nn import MaskedMLP
from ..transforms import AutoregressiveTransform, DependentTransform, MonotonicAffineTransform
from ..utils import broadcast, unpack
from .elementwise import ElementWiseTransform

__all__ = ["MAF", "MaskedAutoregressiveTransform"]


def _univariate_name(univariate) -> str:
    f = getattr(univariate, "func", univariate)
    return getattr(f, "__name__", repr(f))


def dag_diameter(adjacency: BoolTensor) -> int:
    """Number of topological generations of the DAG `adjacency[child, parent]`
    (= sequential passes the inverse needs); asserts acyclicity.  zuko/flows/autoregressive.py:154-185."""
    indegree = adjacency.sum(dim=1).tolist()
    frontier = [n for n, d in enumerate(indegree) if d == 0]
    generations = 0
    seen = 0
    while frontier:
        generations += 1
        seen += len(frontier)
        nxt = []
        for node in frontier:
            for child in adjacency[:, node].nonzero().flatten().tolist():
                indegree[child] -= 1
                if indegree[child] == 0:
                    nxt.append(child)
        frontier = nxt
    assert seen == len(indegree), "The graph contains cycles."
    return generations


class MaskedAutoregressiveTransform(LazyTransform):
    r"""Lazy masked autoregressive transformation.

    Arguments (same as the reference): features, context, passes, order, adjacency, univariate,
    shapes, **kwargs for MaskedMLP.  With `features == 1` an `ElementWiseTransform` is returned.
    """

    def __new__(cls, features: int | None = None, context: int = 0, passes=None, order=None, adjacency=None, *args, **kwargs):
        if features is None or features > 1:
            return super().__new__(cls)
        return ElementWiseTransform(features, context, *args, **kwargs)

    def __init__(
        self,
        features: int,
        context: int = 0,
        passes: int | None = None,
        order: LongTensor | None = None,
        adjacency: BoolTensor | None = None,
        univariate: Callable[..., Transform] = MonotonicAffineTransform,
        shapes: Sequence[Size] = ((), ()),
        **kwargs,
    ) -> None:
        super().__init__()
        self.univariate = univariate
        self.shapes = shapes
        self.total = sum(prod(s) for s in shapes)
        self.features = features
        self.context = context
        self.register_buffer("order", None)

        ctx_adj = None
        if adjacency is None:
            passes = features if passes is None else passes
            order = torch.arange(features) if order is None else torch.as_tensor(order, dtype=int)
            assert order.ndim == 1, "'order' should be a vector."
            assert order.shape[0] == features, f"'order' should have {features} elements."
            self.passes = min(max(passes, 1), features)
            self.order = torch.div(order, ceil(features / self.passes), rounding_mode="floor")
            adjacency = self.order[:, None] > self.order
        else:
            adjacency = torch.as_tensor(adjacency, dtype=bool)
            assert adjacency.ndim == 2, "'adjacency' should be a matrix."
            assert adjacency.shape[0] == features, f"'adjacency' should have {features} rows."
            assert adjacency.shape[1] in (features, features + context), f"'adjacency' should have {features} or {features + context} columns."
            if adjacency.shape[1] > features:
                ctx_adj = adjacency[:, features:]
            adjacency = adjacency[:, :features]
            assert adjacency.diag().all(), "'adjacency' should have ones on the diagonal."
            adjacency = adjacency * ~torch.eye(features, dtype=bool)
            self.passes = dag_diameter(adjacency)

        if context > 0:
            if ctx_adj is None:
                ctx_adj = torch.ones((features, context), dtype=bool)
            adjacency = torch.cat((adjacency, ctx_adj), dim=1)

        # output row f*total + j carries parameter j of feature f
        self.hyper = MaskedMLP(torch.repeat_interleave(adjacency, repeats=self.total, dim=0), **kwargs)

    def extra_repr(self) -> str:
        lines = [f"(base): {_univariate_name(self.univariate)}"]
        if self.order is None:
            lines.append(f"(passes): {self.passes}")
        else:
            o = self.order.tolist()
            text = str(o) if len(o) <= 10 else "[" + ", ".join(map(str, o[:5])) + ", ..., " + ", ".join(map(str, o[-5:])) + "]"
            lines.append(f"(order): {text}")
        return "\n".join(lines)

    def meta(self, c: Tensor | None, x: Tensor) -> Transform:
        """x (and c) -> conditioner -> packed phi[..., D, total] -> univariate transform over D."""
        if c is not None:
            x = torch.cat(broadcast(x, c, ignore=1), dim=-1)
        phi = self.hyper(x).unflatten(-1, (-1, self.total))
        return DependentTransform(self.univariate(*unpack(phi, self.shapes)), 1)

    def forward(self, c: Tensor | None = None) -> Transform:
        return AutoregressiveTransform(partial(self.meta, c), self.passes)


class MAF(Flow):
    r"""Masked autoregressive flow: `transforms` autoregressive layers with alternating
    ascending / descending feature order (or random permutations) over a standard-normal base.
    Mirrors zuko/flows/autoregressive.py:221-316."""

    def __init__(self, features: int, context: int = 0, transforms: int = 3, randperm: bool = False, **kwargs) -> None:
        ascending = torch.arange(features)
        fixed = [ascending, torch.flipud(ascending)]
        layers = [
            MaskedAutoregressiveTransform(
                features=features,
                context=context,
                order=torch.randperm(features) if randperm else fixed[i % 2],
                **kwargs,
            )
            for i in range(transforms)
        ]
        base = UnconditionalDistribution(DiagNormal, loc=torch.zeros(features), scale=torch.ones(features), buffer=True)
        super().__init__(layers, base)
