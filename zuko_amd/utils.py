r"""Shape helpers shared by the flow factories (host-side logic, no arithmetic)."""

from __future__ import annotations

import math
from typing import Any, Callable, Sequence

import torch
import torch.nn as nn
from torch import Tensor

__all__ = ["Partial", "broadcast", "unpack"]


def broadcast(*tensors: Tensor, ignore: int | Sequence[int] = 0) -> list[Tensor]:
    """Expand the leading dims of `tensors` to a common shape, leaving the last `ignore` dims of
    each untouched (same contract as zuko/utils.py:212-244)."""
    skip = [ignore] * len(tensors) if isinstance(ignore, int) else list(ignore)
    cut = [t.dim() - s for t, s in zip(tensors, skip)]
    lead = torch.broadcast_shapes(*[t.shape[:c] for t, c in zip(tensors, cut)])
    return [t.expand(lead + t.shape[c:]) for t, c in zip(tensors, cut)]


def unpack(x: Tensor, shapes: Sequence[Sequence[int]]) -> tuple[Tensor, ...]:
    """Split the last dim of a packed tensor into views of the given shapes
    (same contract as zuko/utils.py:596-622); the pieces alias `x`, which is what lets the
    kernels recognise the packed phi[N, D, total] layout."""
    sizes = [math.prod(s) for s in shapes]
    return tuple(_shape_tail(piece, s) for piece, s in zip(x.split(sizes, -1), shapes))


def _shape_tail(piece: Tensor, shape: Sequence[int]) -> Tensor:
    shape = tuple(shape)
    if len(shape) == 0:
        return piece.squeeze(-1)
    if len(shape) == 1:
        return piece
    return piece.unflatten(-1, shape)


class Partial(nn.Module):
    """`functools.partial` as a module: tensor arguments become buffers or parameters so they move
    with `.to(device)` and appear in the state_dict under `_0, _1, ...` / their keyword
    (key layout of zuko/utils.py:26-115, e.g. `base._0`... NOTE: flows register `loc`/`scale`)."""

    def __init__(self, f: Callable, /, *args: Any, buffer: bool = False, **kwargs: Any) -> None:
        super().__init__()
        self.f = f
        self._nargs = len(args)
        self._keys = list(kwargs)
        for name, value in [(f"_{i}", a) for i, a in enumerate(args)] + list(kwargs.items()):
            if torch.is_tensor(value):
                if buffer:
                    self.register_buffer(name, value)
                else:
                    self.register_parameter(name, nn.Parameter(value))
            else:
                setattr(self, name, value)

    @property
    def args(self) -> list:
        return [getattr(self, f"_{i}") for i in range(self._nargs)]

    @property
    def kwargs(self) -> dict:
        return {k: getattr(self, k) for k in self._keys}

    def extra_repr(self) -> str:
        return "" if isinstance(self.f, nn.Module) else f"(f): {self.f}"

    def forward(self, *args: Any, **kwargs: Any) -> Any:
        return self.f(*self.args, *args, **self.kwargs, **kwargs)
