r"""Coupling transformations, NICE and RealNVP.

API / module-tree mirror of zuko/flows/coupling.py:25-200: dense `zuko_amd.nn.MLP` conditioner
(`hyper.{0,2,...}.{weight,bias}`), boolean buffer `mask` (True = pass-through half).
"""

from __future__ import annotations

import os
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
        return FusedCouplingTransform(self, c)

    def split_indices(self) -> tuple[Tensor, Tensor]:
        """(idx_a, idx_b) of zuko/transforms.py:1037-1038, computed once per version of the `mask` buffer (nonzero() is a
        device -> host synchronisation)."""
        key = (self.mask._version, self.mask.data_ptr(), str(self.mask.device))
        cached = self.__dict__.get("_split_cache")
        if cached is None or cached[0] != key:
            cached = (key, self.mask.nonzero().squeeze(-1), (~self.mask).nonzero().squeeze(-1))
            self.__dict__["_split_cache"] = cached
        return cached[1], cached[2]

    def fused_state(self, device: torch.device):
        """Plan + device tables of the fused coupling kernel (csrc/fused_coupling.hip), or None when the layer does not fit it
        (affine univariate with default shapes, plain (Linear, activation)* conditioner, widths <= 512, inputs <= 256)."""
        from .. import coupling_plan as cp
        from ..nn import Linear, _act_code

        cache = _COUPLING_CACHE.setdefault(self, {})
        key = (str(device), self.mask._version, self.mask.data_ptr())
        if cache.get("key") != key:
            cache.clear()
            cache["key"] = key
            state = None
            u = self.univariate
            f, kw = (u.func, dict(u.keywords)) if isinstance(u, partial) else (u, {})
            slope = kw.pop("slope", 1e-3)
            mods = list(self.hyper)
            lins, acts = mods[0::2], mods[1::2]
            simple = len(mods) == 2 * len(lins) - 1 and all(isinstance(m, Linear) for m in lins) and not any(isinstance(m, Linear) for m in acts)
            codes = {_act_code(m) for m in acts}
            if (f is MonotonicAffineTransform and not kw and not (isinstance(u, partial) and u.args) and [tuple(s_) for s_ in self.shapes] == [(), ()]
                    and simple and len(codes) == 1 and None not in codes and all(l.weight.dtype == torch.float32 for l in lins)):
                idx_a = self.mask.nonzero().squeeze(-1).cpu().numpy()
                idx_b = (~self.mask).nonzero().squeeze(-1).cpu().numpy()
                context = lins[0].weight.shape[1] - len(idx_a)
                plan = cp.build_coupling_plan([tuple(l.weight.shape) for l in lins], idx_a, idx_b, int(self.mask.numel()), context)
                if plan is not None:
                    state = cp.FusedCoupling(plan, device, codes.pop(), slope)
            cache["state"] = state
        return cache["state"]


import weakref

_COUPLING_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


class FusedCouplingTransform(CouplingTransform):
    r"""The Transform `GeneralCouplingTransform.forward(c)` returns: zuko's CouplingTransform(partial(meta, c), mask)
    (zuko/transforms.py:1010-1073) whose `call_and_ladj` runs conditioner + affine map + merge in ONE kernel
    (zk_coupling_forward) when no gradient is required and the layer fits; otherwise the layer-wise kernels through `meta`."""

    def __init__(self, lazy: GeneralCouplingTransform, c: Tensor | None) -> None:
        # (not CouplingTransform.__init__: its two mask.nonzero() calls are a device -> host sync per forward, 16 per
        #  RealNVP step; the split indices are cached per mask version and only the layer-wise path reads them)
        Transform.__init__(self)
        self.meta = partial(lazy.meta, c)
        self.lazy = lazy
        self.c = c

    @property
    def idx_a(self) -> Tensor:
        return self.lazy.split_indices()[0]

    @property
    def idx_b(self) -> Tensor:
        return self.lazy.split_indices()[1]

    def _fused(self, x: Tensor, inverse: bool):
        """(result, ladj of the forward map) from the fused kernel, or None when the layer-wise path has to be taken."""
        lazy, c = self.lazy, self.c
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 1) or os.environ.get("ZUKO_AMD_NO_FUSED_COUPLING", "0") == "1":
            return None
        if torch.is_grad_enabled() and (x.requires_grad or (c is not None and c.requires_grad) or any(p.requires_grad for p in lazy.hyper.parameters())):
            return None
        st = lazy.fused_state(x.device)
        if st is None:
            return None
        D = x.shape[-1]
        feats, ctx = int(lazy.mask.numel()), (0 if c is None else c.shape[-1])
        if D != feats or ctx != st.plan.context:  # (the reference's F.linear raises on these, zuko/nn.py:15)
            raise RuntimeError(f"zuko_amd: input of {D} features + {ctx} context columns given to a coupling transform built for {feats} + {st.plan.context}")
        if c is not None:
            xb, cb = broadcast(x, c, ignore=1)
        else:
            xb, cb = x, None
        batch = xb.shape[:-1]
        x2 = xb.reshape(-1, D)
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        c2 = None
        if cb is not None:
            c2 = cb.reshape(-1, cb.shape[-1])
            c2 = c2 if c2.stride(-1) == 1 else c2.contiguous()
        st.refresh(list(lazy.hyper)[0::2])
        y, ladj = st.run(x2, c2, inverse)
        return y.reshape(batch + (D,)), ladj.reshape(batch)

    def _trained(self, x: Tensor):
        """(y, ladj) through the one-node training path (zuko_amd/coupling_train.py), or None."""
        from .. import coupling_train

        lazy, c = self.lazy, self.c
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 1 and torch.is_grad_enabled()):
            return None
        if c is not None:
            xb, cb = broadcast(x, c, ignore=1)
        else:
            xb, cb = x, None
        batch = xb.shape[:-1]
        out = coupling_train.coupling(lazy, xb.reshape(-1, xb.shape[-1]), None if cb is None else cb.reshape(-1, cb.shape[-1]))
        if out is None:
            return None
        return out[0].reshape(batch + (xb.shape[-1],)), out[1].reshape(batch)

    def call_and_ladj(self, x: Tensor):
        out = self._fused(x, False)
        if out is None:
            out = self._trained(x)
        return super().call_and_ladj(x) if out is None else out

    def _call(self, x: Tensor) -> Tensor:
        return self.call_and_ladj(x)[0]

    def log_abs_det_jacobian(self, x: Tensor, y: Tensor) -> Tensor:
        return self.call_and_ladj(x)[1]

    def inverse_and_ladj(self, y: Tensor):
        """(x, log|det dy/dx| of the FORWARD map at x): one launch of zk_coupling_inverse (CouplingTransform._inverse,
        zuko/transforms.py:1050-1056, plus what rsample_and_log_prob needs, zuko/distributions.py:129-138)."""
        out = self._fused(y, True)
        if out is not None:
            return out
        x = super()._inverse(y)
        return x, super().log_abs_det_jacobian(x, y)

    def _inverse(self, y: Tensor) -> Tensor:
        out = self._fused(y, True)
        return super()._inverse(y) if out is None else out[0]

    @property
    def inv(self):
        from .autoregressive import _FusedInverse

        return _FusedInverse(self)


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
