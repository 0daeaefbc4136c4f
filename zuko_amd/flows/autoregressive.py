r"""Masked autoregressive transformations and MAF.

API / module-tree mirror of zuko/flows/autoregressive.py:24-316.  The conditioner is a
`zuko_amd.nn.MaskedMLP` (keys `hyper.{0,2,4,...}.{weight,bias,mask}`), the feature order is the
buffer `order`, and `forward(c)` returns a Transform whose `call_and_ladj` runs conditioner +
univariate transform + feature-sum of log|det J| on the GPU.
"""

from __future__ import annotations

from functools import partial
import math
import os
from math import ceil, prod
from typing import Callable, Sequence

import torch
from torch import BoolTensor, LongTensor, Size, Tensor
from torch.distributions import Transform

import weakref

import torch.nn as nn

from .. import fused
from ..distributions import DiagNormal
from ..lazy import Flow, LazyTransform, UnconditionalDistribution
from ..nn import MaskedLinear, MaskedMLP, _act_code
from ..transforms import AutoregressiveTransform, DependentTransform, MonotonicAffineTransform, MonotonicRQSTransform
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


_ELEMENTWISE = (nn.ReLU, nn.ReLU6, nn.Tanh, nn.Sigmoid, nn.ELU, nn.CELU, nn.SELU, nn.SiLU, nn.GELU, nn.LeakyReLU, nn.Softplus, nn.Softsign, nn.Hardtanh,
                nn.Hardswish, nn.Hardsigmoid, nn.Mish, nn.Tanhshrink, nn.LogSigmoid, nn.Identity)


def _elementwise_stateless(m: nn.Module) -> bool:
    """True for activations that map every element on its own and hold no parameters or buffers (exact types: a subclass may do anything)."""
    return type(m) in _ELEMENTWISE and next(m.parameters(), None) is None and next(m.buffers(), None) is None


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
        u = self.univariate(*unpack(phi, self.shapes))
        if type(u) in (MonotonicAffineTransform, MonotonicRQSTransform):
            u._packed = phi  # the views above were cut from this tensor: autograd can differentiate it directly
        return DependentTransform(u, 1)

    def forward(self, c: Tensor | None = None) -> Transform:
        return FusedAutoregressiveTransform(self, c)

    def _sweep_features(self, device, passes: int) -> list:
        """Per sweep s of the inverse, (s, the features of order s): (lo, hi) when they are consecutive, else their indices on `device`; empty sweeps
        are dropped.  Host-side, cached per (order version, device): the sweeps themselves then run without a device synchronisation."""
        key = (str(device), passes, self.order._version, self.order.data_ptr())
        cached = getattr(self, "_sweep_cache", None)
        if cached is None or cached[0] != key:
            order = self.order.detach().cpu()
            out = []
            for s_ in range(passes):
                idx = (order == s_).nonzero().squeeze(-1)
                if idx.numel() == 0:
                    continue
                lo, hi = int(idx[0]), int(idx[-1]) + 1
                out.append((s_, (lo, hi) if hi - lo == idx.numel() else idx.to(device)))
            cached = (key, out)
            self._sweep_cache = cached
        return cached[1]

    def _sweep_units(self, device, passes: int):
        """For a plain (MaskedLinear, activation)* MaskedLinear conditioner: per hidden layer and sweep s, the units that become FINAL at the start
        of sweep s — every input they are connected to is a context column or a feature of order < s, or a unit of the layer before that was final
        by then (r(unit) = max over its unmasked inputs; features: order + 1, context: 0).  A unit is evaluated once, in the sweep it becomes
        final; units that are never final within `passes` sweeps feed nothing the inverse reads.  Returns None for other conditioner structures.
        Host-side, cached per (mask / order versions, device)."""
        mods = list(self.hyper) if isinstance(self.hyper, nn.Sequential) else []
        lins = mods[0::2]
        if not mods or len(mods) % 2 == 0 or any(type(m) is not MaskedLinear for m in lins) or any(isinstance(m, (nn.Linear, MaskedLinear)) for m in mods[1::2]):
            return None
        # the sweeps apply the activation to a SUBSET of a layer's units: only modules known to act element by element, without per-unit
        # state, may take that path (nn.PReLU(H), a normalisation layer, Softmax ... fall to the whole-layer wavefront form)
        if any(not _elementwise_stateless(m) for m in mods[1::2]):
            return None
        key = (str(device), passes, self.order._version, self.order.data_ptr()) + tuple((l.mask._version, l.mask.data_ptr()) for l in lins)
        cached = getattr(self, "_unit_cache", None)
        if cached is None or cached[0] != key:
            order = self.order.detach().cpu()
            ready = torch.cat((order + 1, torch.zeros(lins[0].mask.shape[1] - order.numel(), dtype=order.dtype)))
            table = []
            for lin in lins[:-1]:
                m = lin.mask.detach().cpu()
                ready = torch.where(m, ready[None, :].expand_as(m), torch.zeros((), dtype=ready.dtype)).amax(dim=1)  # [out]
                table.append([(ready == s_).nonzero().squeeze(-1) for s_ in range(passes)])
            table = [[(None if u.numel() == 0 else u.to(device)) for u in row] for row in table]
            cached = (key, table)
            self._unit_cache = cached
        return cached[1]

    def _sweep_unit_rows(self, device, passes: int):
        """The hidden layers' weights, masks and biases with their ROWS gathered sweep by sweep (the units of _sweep_units, in that order), so
        that a sweep's skinny GEMM takes a row slice — a view — instead of three gathers.  Columns stay in module order: a unit's dot product
        runs over the same operands in the same order as in the reference's full GEMM.  Per hidden layer: (weight, mask, bias | None, start
        offset of every sweep's rows); cached per parameter version (zuko_amd.invalidate drops it)."""
        from ..nn import _param_stamp

        units = self._sweep_units(device, passes)
        if units is None:
            return None
        lins = list(self.hyper)[0:-1:2]
        key = (str(device), passes, self._unit_cache[0], _param_stamp(lins))  # (the unit table's own key: an id() can be reused by a rebuilt table)
        cached = getattr(self, "_unit_rows_cache", None)
        if cached is None or cached[0] != key:
            out = []
            with torch.no_grad():
                for l, lin in enumerate(lins):
                    rows = [u for u in units[l] if u is not None]
                    starts, n = [], 0
                    for u in units[l]:
                        starts.append(n)
                        n += 0 if u is None else u.numel()
                    starts.append(n)
                    perm = torch.cat(rows) if rows else torch.zeros(0, dtype=torch.long, device=device)
                    out.append((lin.weight.detach().index_select(0, perm), lin.mask.index_select(0, perm), None if lin.bias is None else lin.bias.detach().index_select(0, perm), starts))
            cached = (key, out)
            self._unit_rows_cache = cached
        return cached[1]

    # ---- fused-kernel support ----------------------------------------------------------------

    def _fusable_layout(self):
        """(UniLayout, bound, slope) if conditioner + univariate fit csrc/fused_ar.hip, else None."""
        u = self.univariate
        f, kw = (u.func, dict(u.keywords)) if isinstance(u, partial) else (u, {})
        if isinstance(u, partial) and u.args:
            return None
        shapes = [tuple(s) for s in self.shapes]
        slope = kw.pop("slope", 1e-3)  # (the Bernstein maps take no slope: a partial with one is not fused, see below)
        had_slope = isinstance(u, partial) and "slope" in u.keywords
        if f is MonotonicAffineTransform and shapes == [(), ()] and not kw:
            return fused.uni_layout("affine", 2), 5.0, slope
        from .spline import CircularRQSTransform  # (spline.py imports this module)

        spline_shapes = len(shapes) == 3 and len(shapes[0]) == 1 and shapes[0] == shapes[1] and shapes[2] == (shapes[0][0] - 1,)
        if f is MonotonicRQSTransform and spline_shapes:
            bound = kw.pop("bound", 5.0)
            lay = fused.uni_layout("rqs", self.total, shapes[0][0])
            if lay is not None and not kw and fused.layout_supports(lay, self.features):
                return lay, bound, slope
        if f is CircularRQSTransform and spline_shapes and not kw:
            lay = fused.uni_layout("crqs", self.total, shapes[0][0])
            if lay is not None and fused.layout_supports(lay, self.features):
                return lay, math.pi, slope
        # the polynomial maps at their flows' default sizes (forward only; operand-split static-shape kernels, zuko_amd/static_ar.py)
        from ..transforms import BoundedBernsteinTransform, ShiftedSOSPolynomialTransform
        from ..ops import SOS_BOUND

        if f is ShiftedSOSPolynomialTransform and shapes == [(3, 5), ()] and not kw:
            return fused.uni_layout("sos", 16), SOS_BOUND, slope
        if f is BoundedBernsteinTransform and shapes == [(17,)] and set(kw) <= {"bound", "eps"} and not had_slope:
            lay = fused.uni_layout("bern", 17)
            return lay, kw.get("bound", 5.0), float(kw.get("eps", 1e-6))  # (third slot: the continuation margin, this kind has no slope)
        return None

    def _rqs_spec(self):
        """(K, bound, slope) when the univariate map is a plain MonotonicRQSTransform over packed (K, K, K-1) parameters."""
        u = self.univariate
        f, kw = (u.func, dict(u.keywords)) if isinstance(u, partial) else (u, {})
        if (isinstance(u, partial) and u.args) or f is not MonotonicRQSTransform:
            return None
        shapes = [tuple(s) for s in self.shapes]
        if not (len(shapes) == 3 and len(shapes[0]) == 1 and shapes[0] == shapes[1] and shapes[2] == (shapes[0][0] - 1,)):
            return None
        slope, bound = kw.pop("slope", 1e-3), kw.pop("bound", 5.0)
        return None if kw else (shapes[0][0], bound, slope)

    def incremental_state(self, device: torch.device):
        """Plan + device tables of the incremental inverse kernel (csrc/inc_inverse.hip), or None when the conditioner does
        not fit its aligned-tile layout (zuko_amd/incremental.py) — the partial sweeps are used then."""
        from .. import incremental as inc

        if self.order is None:
            return None
        cache = _FUSED_CACHE.setdefault(self, {})
        lins = [m for m in self.hyper if isinstance(m, MaskedLinear)]
        structure = tuple((l.mask._version, l.mask.data_ptr()) for l in lins) + (self.order._version, self.order.data_ptr())
        key = str(device) + "/inc"
        if key in cache and cache[key][0] != structure:
            del cache[key]
        if key not in cache:
            state = None
            lay = self._fusable_layout()
            mods = list(self.hyper)
            simple = all(isinstance(a, MaskedLinear) != (i % 2 == 1) for i, a in enumerate(mods))
            codes = {_act_code(a) for a in mods if not isinstance(a, MaskedLinear)}
            if lay is not None and lay[0].kind in (0, 1, 2, 3, 5, 6) and simple and len(codes) == 1 and None not in codes and all(l.weight.dtype == torch.float32 for l in lins):
                layout = fused.UniLayout(lay[0].kind, lay[0].total, 1, (lay[0].total + 3) // 4, lay[0].bins)
                plan = inc.build_inc_plan([l.mask for l in lins], self.features, self.order.cpu().numpy(), layout)
                if plan is not None:
                    # (kind 6, the Bernstein map: the layout's third slot is its continuation margin, it has no slope)
                    state = inc.IncAR(plan, lins, device, codes.pop(), lay[1], 1e-3 if lay[0].kind == 6 else lay[2], eps=lay[2] if lay[0].kind == 6 else None)
            cache[key] = (structure, state)
        return cache[key][1]

    def fused_state(self, device: torch.device, inverse: bool = False):
        """Plan + device tables of the fused kernel (built once per device), or None.  `inverse=True`
        returns the group-aligned plan used by the partial (wavefront) inverse sweeps, when the layer
        was built from a feature `order` (not from a free-form adjacency)."""
        cache = _FUSED_CACHE.setdefault(self, {})
        key = str(device) + ("/inv" if inverse else "")
        if inverse and self.order is None:
            return None
        # the plan is a function of the mask / order BUFFERS: load_state_dict() may overwrite them (randperm flows,
        # custom adjacencies), so their versions are part of the cache entry
        lins_all = [m for m in self.hyper if isinstance(m, MaskedLinear)]
        structure = tuple((l.mask._version, l.mask.data_ptr()) for l in lins_all) + ((self.order._version, self.order.data_ptr()) if self.order is not None else ())
        if key in cache and cache[key][0] != structure:
            del cache[key]
        if key not in cache:
            state = None
            lay = self._fusable_layout()
            mods = list(self.hyper)
            lins = [m for m in mods if isinstance(m, MaskedLinear)]
            acts = [m for m in mods if not isinstance(m, MaskedLinear)]
            simple = all(isinstance(a, MaskedLinear) != (i % 2 == 1) for i, a in enumerate(mods))  # lin, act, lin, ...
            codes = {_act_code(a) for a in acts}
            if lay is not None and simple and len(codes) == 1 and None not in codes and all(l.weight.dtype == torch.float32 for l in lins):
                variant = fused.default_variant()
                # conditioners wider than 256 (up to 512) have no generic kernel: forward only, through a static-shape kernel (zuko_amd/static_ar.py)
                wide = max([l.weight.shape[0] for l in lins[:-1]] + [lins[0].weight.shape[1]]) > fused.MAX_WIDTH
                poly = lay[0].kind in (5, 6)  # forward only (their inverse is a bisection: layer-wise kernels)
                plan = None if ((wide or poly) and inverse) else fused.build_plan([l.mask for l in lins], self.features, lay[0], fused.chunk_of(variant), align_groups=inverse,
                                                                        max_width=fused.MAX_WIDTH_WIDE if wide else fused.MAX_WIDTH)
                if plan is not None:
                    state = fused.FusedAR(plan, device, codes.pop(), lay[1], 1e-3 if lay[0].kind == 6 else lay[2], variant)
                    if lay[0].kind == 6:
                        state.eps = lay[2]
                    if inverse:
                        state.set_sweeps(self.order.cpu().numpy(), self.passes)
            cache[key] = (structure, state)
        return cache[key][1]


_FUSED_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _incremental_enabled() -> bool:
    return os.environ.get("ZUKO_AMD_NO_INCREMENTAL", "0") != "1"


def _partial_inverse_enabled() -> bool:
    import os

    return os.environ.get("ZUKO_AMD_FULL_SWEEPS", "0") != "1"


class FusedAutoregressiveTransform(AutoregressiveTransform):
    r"""The Transform `MaskedAutoregressiveTransform.forward(c)` returns.

    Same interface and semantics as zuko's `AutoregressiveTransform(partial(meta, c), passes)`
    (zuko/transforms.py:966-1007), but `call_and_ladj` / `_call` / `log_abs_det_jacobian` run the
    conditioner, the univariate transform and the feature-sum of log|det J| in ONE kernel
    (zk_ar_forward) whenever the layer fits it (fp32, widths <= 256, affine or 8-bin RQS, fusable
    activation); otherwise the layer-by-layer kernels are used through `meta`."""

    def __init__(self, lazy: MaskedAutoregressiveTransform, c: Tensor | None) -> None:
        super().__init__(partial(lazy.meta, c), lazy.passes)
        self.lazy = lazy
        self.c = c

    def _fused(self, x: Tensor, need_generic: bool = False):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 1):
            return None
        if torch.is_grad_enabled() and (x.requires_grad or (self.c is not None and self.c.requires_grad) or any(p.requires_grad for p in self.lazy.hyper.parameters())):
            return None
        st = self.lazy.fused_state(x.device)
        if st is None:
            return None
        rows = x.numel() // max(x.shape[-1], 1)
        if self.c is not None and self.c.dim() > 1:
            rows = max(rows, self.c.numel() // max(self.c.shape[-1], 1))
        # (large batches may compile the conditioner's static-shape kernel here; conditioners wider than the generic kernel need one)
        if not st.ready(rows) or (need_generic and not st.generic_ok):
            return None
        return st

    def _check_widths(self, x: Tensor, st) -> None:
        """The reference's F.linear raises on a feature / context width the conditioner was not built for (zuko/nn.py:217-218);
        the fused kernels walk a weight stream laid out for `plan.din` inputs, so the same misuse must raise here too."""
        D = self.lazy.features
        C = 0 if self.c is None else self.c.shape[-1]
        din = getattr(getattr(st, "plan", None), "din", None)
        if x.shape[-1] != D or (din is not None and D + C != din):
            raise RuntimeError(f"zuko_amd: input of {x.shape[-1]} features + {C} context columns given to a conditioner built for "
                               f"{D} features + {(din - D) if din is not None else '?'} context columns")

    def _bf16_spline(self, x: Tensor):
        """bf16 storage path (cfg5): conditioner on bf16 MFMA, the spline in the last layer's epilogue — phi stays on chip."""
        lazy = self.lazy
        if not (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() >= 1) or os.environ.get("ZUKO_AMD_BF16_UNFUSED", "0") == "1":
            return None
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in lazy.hyper.parameters())):
            return None
        spec = lazy._rqs_spec()
        if spec is None or spec[0] not in (8, 16) or not hasattr(lazy.hyper, "bf16_rqs"):
            return None
        c = self.c
        if c is not None:
            xb, cb = broadcast(x, c, ignore=1)
            inp = torch.cat((xb, cb), dim=-1)
        else:
            xb, inp = x, x
        D = lazy.features
        batch = xb.shape[:-1]
        out = lazy.hyper.bf16_rqs(inp.reshape(-1, inp.shape[-1]), xb.reshape(-1, D), *spec)
        if out is None:
            return None
        return out[0].reshape(batch + (D,)), out[1].reshape(batch)

    def call_and_ladj(self, x: Tensor):
        st = self._fused(x)
        if st is None:
            out = self._bf16_spline(x)
            if out is None:
                out = self._train_fused(x)
            return out if out is not None else super().call_and_ladj(x)
        return self._run_fused(st, x, None)

    def _train_fused(self, x: Tensor):
        """Under autograd: the whole transform as one autograd node (zuko_amd/train.py: AutoregressiveFn) when it is fp32, features + context is a
        multiple of 4 and its conditioner has an operand-split static-shape kernel; None otherwise (conditioner and univariate map as separate nodes)."""
        lazy, c = self.lazy, self.c
        D = lazy.features
        if not (torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 1 and x.shape[-1] == D):
            return None
        if c is not None and not (c.is_cuda and c.dtype == torch.float32 and (D + c.shape[-1]) % 4 == 0):
            return None
        lay = lazy._fusable_layout()
        if lay is None or lay[0].kind not in (0, 1):
            return None
        from .. import train

        sizes = (1, 1) if lay[0].kind == 0 else (lay[0].bins, lay[0].bins, lay[0].bins - 1)
        if c is not None:  # the conditioner's input, as the reference builds it (zuko/flows/autoregressive.py:209-210): autograd splits its gradient
            xb, cb = broadcast(x, c, ignore=1)
            inp = torch.cat((xb, cb), dim=-1)
        else:
            xb, inp = x, x
        batch = xb.shape[:-1]
        out = train.autoregressive(lazy.hyper, (lay[0].kind, float(lay[1]), float(lay[2]), sizes), inp.reshape(-1, inp.shape[-1]), features=D)
        return None if out is None else (out[0].reshape(batch + (D,)), out[1].reshape(batch))

    def call_and_accumulate_ladj(self, x: Tensor, total: Tensor):
        """y of call_and_ladj(x), with log|dy/dx| ADDED to `total` by the kernel (`accumulate` of zk_ar_forward) instead of returned:
        what ComposedTransform's `total + ladj` (zuko/transforms.py:141-150) amounts to, without the extra elementwise launch per
        transform.  Returns None when this call does not run on the fused kernel or `total` cannot be accumulated into in place."""
        if torch.is_grad_enabled() and (x.requires_grad or total.requires_grad):
            return None
        st = self._fused(x)
        batch = x.shape[:-1] if self.c is None else torch.broadcast_shapes(x.shape[:-1], self.c.shape[:-1])
        if st is None or total.dtype != x.dtype or total.device != x.device or tuple(total.shape) != tuple(batch) or not total.is_contiguous():
            return None
        return self._run_fused(st, x, total)[0]

    def _run_fused(self, st, x: Tensor, total):
        self._check_widths(x, st)
        lazy, c = self.lazy, self.c
        D = lazy.features
        if c is not None:
            xb, cb = broadcast(x, c, ignore=1)
        else:
            xb, cb = x, None
        batch = xb.shape[:-1]
        x2 = xb.reshape(-1, D)
        din = D + (0 if cb is None else cb.shape[-1])
        dinp = -(-din // 4) * 4
        if cb is None and dinp == D and x2.stride(-1) == 1 and x2.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0:
            inp = x2
        else:
            inp = x2.new_zeros((x2.shape[0], dinp))
            inp[:, :D] = x2
            if cb is not None:
                inp[:, D:din] = cb.reshape(-1, cb.shape[-1])
        y = torch.empty((x2.shape[0], D), dtype=x.dtype, device=x.device)
        ladj = torch.empty(x2.shape[0], dtype=x.dtype, device=x.device) if total is None else total.view(-1)
        st.refresh([m for m in lazy.hyper if isinstance(m, MaskedLinear)])
        st.run(inp, y, ladj, total is not None)
        return y.reshape(batch + (D,)), ladj.reshape(batch)

    def _call(self, x: Tensor) -> Tensor:
        return self.call_and_ladj(x)[0]

    def log_abs_det_jacobian(self, x: Tensor, y: Tensor) -> Tensor:
        return self.call_and_ladj(x)[1]

    @property
    def inv(self) -> Transform:
        return _FusedInverse(self)

    def inverse_and_ladj(self, y: Tensor):
        """(x, log|det dy/dx| of the FORWARD map at x).  With the incremental kernel both come out of the one launch that
        inverts (what `rsample_and_log_prob` needs, zuko/distributions.py:129-138); otherwise inverse + one forward."""
        out = self._inverse_impl(y, True)
        if isinstance(out, tuple):
            return out
        return out, self.log_abs_det_jacobian(out, y)

    def _inverse(self, y: Tensor) -> Tensor:
        return self._inverse_impl(y, False)

    def _inverse_impl(self, y: Tensor, want_ladj: bool):
        """`passes` sweeps x <- meta(x).inv(y) from x = 0 (zuko/transforms.py:994-1000): one incremental launch when the
        conditioner fits the aligned-tile plan, else one fused launch per (partial) sweep, updating the buffer in place."""
        st = self._fused(y, need_generic=True)  # (the sweeps below run on the generic kernel)
        if st is None:
            out = self._poly_incremental(y, want_ladj)  # the polynomial maps: no generic kernel, but the incremental launch inverts them (round 6)
            return out if out is not None else self._ordered_inverse(y)
        self._check_widths(y, st)
        lazy, c = self.lazy, self.c
        D = lazy.features
        if c is not None:
            yb, cb = broadcast(y, c, ignore=1)
        else:
            yb, cb = y, None
        batch = yb.shape[:-1]
        y2 = yb.reshape(-1, D).contiguous()
        inc_state = lazy.incremental_state(y.device) if _incremental_enabled() and _partial_inverse_enabled() else None
        if inc_state is not None:
            # incremental form: one launch, every off-diagonal weight tile multiplied once per sample
            inc_state.refresh([m for m in lazy.hyper if isinstance(m, MaskedLinear)])
            c2 = None if cb is None else cb.reshape(-1, cb.shape[-1]).contiguous()
            x2, l2 = inc_state.run(y2, c2, want_ladj)
            return (x2.reshape(batch + (D,)), l2.reshape(batch)) if want_ladj else x2.reshape(batch + (D,))
        din = D + (0 if cb is None else cb.shape[-1])
        buf = y2.new_zeros((y2.shape[0], -(-din // 4) * 4))
        if cb is not None:
            buf[:, D:din] = cb.reshape(-1, cb.shape[-1])
        lins = [m for m in lazy.hyper if isinstance(m, MaskedLinear)]
        part = lazy.fused_state(y.device, inverse=True) if _partial_inverse_enabled() else None
        if part is not None:
            # wavefront form: sweep s only re-evaluates the features of order s and the part of the
            # conditioner they depend on (same result as `passes` full sweeps, ~5x less work)
            part.refresh(lins)
            for s_ in range(self.passes):
                part.run_inverse_partial(buf, y2, s_)
        else:
            st.refresh(lins)
            for _ in range(self.passes):
                st.run_inverse_sweep(buf, y2)
        return buf[:, :D].reshape(batch + (D,)).contiguous() if buf.shape[1] != D else buf.reshape(batch + (D,))


    def _poly_incremental(self, y: Tensor, want_ladj: bool):
        """SOSPF / BPF layers (uni kinds 5, 6): x = f^{-1}(y) in ONE incremental launch with the bisection of zuko/transforms.py:608-617 in the kernel's group
        epilogue (csrc/inc_inverse.hip: IncSos3x5, IncBern17), or None when the layer is something else / does not fit the aligned-tile plan / autograd is on."""
        lazy, c = self.lazy, self.c
        if not (y.is_cuda and y.dtype == torch.float32 and y.dim() >= 1 and _incremental_enabled() and _partial_inverse_enabled()) or want_ladj:
            return None
        if torch.is_grad_enabled() and (y.requires_grad or (c is not None and c.requires_grad) or any(p.requires_grad for p in lazy.hyper.parameters())):
            return None
        lay = lazy._fusable_layout()
        if lay is None or lay[0].kind not in (5, 6) or os.environ.get("ZUKO_AMD_FULL_SWEEPS", "0") == "1":
            return None
        inc_state = lazy.incremental_state(y.device)
        if inc_state is None:
            return None
        D = lazy.features
        if y.shape[-1] != D:
            return None
        if c is not None:
            yb, cb = broadcast(y, c, ignore=1)
        else:
            yb, cb = y, None
        batch = yb.shape[:-1]
        inc_state.refresh([m for m in lazy.hyper if isinstance(m, MaskedLinear)])
        c2 = None if cb is None else cb.reshape(-1, cb.shape[-1]).contiguous()
        x2, _ = inc_state.run(yb.reshape(-1, D).contiguous(), c2, False)
        return x2.reshape(batch + (D,))

    def _ordered_inverse(self, y: Tensor) -> Tensor:
        """The reference's loop (zuko/transforms.py:994-1000: `passes` times x <- meta(x).inv(y) from x = 0) for layers that have no fused
        inverse kernel (the polynomial maps' bisection, conditioners wider than 256, activations / univariate maps the kernels do not know),
        in WAVEFRONT form: sweep s evaluates the hidden layers once, then only the last layer's rows of the features of order s and only
        their univariate inverse — the features of lower order are final and keep their values, those of higher order are not read by
        anything sweep s computes (their weights are masked), so every feature receives exactly the value the reference's last sweep gives
        it, with 1 / passes of the last layer's and of the inverse map's work per sweep (for SOSPF / BPF: of the bisections)."""
        lazy, c = self.lazy, self.c
        mods = list(lazy.hyper) if isinstance(lazy.hyper, torch.nn.Sequential) else []
        grad = torch.is_grad_enabled() and (y.requires_grad or (c is not None and c.requires_grad) or any(p.requires_grad for p in lazy.hyper.parameters()))
        if lazy.order is None or grad or not mods or type(mods[-1]) is not MaskedLinear or not y.is_cuda or y.dtype not in (torch.float32, torch.float64) or os.environ.get("ZUKO_AMD_FULL_SWEEPS", "0") == "1":
            return super()._inverse(y)  # free-form adjacency / a graph for autograd / a residual conditioner: the loop as the reference writes it
        from .. import ops

        if c is not None:
            yb, cb = broadcast(y, c, ignore=1)
        else:
            yb, cb = y, None
        batch = yb.shape[:-1]
        y2 = yb.reshape(-1, lazy.features)
        c2 = None if cb is None else cb.reshape(-1, cb.shape[-1])

        def linear(h, w, b, m, act_module):
            code = None if act_module is None else _act_code(act_module)
            if act_module is None or code is not None:
                return ops.linear(h, w, b, m, code or 0)
            return act_module(ops.linear(h, w, b, m))

        x2 = wavefront_inverse(lazy, y2, c2, self.passes, linear, lambda phi, ys: lazy.univariate(*unpack(phi, lazy.shapes)).inv(ys))
        return x2.reshape(batch + (lazy.features,))


def wavefront_inverse(lazy: "MaskedAutoregressiveTransform", y2: Tensor, c2: Tensor | None, passes: int, linear, inverse_of, stack=None) -> Tensor:
    """The sweep loop of FusedAutoregressiveTransform._ordered_inverse on y2 [N, D] / c2 [N, C] | None; `linear(h, weight, bias, mask, activation
    module | None)` evaluates act(h (mask * weight)^T + bias) and `inverse_of(phi [N, k, total], y [N, k])` inverts the univariate maps of k
    features — the product passes the HIP kernels (ops.linear, the univariate transform's inv), tests/test_wavefront_inverse.py torch / oracle
    stand-ins to check the schedule on the CPU against the reference's loop.  `stack(modules, h)` evaluates the hidden layers as a whole when
    the conditioner has no per-unit schedule (default: nn.apply_stack, the HIP layer kernels)."""
    from ..nn import apply_stack

    stack = apply_stack if stack is None else stack

    mods = list(lazy.hyper)
    D, total, last = lazy.features, lazy.total, mods[-1]
    dev = y2.device
    x2 = torch.zeros_like(y2)
    units = lazy._sweep_units(dev, passes)
    if units is not None:
        # incremental form: a hidden unit is evaluated ONCE, in the sweep in which its inputs are all final (skinny GEMMs over gathered weight
        # rows, written into persistent activation buffers) — the reference's loop re-evaluates every unit in every sweep and obtains the
        # same number from the same dot product, because the inputs a unit is connected to no longer change
        hbuf = [y2.new_zeros((y2.shape[0], m.weight.shape[0])) for m in mods[0:-1:2]]
        urows = lazy._sweep_unit_rows(dev, passes) if y2.dtype == mods[0].weight.dtype else None
    sweep_no = {}
    for s_, idx in lazy._sweep_features(dev, passes):
        if units is None:
            h = stack(mods[:-1], x2 if c2 is None else torch.cat((x2, c2), dim=-1))
        else:
            h = x2 if c2 is None else torch.cat((x2, c2), dim=-1)
            for l, lin in enumerate(mods[0:-1:2]):
                # (sweeps without features were dropped by _sweep_features: their units join the next sweep that has some)
                first_t = sweep_no.get(l, 0)
                todo = [units[l][t] for t in range(first_t, s_ + 1) if units[l][t] is not None]
                sweep_no[l] = s_ + 1
                if todo:
                    u_ = todo[0] if len(todo) == 1 else torch.cat(todo)
                    if urows is not None:  # the sweeps' rows are consecutive in the gathered copies: slices
                        wr, mr, br, starts = urows[l]
                        r0, r1 = starts[first_t], starts[s_ + 1]
                        w_, m_, bias = wr[r0:r1], mr[r0:r1], None if br is None else br[r0:r1]
                    else:
                        w_, m_, bias = lin.weight.index_select(0, u_), lin.mask.index_select(0, u_), None if lin.bias is None else lin.bias.index_select(0, u_)
                    hbuf[l].index_copy_(1, u_, linear(h, w_, bias, m_, mods[2 * l + 1]))
                h = hbuf[l]
        if isinstance(idx, tuple):  # a run of consecutive features (the usual orders): row / column slices are views, no gather launches
            lo, hi = idx
            rows, k = slice(lo * total, hi * total), hi - lo
            w, b, m, ys = last.weight[rows], None if last.bias is None else last.bias[rows], last.mask[rows], y2[:, lo:hi]
        else:
            rows, k = (idx[:, None] * total + torch.arange(total, device=dev)[None, :]).reshape(-1), idx.numel()
            w, b, m, ys = last.weight.index_select(0, rows), None if last.bias is None else last.bias.index_select(0, rows), last.mask.index_select(0, rows), y2.index_select(1, idx)
        xs = inverse_of(linear(h, w, b, m, None).unflatten(-1, (k, total)), ys)
        if isinstance(idx, tuple):
            x2[:, idx[0] : idx[1]] = xs
        else:
            x2[:, idx] = xs
    return x2


class _FusedInverse(Transform):
    """`FusedAutoregressiveTransform.inv`: the inverse as a Transform of its own (as torch's _InverseTransform), whose
    `call_and_ladj` takes x AND the log-determinant from the single incremental launch."""

    domain = AutoregressiveTransform.codomain
    codomain = AutoregressiveTransform.domain
    bijective = True

    def __init__(self, fwd: FusedAutoregressiveTransform) -> None:
        super().__init__()
        self._fwd = fwd

    @property
    def inv(self) -> Transform:
        return self._fwd

    def _call(self, y: Tensor) -> Tensor:
        return self._fwd._inverse(y)

    def _inverse(self, x: Tensor) -> Tensor:
        return self._fwd._call(x)

    def log_abs_det_jacobian(self, y: Tensor, x: Tensor) -> Tensor:
        return -self._fwd.log_abs_det_jacobian(x, y)

    def call_and_ladj(self, y: Tensor):
        x, ladj = self._fwd.inverse_and_ladj(y)
        return x, -ladj


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
