r"""Conditioner networks of the hot path: `MaskedMLP` (autoregressive) and `MLP` (coupling).

Module trees, parameter names, initialisation order and mask construction match
zuko/nn.py:51-192 (Linear, MLP) and :202-318 (MaskedLinear, MaskedMLP) so that
`load_state_dict` from a reference flow works and `torch.manual_seed(s)` reproduces the
reference's initial weights bit for bit.  The forward pass is one HIP GEMM per layer with the mask
applied in the weight-staging path and bias + activation fused in the epilogue (zk_linear).
"""

from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.nn as nn
from torch import BoolTensor, Tensor

from . import ops

__all__ = ["MLP", "Linear", "MaskedLinear", "MaskedMLP", "Residual"]


def _act_code(module: nn.Module | None) -> int | None:
    """C-ABI activation code for modules the GEMM epilogue can fuse, else None."""
    if module is None:
        return 0
    t = type(module)
    if t is nn.ELU and module.alpha != 1.0:
        return None
    if t is nn.LeakyReLU and module.negative_slope != 0.01:
        return None
    if t is nn.GELU and getattr(module, "approximate", "none") != "none":
        return None
    return ops.ACTIVATIONS.get(t)


class Linear(nn.Module):
    r"""y = x W^T + b with U(-1/sqrt(in), 1/sqrt(in)) init.  Mirrors zuko/nn.py:51-119
    (the `stack=` variant of the reference is outside the hot path and not provided)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, stack: int | None = None) -> None:
        super().__init__()
        if stack is not None:
            raise NotImplementedError("zuko_amd.nn.Linear: stacked operators are outside the hot path")
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        self.in_features = in_features
        self.out_features = out_features
        self.reset_parameters()

    def reset_parameters(self) -> None:
        bound = 1 / self.weight.shape[-1] ** 0.5
        nn.init.uniform_(self.weight, -bound, bound)
        if self.bias is not None:
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}"

    def forward(self, x: Tensor, act: int = 0) -> Tensor:
        return ops.linear(x, self.weight, self.bias, None, act)


class MaskedLinear(nn.Linear):
    r"""y = x (W * A)^T + b for a boolean adjacency A[out, in].  Mirrors zuko/nn.py:202-218;
    the product mask*W is never materialised (the mask gates the weight tile inside the GEMM)."""

    def __init__(self, adjacency: BoolTensor, **kwargs) -> None:
        super().__init__(adjacency.shape[1], adjacency.shape[0], **kwargs)
        self.register_buffer("mask", adjacency)

    def forward(self, x: Tensor, act: int = 0) -> Tensor:
        return ops.linear(x, self.weight, self.bias, self.mask, act)


def apply_stack(mods: Sequence[nn.Module], x: Tensor) -> Tensor:
    """mods applied in order, every (linear, activation) pair as ONE kernel when the activation is one the GEMM epilogue knows."""
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, (Linear, MaskedLinear)):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            code = _act_code(nxt) if nxt is not None and not isinstance(nxt, (Linear, MaskedLinear)) else None
            if code is not None and torch.is_grad_enabled() and code not in (0, 1, 2, 3, 6, 7):
                code = None  # SiLU / GELU need the pre-activation for their derivative: leave them to torch
            if code is not None and nxt is not None:
                x = m(x, code)
                i += 2
                continue
            x = m(x)
        else:
            x = m(x)
        i += 1
    return x


class _FusedSequential(nn.Sequential):
    """Sequential whose (linear, activation) pairs run as ONE kernel when the activation is one
    the GEMM epilogue knows; other modules are applied as-is."""

    def forward(self, x: Tensor) -> Tensor:
        if torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from . import train

            out = train.conditioner(self, x)  # HIP forward + mask-aware dgrad / wgrad (csrc/train.hip) for plain (linear, act)* stacks
            if out is not None:
                return out
        return apply_stack(list(self), x)


class MLP(_FusedSequential):
    r"""Dense multi-layer perceptron (coupling conditioner).  Mirrors zuko/nn.py:122-192
    (`normalize=True` LayerNorm variant is outside the hot path)."""

    def __init__(
        self,
        in_features: int,
        out_features: int,
        hidden_features: Sequence[int] = (64, 64),
        activation: Callable[[], nn.Module] | None = None,
        normalize: bool = False,
        **kwargs,
    ) -> None:
        if normalize:
            raise NotImplementedError("zuko_amd.nn.MLP: normalize=True is outside the hot path")
        activation = nn.ReLU if activation is None else activation
        widths = (in_features, *hidden_features, out_features)
        layers: list[nn.Module] = []
        for i, (a, b) in enumerate(zip(widths[:-1], widths[1:])):
            layers.append(Linear(a, b, **kwargs))
            if i + 2 < len(widths):
                layers.append(activation())
        super().__init__(*layers)
        self.in_features = in_features
        self.out_features = out_features


class Residual(_FusedSequential):
    """x + block(x).  Mirrors zuko/nn.py:195-199."""

    def forward(self, x: Tensor) -> Tensor:
        return x + super().forward(x)


def masked_mlp_masks(adjacency: BoolTensor, hidden_features: Sequence[int], residual_masks: list | None = None) -> list[BoolTensor]:
    r"""Layer masks of a masked MLP whose Jacobian dy_i/dx_j vanishes where adjacency[i, j] is False.

    Follows zuko/nn.py:265-295: outputs with identical dependency sets are merged; a hidden unit is
    tagged with one of the (non-empty) dependency sets, cycling through them; unit u may feed unit v
    iff deps(u) is a subset of deps(v); the last layer maps tags back to output rows."""
    rows, inverse = torch.unique(adjacency, dim=0, return_inverse=True)
    counts = rows.sum(dim=-1)
    subset = (rows.double() @ rows.double().t()) == counts  # subset[i, j]: deps(j) within deps(i)
    masks: list[BoolTensor] = []
    tags = None
    n_hidden = len(hidden_features)
    for depth, width in enumerate((*hidden_features, adjacency.shape[0])):
        full = rows if depth == 0 else subset[:, tags]
        if not full.any():
            raise ValueError("The adjacency matrix leads to a null Jacobian.")
        if depth < n_hidden:
            usable = full.sum(dim=-1).nonzero().squeeze(dim=-1)
            tags = usable[torch.arange(width) % len(usable)]
            masks.append(full[tags])
        else:
            masks.append(full[inverse])
        if residual_masks is not None:
            residual_masks.append(subset[tags, :][:, tags])  # unit -> unit precedence inside the layer
    return masks


def _param_stamp(lins) -> tuple:
    """Identity + version of every weight / bias of `lins`.  In-place updates made through autograd-visible ops
    (optimizers, `copy_`, `load_state_dict`) bump `_version`; writes through `.data` / raw pointers do not — after those,
    call `zuko_amd.invalidate(module)`."""
    return tuple((l.weight._version, l.weight.data_ptr(), -1 if l.bias is None else l.bias._version, 0 if l.bias is None else l.bias.data_ptr()) for l in lins)


def live_tile_masks(mask: Tensor, rows: int = 256, cols: int = 64) -> Tensor:
    """int64 [ceil(out / rows)]: bit k of word p is set iff rows [rows p, rows p + rows) x columns [cols k, cols k + cols) of
    `mask` (bool or numeric [out, in], in % cols == 0, in / cols <= 64) hold a non-zero — the `tile_live_mask` argument of
    zk_linear_bf16 / zk_linear_bf16_rqs (256 x 64 tiles) and of zk_linear_bf16_rqs_lanes (192 x 32)."""
    out_f, in_f = mask.shape
    kt = in_f // cols
    pad = (-out_f) % rows
    dev = mask.device
    mask = mask.detach().cpu()  # (bookkeeping on the host, once per module / parameter version: on the device these were boolean reduce launches in a product trace)
    t = mask != 0
    t = torch.nn.functional.pad(t, (0, 0, 0, pad)) if pad else t
    live = t.reshape(-1, rows, kt, cols).any(dim=3).any(dim=1)  # [panels, kt]
    weights = torch.ones(kt, dtype=torch.int64) << torch.arange(kt, dtype=torch.int64)
    return (live.to(torch.int64) * weights).sum(dim=1).contiguous().to(dev)  # (bit 63 lands in the sign bit: the kernel reads raw bits)


class _Bf16Plan:
    """Device-side tables for running a plain MaskedMLP (linear, activation, linear, ...) in bf16 through
    zk_linear_bf16: hidden units of every layer are reordered by dependency count (a reparametrisation:
    rows of W_l / b_l and columns of W_{l+1} move together, inputs and outputs keep their order), which
    turns the masks of zuko/nn.py:265-295 into block-triangular matrices; `mask * W` is formed once per
    parameter version in that order, and 256 x 64 weight tiles the mask zeroes completely are marked
    dead so the kernel neither fetches nor multiplies them."""

    def __init__(self, lins: Sequence["MaskedLinear"]):
        masks = [l.mask for l in lins]
        dev = masks[0].device
        # the dependency bookkeeping (boolean matrix products over the masks) runs on the HOST, once per module: on the device it would be a
        # library float64 GEMM in a product trace for a few thousand integer operations
        dep = torch.eye(masks[0].shape[1], dtype=torch.float64)
        self.perms: list[Tensor | None] = []
        prev = None
        self.live: list[Tensor | None] = []
        self.masks_p: list[Tensor] = []
        for i, m in enumerate(masks):
            dep = ((m.detach().cpu().double() @ dep) > 0).double()  # [units, inputs]: which inputs each unit can see
            if i + 1 < len(masks):
                perm_c = torch.argsort(dep.sum(dim=1), stable=True)
                dep = dep[perm_c]
                perm = perm_c.to(dev)
            else:
                perm = None  # the outputs keep the reference's column order f * total + j
            mp = m if perm is None else m[perm]
            mp = mp if prev is None else mp[:, prev]
            self.perms.append(perm)
            self.masks_p.append(mp.contiguous())
            out_f, in_f = mp.shape
            if in_f % 64 == 0 and in_f // 64 <= 64:
                self.live.append(live_tile_masks(mp))
            else:
                self.live.append(None)
            prev = perm
        self.version = None
        self.weights: list[Tensor] = []
        self.biases: list[Tensor | None] = []

    def refresh(self, lins: Sequence["MaskedLinear"]) -> None:
        version = _param_stamp(lins)
        if version == self.version:
            return
        self.weights, self.biases = [], []
        prev = None
        for l, perm, mp in zip(lins, self.perms, self.masks_p):
            w = l.weight.detach()
            w = w if perm is None else w[perm]
            w = w if prev is None else w[:, prev]
            self.weights.append((w * mp).contiguous())
            b = None if l.bias is None else l.bias.detach()
            self.biases.append(b if (b is None or perm is None) else b[perm].contiguous())
            prev = perm
        self.version = version

    def spline_panels(self, lins: Sequence["MaskedLinear"], K: int, features: int):
        """(weight_panels, bias_panels, live) of the LAST layer for zk_linear_bf16_rqs: its rows (feature f, parameter j) ->
        panel f // FP, row (f % FP) * (3K - 1) + j, FP = 256 // (3K - 1), zero rows behind.  Cached per parameter version
        (call after `refresh`)."""
        key = (self.version, K, features)
        if self.__dict__.get("_sp_key") != key:
            total = 3 * K - 1
            fp = 256 // total
            panels = -(-features // fp)
            w, b, mp = self.weights[-1], self.biases[-1], self.masks_p[-1]
            in_f = w.shape[1]
            feat = torch.arange(features, device=w.device)
            dst = ((feat // fp) * 256 + (feat % fp) * total).repeat_interleave(total) + torch.arange(total, device=w.device).repeat(features)
            wp = torch.zeros((panels * 256, in_f), dtype=w.dtype, device=w.device)
            wp[dst] = w
            bp = None
            if b is not None:
                bp = torch.zeros(panels * 256, dtype=b.dtype, device=w.device)
                bp[dst] = b
            mpan = torch.zeros((panels * 256, in_f), dtype=torch.bool, device=w.device)
            mpan[dst] = mp.bool()
            live = live_tile_masks(mpan) if in_f // 64 <= 64 else None
            self._sp_key, self._sp = key, (wp, bp, live)
        return self._sp

    def spline_lane_panels(self, lins: Sequence["MaskedLinear"], K: int, features: int):
        """(weight_panels, bias_panels, live) of the LAST layer for zk_linear_bf16_rqs_lanes (include/zuko_amd.h): panels of 192 rows in which
        the 16 outputs a lane of the matrix instruction owns per 32-row block hold the parameters of whole features.  Cached per parameter
        version (call after `refresh`)."""
        key = (self.version, K, features)
        if self.__dict__.get("_lp_key") != key:
            total = 3 * K - 1
            ts = 48 if K == 16 else 24
            fpl = 48 // ts
            fpp = 4 * fpl
            panels = -(-features // fpp)
            w, b, mp = self.weights[-1], self.biases[-1], self.masks_p[-1]
            in_f = w.shape[1]
            dev = w.device
            o = torch.arange(192, device=dev)
            wn, c = o // 96, o % 96
            j, q, kg, t = c // 32, (c % 32) // 8, (c % 8) // 4, c % 4
            slot = 16 * j + 4 * q + t
            feat = torch.arange(panels, device=dev)[:, None] * fpp + (wn * 2 * fpl + kg * fpl + slot // ts)[None, :]  # [panels, 192]
            par = (slot % ts)[None, :].expand_as(feat)
            ok = (par < total) & (feat < features)
            src = (feat * total + par)[ok]  # the reference's row f * total + j (zuko/flows/autoregressive.py:188-190)
            dst = torch.nonzero(ok.reshape(-1)).squeeze(1)
            wp = torch.zeros((panels * 192, in_f), dtype=w.dtype, device=dev)
            wp[dst] = w[src]
            bp = None
            if b is not None:
                bp = torch.zeros(panels * 192, dtype=b.dtype, device=dev)
                bp[dst] = b[src]
            mpan = torch.zeros((panels * 192, in_f), dtype=torch.bool, device=dev)
            mpan[dst] = mp.bool()[src]
            live = live_tile_masks(mpan, 192, 32)
            self._lp_key, self._lp = key, (wp, bp, live)
        return self._lp

    def live_fraction(self) -> list[float]:
        """Fraction of 256 x 64 weight tiles each layer actually multiplies."""
        out = []
        for t, mp in zip(self.live, self.masks_p):
            if t is None:
                out.append(1.0)
                continue
            kt = mp.shape[1] // 64
            bits = (t.unsqueeze(-1) >> torch.arange(kt, device=t.device)) & 1
            out.append(float(bits.double().mean()))
        return out


class MaskedMLP(_FusedSequential):
    r"""Masked MLP (autoregressive conditioner).  Mirrors zuko/nn.py:221-318, including the
    `residual=True` variant (zuko/nn.py:297-309): every layer is followed by a masked residual block
    x + W2 act(W1 x), square hidden-to-hidden layers are replaced by their block; the modules are
    created (and their initialisation drawn) in the reference's order so seeds reproduce its weights."""

    def __init__(
        self,
        adjacency: BoolTensor,
        hidden_features: Sequence[int] = (64, 64),
        activation: Callable[[], nn.Module] | None = None,
        residual: bool = False,
    ) -> None:
        activation = nn.ReLU if activation is None else activation
        out_features, in_features = adjacency.shape
        inner: list = []
        masks = masked_mlp_masks(adjacency, hidden_features, inner if residual else None)
        layers: list[nn.Module] = []
        n_hidden = len(hidden_features)
        for i, m in enumerate(masks):
            layers.append(MaskedLinear(adjacency=m))
            if residual:
                if 0 < i < n_hidden and m.shape[0] == m.shape[1]:
                    layers.pop()  # (its parameters were still drawn, as in the reference)
                layers.append(Residual(MaskedLinear(adjacency=inner[i]), activation(), MaskedLinear(adjacency=inner[i])))
            else:
                layers.append(activation())
        layers.pop()
        super().__init__(*layers)
        self.in_features = in_features
        self.out_features = out_features

    def _bf16_plan(self):
        """Plan of the bf16 fast path, or None when the module tree is not (linear, fusable activation)*."""
        structure = tuple((m.mask._version, m.mask.data_ptr()) for m in self if isinstance(m, MaskedLinear))
        if self.__dict__.get("_bf16_plan_structure") != structure:  # masks overwritten (load_state_dict): rebuild
            self.__dict__.pop("_bf16_plan_cache", None)
            self.__dict__["_bf16_plan_structure"] = structure
        plan = self.__dict__.get("_bf16_plan_cache")
        if plan is None:
            mods = list(self)
            simple = all(isinstance(m, MaskedLinear) == (i % 2 == 0) for i, m in enumerate(mods))
            codes = {_act_code(m) for i, m in enumerate(mods) if i % 2 == 1}
            plan = False
            if simple and len(codes) <= 1 and None not in codes and all(l.in_features % 64 == 0 for l in mods[0::2]):
                plan = _Bf16Plan(mods[0::2])
                plan.act = codes.pop() if codes else 0
            self.__dict__["_bf16_plan_cache"] = plan
        return plan or None

    def _apply(self, fn, *args, **kwargs):  # device / dtype moves invalidate the bf16 tables
        self.__dict__.pop("_bf16_plan_cache", None)
        return super()._apply(fn, *args, **kwargs)

    def bf16_rqs(self, inp: Tensor, x: Tensor, K: int, bound: float, slope: float):
        """bf16 autoregressive spline layer with phi kept on chip: hidden layers through zk_linear_bf16, last layer +
        spline + feature sum through zk_linear_bf16_rqs.  `inp` [N, in] conditioner input, x [N, D] transform input.
        Returns (y, ladj) or None when the module has no bf16 plan."""
        plan = self._bf16_plan()
        if plan is None:
            return None
        lins = list(self)[0::2]
        plan.refresh(lins)
        h = inp
        for w, b, live in zip(plan.weights[:-1], plan.biases[:-1], plan.live[:-1]):
            h = ops.linear_bf16(h, w, b, live, plan.act)
        import os

        if os.environ.get("ZUKO_AMD_BF16_PANELS256", "0") != "1" and h.shape[-1] % 64 == 0 and h.shape[-1] <= 2048:  # second-generation kernel (lane-owned features)
            wp, bp, lv = plan.spline_lane_panels(lins, K, x.shape[-1])
            return ops.linear_bf16_rqs(h, wp, bp, lv, x, K, bound, slope, lanes=True)
        wp, bp, lv = plan.spline_panels(lins, K, x.shape[-1])
        return ops.linear_bf16_rqs(h, wp, bp, lv, x, K, bound, slope)

    def forward(self, x: Tensor) -> Tensor:
        if x.dtype == torch.bfloat16 and x.is_cuda and not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            plan = self._bf16_plan()
            if plan is not None:
                lins = list(self)[0::2]
                plan.refresh(lins)
                h = x
                for i, (w, b, live) in enumerate(zip(plan.weights, plan.biases, plan.live)):
                    h = ops.linear_bf16(h, w, b, live, plan.act if i + 1 < len(lins) else 0)
                return h
        return super().forward(x)
