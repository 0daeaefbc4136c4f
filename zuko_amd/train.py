r"""Training path of the conditioner networks: `phi = hyper(x)` with HIP forward, dgrad and wgrad.

The reference trains through `F.linear(x, mask * W, b)` (zuko/nn.py:217-218) with PyTorch autograd: per layer one dense
GEMM forward and two dense GEMMs backward, all of which spend half their time on weights the mask holds at zero.  Here a
plain (linear, activation)* network — `MaskedMLP` (autoregressive) or `MLP` (coupling) — is evaluated in a
reparametrisation whose hidden units are sorted by the size of their dependency set (rows of W_l / b_l and columns of
W_{l+1} permuted together: the function is unchanged), which makes the masks block lower-triangular; the three GEMMs of
every layer then skip the blocks the mask zeroes (csrc/train.hip):

    forward   h_{l+1} = act(h_l Ws_l^T + bs_l)              zk_gemm_f32_skip, (128 x 32) tile skipping
    dgrad     g_l     = (g_{l+1} Ws_l) * act'(h_l)           the same kernel on Ws_l^T, derivative in the epilogue
    wgrad     dWs_l   = mask * (g_{l+1}^T h_l), dbs_l = colsum(g_{l+1})   zk_wgrad_f32 (split-K, live 128 x 128 blocks), zk_colsum_f32

Only integer bookkeeping (permutations, gather / scatter indices, skip maps) is done with torch ops, once per module.
"""

from __future__ import annotations

import ctypes
import os

import torch
from torch.autograd.function import once_differentiable
from torch import Tensor

from . import _C

import weakref

_PLAN_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()
TRAIN_ACTS = (0, 1, 2, 3, 6, 7)  # activations whose derivative is a function of their output (as zk_act_backward)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream() -> int:
    return _C.stream()


class SortedPlan:
    """Index tables of one (linear, activation)* network on one device."""

    def __init__(self, lins, act_code: int, device: torch.device) -> None:
        self.act = act_code
        self.device = device
        n = len(lins)
        masks = [getattr(l, "mask", None) for l in lins]
        shapes = [tuple(l.weight.shape) for l in lins]
        # dependency-count sort of the hidden layers (identity for dense networks)
        perms: list[Tensor | None] = []
        if all(m is not None for m in masks):
            dep = torch.eye(shapes[0][1], dtype=torch.float64)
            for i, m in enumerate(masks):
                dep = ((m.detach().cpu().double() @ dep) > 0).double()
                if i + 1 < n:
                    # (dep stays in the module's unit order: the next mask's columns index it)
                    perms.append(torch.argsort(dep.sum(dim=1), stable=True))
                else:
                    perms.append(None)
        else:
            perms = [None] * n
        self.perms = perms
        self.idx_w, self.idx_wt, self.idx_b, self.kskip_f, self.kskip_b, self.pairs, self.mask_s, self.shapes = [], [], [], [], [], [], [], shapes
        self.mask_u8 = []
        self.cs_flag = []
        self.cols_dev = []  # sorted column -> module column of every layer (idx_b holds the rows)
        self.rows_cpu, self.cols_cpu, self.mask_s_cpu = [], [], []  # host copies (tables of the dgrad-chain kernel)
        prev = None
        for i, (out_f, in_f) in enumerate(shapes):
            rows = perms[i] if perms[i] is not None else torch.arange(out_f)
            cols = prev if prev is not None else torch.arange(in_f)
            if out_f * in_f >= 2**31:
                raise ValueError(f"zuko_amd: layer {i} has {out_f} x {in_f} >= 2^31 weights: the sorted-domain gather tables are int32")
            idx = (rows[:, None] * in_f + cols[None, :]).to(torch.int32)  # sorted [out, in] -> position in W.flatten()
            m = masks[i]
            ms = torch.ones((out_f, in_f), dtype=torch.bool) if m is None else m.detach().cpu().bool()[rows][:, cols]
            self.idx_w.append(idx.reshape(-1).contiguous().to(device))
            self.idx_wt.append(idx.t().reshape(-1).contiguous().to(device))
            self.idx_b.append(rows.to(torch.int32).to(device))
            self.cols_dev.append(cols.to(torch.int32).to(device))
            self.rows_cpu.append(rows.numpy().copy()), self.cols_cpu.append(cols.numpy().copy()), self.mask_s_cpu.append(None if m is None else ms.numpy().copy())
            kf, kb = (None, None) if m is None else (self._kskip(ms), self._kskip(ms.t().contiguous()))
            self.kskip_f.append(None if kf is None else kf.to(device))
            self.kskip_b.append(None if kb is None else kb.to(device))
            ob, ib = -(-out_f // 128), -(-in_f // 128)
            pad = torch.zeros((ob * 128, ib * 128), dtype=torch.bool)
            pad[:out_f, :in_f] = ms
            live = pad.reshape(ob, 128, ib, 128).any(dim=3).any(dim=1)
            pr = live.nonzero().to(torch.int32).contiguous()
            self.pairs.append(pr.to(device))
            # one designated pair per out block also sums the columns of g (bias gradient); needs every out block present
            flag = torch.zeros(pr.shape[0], dtype=torch.uint8)
            seen = set()
            for k in range(pr.shape[0]):
                o = int(pr[k, 0])
                if o not in seen:
                    seen.add(o)
                    flag[k] = 1
            self.cs_flag.append(flag.to(device) if len(seen) == ob else None)
            self.mask_s.append(None if m is None else ms.to(torch.uint8).contiguous().to(device))
            self.mask_u8.append(None if m is None else m.detach().to(device=device, dtype=torch.uint8).contiguous())
            prev = perms[i]
        self.idx_b64 = [t.long() for t in self.idx_b]
        self.kept = [float(k.shape[0]) / (-(-s[0] // 128) * -(-s[1] // 128)) for k, s in zip(self.pairs, shapes)]

    @staticmethod
    def _kskip(ms: Tensor):
        """int64 word per 128-row block (raw bits of a uint64): bit kt set iff columns [32 kt, 32 kt + 32) of the block hold a
        non-zero; None (no skipping) for layers with more than 2048 inputs."""
        import numpy as np

        out_f, in_f = ms.shape
        if in_f > 2048:
            return None
        ob, kt = -(-out_f // 128), -(-in_f // 32)
        pad = np.zeros((ob * 128, kt * 32), dtype=bool)
        pad[:out_f, :in_f] = ms.numpy()
        live = pad.reshape(ob, 128, kt, 32).any(axis=3).any(axis=1)  # [ob, kt]
        words = (live.astype(np.uint64) << np.arange(kt, dtype=np.uint64)[None, :]).sum(axis=1, dtype=np.uint64)
        return torch.from_numpy(words.view(np.int64).copy())

    # ---- device work -------------------------------------------------------------------------------------------

    def gather(self, lins, forward: bool = True, transposes=None):
        """Sorted, masked weights Ws_l [out, in], their transposes [in, out] and sorted biases (fresh tensors).  forward=False:
        only what the backward pass reads (the transposes; ws entries are None, biases are reported as present / absent).
        transposes: the layers whose transpose is wanted (None = all; the others come back as None)."""
        lib = _C.lib()
        ws, wts, bs = [], [], []
        for l, lin in enumerate(lins):
            out_f, in_f = self.shapes[l]
            w = lin.weight.detach().contiguous()
            if transposes is None or l in transposes:
                wt_l = torch.empty((in_f, out_f), dtype=torch.float32, device=self.device)
                _C.check(lib.zk_gather_f32(_ptr(w), _ptr(self.mask_u8[l]), _ptr(self.idx_wt[l]), out_f * in_f, _ptr(wt_l), _stream()), "zk_gather_f32")
                wts.append(wt_l)
            else:
                wts.append(None)
            if not forward:
                ws.append(None)
                bs.append(None if lin.bias is None else True)
                continue
            ws_l = torch.empty((out_f, in_f), dtype=torch.float32, device=self.device)
            _C.check(lib.zk_gather_f32(_ptr(w), _ptr(self.mask_u8[l]), _ptr(self.idx_w[l]), out_f * in_f, _ptr(ws_l), _stream()), "zk_gather_f32")
            ws.append(ws_l)
            if lin.bias is None:
                bs.append(None)
            else:
                b = torch.empty(out_f, dtype=torch.float32, device=self.device)
                _C.check(lib.zk_gather_f32(_ptr(lin.bias.detach().contiguous()), None, _ptr(self.idx_b[l]), out_f, _ptr(b), _stream()), "zk_gather_f32")
                bs.append(b)
        return ws, wts, bs

    def gemm(self, x: Tensor, w: Tensor, kskip, bias, act: int, gate: Tensor | None = None, gate_act: int = 0) -> Tensor:
        out_f, in_f = w.shape
        y = torch.empty((x.shape[0], out_f), dtype=torch.float32, device=x.device)
        err = _C.lib().zk_gemm_f32_skip(x.shape[0], in_f, out_f, _ptr(x), x.stride(0), _ptr(w), _ptr(kskip), _ptr(bias), act,
                                        _ptr(gate), 0 if gate is None else gate.stride(0), gate_act, _ptr(y), out_f, _stream())
        _C.check(err, "zk_gemm_f32_skip")
        return y

    def wgrad(self, l: int, g: Tensor, h: Tensor, want_bias: bool = False):
        """Weight gradient dW_l [out, in] in the MODULE's unit order (zeros where the mask is false): computed on the sorted-domain
        operands, written through the row / column permutations by the reduction kernel.  want_bias: returns (dW_l, db_l) with the bias
        gradient from the same pass over g, ALSO in the module's unit order (zk_wgrad_bias_f32 writes db[rows[o]]; assign it as it is —
        None when some 128-row out block has no live weight block: the caller then takes the column-sum path)."""
        lib = _C.lib()
        out_f, in_f = self.shapes[l]
        N = g.shape[0]
        pairs = self.pairs[l]
        npairs = pairs.shape[0]
        dw = torch.zeros((out_f, in_f), dtype=torch.float32, device=g.device)
        ns = lib.zk_wgrad_slices(N, npairs)
        partial = torch.empty(max(1, ns) * npairs * 128 * 128, dtype=torch.float32, device=g.device)
        if want_bias and self.cs_flag[l] is not None and npairs > 0:
            cs_partial = torch.empty(max(1, ns) * (-(-out_f // 128) * 128), dtype=torch.float32, device=g.device)
            db = torch.zeros(out_f, dtype=torch.float32, device=g.device)  # (zeros: the kernel returns early on an empty batch, the reference yields zeros)
            err = lib.zk_wgrad_bias_f32(N, out_f, in_f, _ptr(g), g.stride(0), _ptr(h), h.stride(0), _ptr(pairs), npairs, _ptr(partial), _ptr(self.mask_s[l]), _ptr(dw), 0,
                                        _ptr(self.cs_flag[l]), _ptr(cs_partial), _ptr(db), _ptr(self.idx_b[l]), _ptr(self.cols_dev[l]), _stream())
            _C.check(err, "zk_wgrad_bias_f32")
            return dw, db
        err = lib.zk_wgrad_f32(N, out_f, in_f, _ptr(g), g.stride(0), _ptr(h), h.stride(0), _ptr(pairs), npairs, _ptr(partial), _ptr(self.mask_s[l]), _ptr(dw), 0,
                               _ptr(self.idx_b[l]), _ptr(self.cols_dev[l]), _stream())
        _C.check(err, "zk_wgrad_f32")
        return (dw, None) if want_bias else dw

    def wgrad_multi(self, items, packed_last=None, amax=None):
        """Weight and bias gradients of several layers in two launches (zk_wgrad_multi).  items: [(layer, g, h)] with every layer's
        cs_flag present; returns {layer: (dW in the module's order, db in the module's order)}.  packed_last (PackedRows): the LAST layer's g
        has its columns in the fused kernels' packed order (padding slots included) — the row tables of that order are used for it."""
        import ctypes

        lib = _C.lib()
        cls = _C.STRUCTS["zk_wgrad_layer_v1"]
        n = len(items)
        N = items[0][1].shape[0]
        dev = items[0][1].device
        last = len(self.shapes) - 1

        def side(l):  # (width of g, live pairs, sorted-domain mask, column-sum flags, row table)
            if packed_last is not None and l == last:
                return packed_last.width, packed_last.pairs, packed_last.mask_s, packed_last.cs_flag, packed_last.rows
            return self.shapes[l][0], self.pairs[l], self.mask_s[l], self.cs_flag[l], self.idx_b[l]

        sides = [side(l) for l, _, _ in items]
        sizes_w = [self.shapes[l][0] * self.shapes[l][1] for l, _, _ in items]
        sizes_b = [self.shapes[l][0] for l, _, _ in items]
        ns = [_wgrad_slices(N, sd[1].shape[0]) for sd in sides]
        sizes_p = [ns[i] * sd[1].shape[0] * 128 * 128 for i, sd in enumerate(sides)]
        sizes_c = [ns[i] * (-(-sd[0] // 128) * 128) for i, sd in enumerate(sides)]
        flat = torch.zeros(sum(sizes_w) + sum(sizes_b), dtype=torch.float32, device=dev)  # (dW: only the live blocks are written)
        work = torch.empty(sum(sizes_p) + sum(sizes_c), dtype=torch.float32, device=dev)
        arr = (cls * n)()
        out = {}
        ow, ob, op, oc = 0, sum(sizes_w), 0, sum(sizes_p)
        for i, (l, g, h) in enumerate(items):
            out_f, in_f = self.shapes[l]
            width, pairs, mask_s, cs_flag, rows = sides[i]
            dw, db = flat[ow : ow + sizes_w[i]].view(out_f, in_f), flat[ob : ob + sizes_b[i]]
            d = arr[i]
            d.struct_size = ctypes.sizeof(cls)
            d.out_features, d.in_features, d.npairs, d.ldg, d.ldh = width, in_f, pairs.shape[0], g.stride(0), h.stride(0)
            for name, t in (("g", g), ("h", h), ("pairs", pairs), ("partial", work[op:]), ("mask", mask_s), ("dw", dw), ("cs_flag", cs_flag),
                            ("cs_partial", work[oc:]), ("db", db), ("rows", rows), ("cols", self.cols_dev[l])):
                setattr(d, name, None if t is None else t.data_ptr())
            if amax is not None:  # {layer: (maxima of |g|, of |h|)} on the device: two-part f16 products (csrc/train.hip: wgrad_split_body<true>)
                d.g_amax, d.h_amax = amax[l][0].data_ptr(), amax[l][1].data_ptr()
            out[l] = (dw, db)
            ow += sizes_w[i]; ob += sizes_b[i]; op += sizes_p[i]; oc += sizes_c[i]
        _C.check(lib.zk_wgrad_multi(n, ctypes.cast(arr, ctypes.c_void_p), N, _stream()), "zk_wgrad_multi")
        return out

    def colsum(self, g: Tensor) -> Tensor:
        lib = _C.lib()
        N, C = g.shape
        ws = torch.empty(lib.zk_colsum_slices(max(N, 1)) * C, dtype=torch.float32, device=g.device)
        out = torch.empty(C, dtype=torch.float32, device=g.device)
        _C.check(lib.zk_colsum_f32(N, C, _ptr(g), g.stride(0), _ptr(ws), _ptr(out), 0, _stream()), "zk_colsum_f32")
        return out


_SLICES: dict = {}


def _wgrad_slices(N: int, npairs: int) -> int:
    """max(1, zk_wgrad_slices(N, npairs)), remembered (a pure function of its arguments; four library calls per transform and step otherwise)."""
    key = (N, npairs)
    v = _SLICES.get(key)
    if v is None:
        v = _SLICES[key] = max(1, _C.lib().zk_wgrad_slices(N, npairs))
    return v


class PackedRows:
    """Output-side tables of the LAST layer's weight gradient when g_phi comes in the fused kernels' packed order (csrc/zk_ar_common.h:
    ArArgs::phi_packed): column u of g is row mod_row[u] of the weight (-1: a padding slot, always zero, no destination)."""

    def __init__(self, plan: SortedPlan, mod_row, device) -> None:
        import numpy as np

        last = len(plan.shapes) - 1
        out_f, in_f = plan.shapes[last]
        mod_row = np.asarray(mod_row, dtype=np.int64)
        self.width = int(mod_row.shape[0])
        ms_mod = plan.mask_s_cpu[last]  # [out (module order: the last layer's rows are not sorted), in (sorted)]
        ms = np.where((mod_row >= 0)[:, None], ms_mod[np.maximum(mod_row, 0)], False)
        ob, ib = -(-self.width // 128), -(-in_f // 128)
        pad = np.zeros((ob * 128, ib * 128), dtype=bool)
        pad[: self.width, :in_f] = ms
        live = torch.from_numpy(pad.reshape(ob, 128, ib, 128).any(axis=3).any(axis=1))
        pr = live.nonzero().to(torch.int32).contiguous()
        flag = torch.zeros(pr.shape[0], dtype=torch.uint8)
        seen = set()
        for k in range(pr.shape[0]):
            o = int(pr[k, 0])
            if o not in seen:
                seen.add(o)
                flag[k] = 1
        self.ok = len(seen) == ob and pr.shape[0] > 0
        self.pairs, self.cs_flag = pr.to(device), flag.to(device)
        self.mask_s = torch.from_numpy(ms.astype(np.uint8)).contiguous().to(device)
        self.rows = torch.from_numpy(mod_row.astype(np.int32)).to(device)


_FUSED_TRAIN = weakref.WeakKeyDictionary()  # SortedPlan -> {(features, total) as passed: FusedAR (static-shape kernel) or False}


def _fused_forward_state(plan: "SortedPlan", lins, device, rows: int = 0, features: int | None = None, total: int | None = None):
    """The static-shape fused kernel (csrc/fused_ar_static.hip, training instantiation) for the forward of this network under autograd, or
    None: the conditioner of MaskedAutoregressiveTransform(features, hidden_features=[..up to three..]) with a spline or affine head.  Its
    hidden activations come out in the kernel's sorted unit order, which is this plan's (both sort stably by dependency count) — checked once
    against the layer-wise kernels on a random batch.  A kernel that is not on disk yet is compiled once the batches (`rows`, this call's)
    pass the JIT threshold, as for inference (zuko_amd/static_ar.py: effective_rows)."""
    import os

    if os.environ.get("ZUKO_AMD_NO_FUSED_TRAIN", "0") == "1":
        return None
    per_plan = _FUSED_TRAIN.setdefault(plan, {})  # keyed by what the head was said to be: a call that cannot know (features=None) must not decide for one that does
    key = (features, total)
    st = per_plan.get(key)
    if st is None:
        st = False
        try:
            from . import fused

            shapes = plan.shapes
            din = shapes[0][1]
            if features is None:
                # (not told: no context — the conditioner's inputs are the features; affine head = 2, 8-bin spline head = 23 parameters per feature)
                # (also the polynomial heads, round 6: 16 = shifted SOS, 17 = bounded Bernstein — their conditioner-only forward, the maps keep their own autograd node)
                total = {2 * din: 2, 23 * din: 23, 16 * din: 16, 17 * din: 17}.get(shapes[-1][0], 0)
                features = din
            elif total not in (2, 23, 16, 17) or features * total != shapes[-1][0] or features > din:
                total = 0
            layout = {2: fused.uni_layout("affine", 2), 23: fused.uni_layout("rqs", 23, 8), 16: fused.uni_layout("sos", 16), 17: fused.uni_layout("bern", 17)}.get(total) or fused.uni_layout("affine", 2)
            ok = (total and 2 <= len(lins) <= 4 and plan.act == 1 and din % 4 == 0
                  and all(getattr(l, "mask", None) is not None for l in lins) and all(s[0] % 16 == 0 and s[0] <= fused.MAX_WIDTH for s in shapes[:-1]))
            if ok:
                fp = fused.build_plan([l.mask for l in lins], features, layout)
                if fp is not None:
                    st = fused.FusedAR(fp, device, 1, 1.0, 1e-3)
                    st._train_checked = None  # None: not validated yet; True / False afterwards
        except Exception:
            st = False
        per_plan[key] = st
    if not st:
        return None
    if rows > 0 and (st.static is None or not st.static[0].meta.get("split")):
        held = st.static
        st.ready(rows)  # (compiles the operand-split kernel when the batches seen so far justify it)
        if st.static is not held:
            st._train_checked = None
    if st.static is None or not st.static[0].meta["TRAIN_OK"] or st._train_checked is False:
        return None
    if st._train_checked is None:
        try:
            shapes = plan.shapes
            st.refresh(lins)
            g = torch.Generator(device="cpu").manual_seed(0)
            xt = torch.randn(200, shapes[0][1], generator=g).to(device)
            with torch.no_grad():
                hs_f, phi_f = _fused_forward(st, xt, shapes[-1][0])
                ws, _, bs = plan.gather(lins)
                h = xt
                same = True
                n = len(lins)
                for l in range(n):
                    h = plan.gemm(h, ws[l], plan.kskip_f[l], bs[l], plan.act if l + 1 < n else 0)
                    ref = hs_f[l] if l + 1 < n else phi_f
                    same = same and bool(torch.allclose(ref, h, rtol=1e-4, atol=1e-4))
            st._train_checked = same
        except Exception:
            st._train_checked = False
        if not st._train_checked:
            return None
    return st


def _fused_forward(st, x: Tensor, out_features: int, uni=None, packed_width: int = 0, amax=None):
    """([h_1, ...] sorted-domain hidden activations, phi) from one launch of zk_ar_forward_train (a static-shape kernel of
    zuko_amd/static_ar.py in its training instantiation).  uni = (bound, slope) (operand-split kernels): the launch also evaluates the
    univariate map, and (hs, phi, y, ladj) is returned.  packed_width: phi [N, packed_width] in the kernels' packed order instead."""
    p = st.plan
    N = x.shape[0]
    hs = [torch.empty((N, w), dtype=torch.float32, device=x.device) for w in p.widths]
    phi = torch.empty((N, packed_width or out_features), dtype=torch.float32, device=x.device)
    kern, rev = st.static
    hp = [_ptr(h) for h in hs] + [None] * (3 - len(hs))
    extra = {}
    if amax is not None:  # [maxima of x, h_1, h_2, ..] (device, zeroed): the launch folds max |x|, max |h_l| into them (operand-split kernels)
        extra.update({f"amax{i}": _ptr(t) for i, t in enumerate(amax[1:4])})
        extra["amax3"] = _ptr(amax[0])  # (x)
    if uni is not None:
        y, ladj = torch.empty((N, p.features), dtype=torch.float32, device=x.device), torch.empty(N, dtype=torch.float32, device=x.device)
        extra.update(dict(y=_ptr(y), ldy=y.stride(0), ladj=_ptr(ladj), bound=float(uni[0]), slope=float(uni[1])))
    a = _C.args("zk_ar_args_v1", launcher=kern.launcher, rev=rev, uni_kind=p.layout.kind, N=N, D=p.features, DIN=x.shape[1], x=_ptr(x), ldx=x.stride(0), h1=hp[0], h2=hp[1], h3=hp[2],
                phi=_ptr(phi), ldphi=phi.stride(0), phi_packed=int(packed_width > 0), wstream=_ptr(st.fine_stream), bias=_ptr(st.bias), bias_floats=st.bias_floats, featmap=_ptr(st.featmap),
                n_layers=p.n_layers, n_groups=p.n_groups, n_chunks=st.fine_n_chunks, act=1, **extra)
    err = _C.lib().zk_ar_forward_train(a, _stream())
    _C.check(err, "zk_ar_forward_train")
    return (hs, phi) if uni is None else (hs, phi, y, ladj)


class DgradChain:
    """Backward of the linear layers in one launch of a generated kernel whose weight stream holds the TRANSPOSED sorted masked weights.
    `full` (operand-split kernel, csrc/fused_ar_split_impl.h: arxd_kernel, C ABI zk_ar_dgrad_full): every layer, from the gradient of the
    packed parameters; otherwise (csrc/fused_ar_static_impl.h: ars_dgrad_kernel, zk_ar_dgrad_chain) every layer but the last, whose
    dgrad stays a stand-alone GEMM."""

    def __init__(self, plan: SortedPlan, kernel, tables: dict, gathers, device) -> None:
        self.kernel, self.t = kernel, tables
        self.full = tables.get("chain") in (2, 3)
        self.fused = tables.get("chain") == 3  # the transform's whole backward (univariate adjoint + chain): run_backward
        self.packed = PackedRows(plan, tables["MODROW"], device) if self.fused else None  # phi / g_phi travel in the packed order then
        self.idx = [torch.from_numpy(g).to(device) for g in gathers]
        self.offsets = [b * 256 for b in tables["BASE"]]
        self.device = device

    def gather(self, plan: SortedPlan, lins) -> Tensor:
        """The kernel's weight stream from the module's CURRENT parameters (a fresh tensor: the forward saves it for its backward)."""
        stream = torch.empty((self.t["STREAM_IMAGES"] if self.full else self.t["NCHUNK"] * 24) * 256, dtype=torch.float32, device=self.device)
        n1 = len(self.idx)
        items = []
        for c, idx in enumerate(self.idx):
            l = n1 - 1 - c
            w = lins[l].weight.detach().contiguous()
            items.append((w, plan.mask_u8[l], idx, idx.numel() // 512 if self.full else idx.numel(), stream[self.offsets[c] :], 1 if self.full else 0))
        _C.gather_multi(items, _stream())
        return stream

    def run(self, plan: SortedPlan, stream: Tensor, g_in: Tensor, hs, gx_add: Tensor | None = None):
        """g_in: gradient of the packed parameters [N, out_features] (full) or of the last hidden layer's pre-activations [N, width];
        hs = [x, h_1, ..] as saved by the forward.  Returns ([g_1, .., g_{n-1}] gradients of the hidden pre-activations, gx).
        gx_add (full chains): a [N, in] contiguous tensor the input gradient is ADDED to in place (and which is returned as gx)."""
        n = len(plan.shapes)
        N = g_in.shape[0]
        dev = g_in.device
        n_out = n - 1 if self.full else n - 2
        gs = [torch.empty((N, plan.shapes[l][0]), dtype=torch.float32, device=dev) for l in range(n_out)]
        gx = torch.empty((N, plan.shapes[0][1]), dtype=torch.float32, device=dev) if gx_add is None else gx_add
        hp = [_ptr(hs[1 + l]) for l in range(n_out)] + [None] * 3
        gp = [_ptr(g) for g in gs] + [None] * 3
        a = _C.args("zk_ar_args_v1", launcher=self.kernel.launcher, N=N, D=plan.shapes[0][1], DIN=g_in.shape[1], x=_ptr(g_in), ldx=g_in.stride(0), h1=hp[0], h2=hp[1], h3=hp[2],
                    gh1=gp[0], gh2=gp[1], gh3=gp[2], y=_ptr(gx), ldy=gx.stride(0), wstream=_ptr(stream), n_layers=n, n_chunks=self.t["NCHUNK"], act=1,
                    accumulate=int(gx_add is not None and self.full))
        if self.full:
            _C.check(_C.lib().zk_ar_dgrad_full(a, _stream()), "zk_ar_dgrad_full")
            return gs, gx
        _C.check(_C.lib().zk_ar_dgrad_chain(a, _stream()), "zk_ar_dgrad_chain")
        return gs + [g_in], gx


    def run_backward(self, plan: SortedPlan, stream: Tensor, st, uni, x: Tensor, phi: Tensor, gy: Tensor, gl: Tensor, hs, amax=None):
        """Fused chains: (g_phi [N, packed width] in the packed order of phi, [g_1, ..], gx) from d loss / d (y, ladj) in one launch
        (zk_ar_backward_full); phi: the forward's, packed; st = the forward's FusedAR (feature grouping), uni = (kind, bound, slope, ..)."""
        n = len(plan.shapes)
        N = x.shape[0]
        dev = x.device
        gs = [torch.empty((N, plan.shapes[l][0]), dtype=torch.float32, device=dev) for l in range(n - 1)]
        gx = torch.empty((N, plan.shapes[0][1]), dtype=torch.float32, device=dev)
        gphi = torch.empty_like(phi)
        hp = [_ptr(hs[1 + l]) for l in range(n - 1)] + [None] * 3
        gp = [_ptr(g) for g in gs] + [None] * 3
        a = _C.args("zk_ar_args_v1", launcher=self.kernel.launcher, uni_kind=uni[0], N=N, D=st.plan.features, DIN=x.shape[1], x=_ptr(x), ldx=x.stride(0), h1=hp[0], h2=hp[1], h3=hp[2],
                    gh1=gp[0], gh2=gp[1], gh3=gp[2], y=_ptr(gx), ldy=gx.stride(0), y_in=_ptr(gy), ldo=gy.stride(0), ladj=_ptr(gl), phi=_ptr(phi), x_out=_ptr(gphi), ldphi=phi.stride(0),
                    wstream=_ptr(stream), featmap=_ptr(st.featmap), n_layers=n, n_groups=st.plan.n_groups, n_chunks=self.t["NCHUNK"], act=1, bound=float(uni[1]), slope=float(uni[2]),
                    **({} if amax is None else {"amax3": _ptr(amax["gphi"]), **{f"amax{c}": _ptr(amax["g"][n - 2 - c]) for c in range(n - 1)}}))
        _C.check(_C.lib().zk_ar_backward_full(a, _stream()), "zk_ar_backward_full")
        return gphi, gs, gx


_CHAINS = weakref.WeakKeyDictionary()  # SortedPlan -> DgradChain or False
_BACKWARDS = weakref.WeakKeyDictionary()  # SortedPlan -> DgradChain (fused) or False


def _backward_kernel(plan: SortedPlan, st, rows: int):
    """The one-launch backward of the transform (DgradChain with .fused) for the forward state `st`, or None (kernel not built and compiling
    not allowed for this batch, ZUKO_AMD_NO_FUSED_AR_BACKWARD=1)."""
    import os

    if os.environ.get("ZUKO_AMD_NO_FUSED_AR_BACKWARD", "0") == "1":
        return None
    bk = _BACKWARDS.get(plan)
    if bk is None:
        from . import static_ar

        lay = st.plan.layout
        tg = _memo(plan, ("packed", lay.kind, lay.nt, lay.fpl, lay.total), lambda: static_ar.chain_split_tables(
            plan.mask_s_cpu, plan.rows_cpu, plan.cols_cpu, packed={"uni": lay.kind, "featmap": st.plan.featmap, "nt": lay.nt, "fpl": lay.fpl, "total": lay.total}))
        if tg is None:
            _BACKWARDS[plan] = False
            return None
        kern = static_ar.chain_kernel(tg[0], allow_compile=static_ar.jit_enabled() and static_ar.effective_rows(st, rows, count=False) >= static_ar.jit_min_rows())  # (st.ready counted this step's rows)
        if kern is None:
            return None  # (not cached: a later, larger batch may be allowed to compile)
        bk = DgradChain(plan, kern, tg[0], tg[1], plan.device)
        if not bk.packed.ok:  # (some 128-column block of the packed gradient without a live weight block: the bias gradient could not ride on the weight pass)
            bk = False
        _BACKWARDS[plan] = bk
    return bk or None


def _memo(plan, key, make):
    """Tables derived from a plan's masks alone (numpy work + digests), computed once per plan: the kernel lookups below run on every
    forward while a kernel is still waiting for its JIT threshold."""
    cache = plan.__dict__.setdefault("_table_memo", {})
    if key not in cache:
        cache[key] = make()
    return cache[key]


def _dgrad_chain(plan: SortedPlan, lins, rows: int):
    """The dgrad-chain kernel of this network, or None (not a masked ReLU conditioner of 2-4 layers with hidden widths that are
    multiples of 16 up to 256, kernel not built and compiling not allowed, ZUKO_AMD_NO_DGRAD_CHAIN=1)."""
    import os

    if os.environ.get("ZUKO_AMD_NO_DGRAD_CHAIN", "0") == "1":
        return None
    st = _CHAINS.get(plan)
    if st is None:
        st = False
        n = len(lins)
        if 2 <= n <= 4 and plan.act == 1 and all(m is not None for m in plan.mask_s_cpu):
            from . import static_ar

            allow = static_ar.jit_enabled() and static_ar.effective_rows(plan, rows) >= static_ar.jit_min_rows()
            if static_ar.split_enabled() and plan.shapes[-1][0] % 4 == 0:  # every layer in one launch of the operand-split kernel
                tg = _memo(plan, "split", lambda: static_ar.chain_split_tables(plan.mask_s_cpu, plan.rows_cpu, plan.cols_cpu))
                kern = static_ar.chain_kernel(tg[0], allow_compile=allow) if tg is not None else None
                if kern is not None:
                    st = DgradChain(plan, kern, tg[0], tg[1], plan.device)
                    _CHAINS[plan] = st
                    return st
                split_pending = tg is not None and not allow and static_ar.jit_enabled()  # a later, larger batch may compile the one-launch split chain
            else:
                split_pending = False
            tg = _memo(plan, "f32", lambda: static_ar.chain_tables(plan.mask_s_cpu[: n - 1], plan.rows_cpu[: n - 1], plan.cols_cpu[: n - 1]))
            if tg is not None:
                held = getattr(plan, "_f32_chain", None)  # (the un-cached f32 chain of earlier small batches: its index uploads are made once)
                kern = held.kernel if held is not None else static_ar.chain_kernel(tg[0], allow_compile=allow)
                if kern is not None:
                    st = held if held is not None else DgradChain(plan, kern, tg[0], tg[1], plan.device)
                    if split_pending:
                        plan._f32_chain = st
                        return st  # (the f32 chain serves this batch, but is NOT entered as the plan's chain: re-probe, as FusedAR._acquire_static does)
                else:
                    return None  # (not cached: a later, larger batch may be allowed to compile)
        _CHAINS[plan] = st
    return st or None


class ConditionerFn(torch.autograd.Function):
    """phi = net(x) for a plain (linear, activation)* network; x [N, in] contiguous fp32.  Inputs after `x`: weight_0,
    bias_0 (or None), weight_1, ... in layer order."""

    @staticmethod
    def forward(ctx, plan: SortedPlan, lins, x: Tensor, *params):
        n = len(lins)
        st = _fused_forward_state(plan, lins, x.device, x.shape[0]) if (x.shape[1] % 4 == 0 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0) else None
        chain = _dgrad_chain(plan, lins, x.shape[0]) if n >= 2 else None
        ws, wts, bs = plan.gather(lins, forward=st is None, transposes=None if chain is None else (set() if chain.full else {n - 1}))
        if chain is not None:  # (the backward reads this stream instead of the transposes; the stand-alone chain keeps the last layer's)
            wts = [chain.gather(plan, lins)] + ([] if chain.full else [wts[n - 1]])
        if st is not None:  # whole forward in one launch of the static-shape kernel
            st.refresh(lins, fine_only=True)
            acts, h = _fused_forward(st, x, plan.shapes[-1][0])
            hs = [x, *acts, h]
        else:
            hs = [x]
            h = x
            for l in range(n):
                h = plan.gemm(h, ws[l], plan.kskip_f[l], bs[l], plan.act if l + 1 < n else 0)
                hs.append(h)
        ctx.plan, ctx.n, ctx.chain = plan, n, chain
        ctx.has_bias = [b is not None for b in bs]
        ctx.save_for_backward(*hs[:-1], *wts)
        return h

    @staticmethod
    @once_differentiable  # raw-pointer HIP kernels on detached data: double backward (create_graph=True) must raise, not return graph-less grads
    def backward(ctx, g_phi: Tensor):
        plan, n = ctx.plan, ctx.n
        saved = ctx.saved_tensors
        hs, wts = saved[:n], saved[n:]
        g = g_phi.contiguous()
        grads: list = [None] * (2 * n)
        if ctx.chain is not None:
            return ConditionerFn._backward_chain(ctx, ctx.chain, g, hs, wts, grads)
        for l in range(n - 1, -1, -1):
            ConditionerFn._param_grads(ctx, l, g, hs[l], grads)  # (written where the module keeps them: no scatter back from the sorted order)
            if l > 0 or ctx.needs_input_grad[2]:
                # g_l = (g_{l+1} Ws_l) * act'(h_l): Ws_l^T plays the weight, h_l (a saved activation OUTPUT) the gate
                g = plan.gemm(g, wts[l], plan.kskip_b[l], None, 0, hs[l] if l > 0 else None, plan.act if l > 0 else 0)
        gx = g if ctx.needs_input_grad[2] else None
        return (None, None, gx, *grads)

    @staticmethod
    def _param_grads(ctx, l: int, g: Tensor, h: Tensor, grads: list) -> None:
        plan = ctx.plan
        out_f = plan.shapes[l][0]
        want_b = ctx.has_bias[l] and ctx.needs_input_grad[4 + 2 * l]
        db = None  # module order
        if ctx.needs_input_grad[3 + 2 * l]:
            grads[2 * l], db = plan.wgrad(l, g, h, want_bias=True) if want_b else (plan.wgrad(l, g, h), None)
        if want_b:
            if db is None:  # (no pass over g to ride on: column sums in the sorted order, scattered to the module's)
                db, srt = torch.empty(out_f, dtype=torch.float32, device=g.device), plan.colsum(g)
                db[plan.idx_b64[l]] = srt
            grads[2 * l + 1] = db

    @staticmethod
    def _backward_chain(ctx, chain: "DgradChain", g: Tensor, hs, wts, grads: list):
        """Last layer as the layer-wise path (its K = out_features product is a plain GEMM), every other dgrad in one launch."""
        plan, n = ctx.plan, ctx.n
        if chain.full and ConditionerFn._multi_ok(ctx):  # every dgrad in one launch, then every weight / bias gradient in two
            gs, gx = chain.run(plan, wts[0], g, hs)
            res = plan.wgrad_multi([(l, gs[l] if l + 1 < n else g, hs[l]) for l in range(n)])
            for l in range(n):
                grads[2 * l], grads[2 * l + 1] = res[l]
            return (None, None, gx if ctx.needs_input_grad[2] else None, *grads)
        ConditionerFn._param_grads(ctx, n - 1, g, hs[n - 1], grads)
        if chain.full:
            gs, gx = chain.run(plan, wts[0], g, hs)
        else:
            stream, wt_last = wts
            g_last = plan.gemm(g, wt_last, plan.kskip_b[n - 1], None, 0, hs[n - 1], plan.act)
            gs, gx = chain.run(plan, stream, g_last, hs)
        for l in range(n - 2, -1, -1):
            ConditionerFn._param_grads(ctx, l, gs[l], hs[l], grads)
        return (None, None, gx if ctx.needs_input_grad[2] else None, *grads)

    @staticmethod
    def _multi_ok(ctx) -> bool:
        """All layers want dW and db, every layer's bias gradient can ride on its wgrad pass, the split kernels are in use."""
        import os

        plan, n = ctx.plan, ctx.n
        return (n <= 4 and os.environ.get("ZUKO_AMD_EXACT_F32", "0") != "1" and os.environ.get("ZUKO_AMD_NO_WGRAD_MULTI", "0") != "1"
                and all(ctx.needs_input_grad[3 + 2 * l] and ctx.has_bias[l] and ctx.needs_input_grad[4 + 2 * l] and plan.cs_flag[l] is not None and plan.pairs[l].shape[0] > 0
                        for l in range(n)))


class AutoregressiveFn(torch.autograd.Function):
    """(y, ladj) = univariate(net(x)).call_and_ladj(x) of one unconditional masked autoregressive transform
    (zuko/flows/autoregressive.py:207-218 + zuko/transforms.py:966-992) as ONE autograd node: the forward is one launch (conditioner,
    univariate map and the feature sum of log|dy/dx|; phi and the hidden activations are kept for the backward), the backward is the
    univariate adjoint, one launch of the dgrad chain — which adds its input gradient to the adjoint's direct d/dx term — and the
    two launches of the weight / bias gradients.  uni = (kind, bound, slope, sizes): kind 0 affine / 1 spline, sizes = widths of the
    packed pieces.  Inputs after `x`: weight_0, bias_0, weight_1, ... (all trainable, see autoregressive())."""

    @staticmethod
    def forward(ctx, plan: SortedPlan, lins, st, chain, uni, x: Tensor, *params):
        # x: the conditioner's input [N, features + context] (features first), as the reference's cat(x, c) (zuko/flows/autoregressive.py:209-210)
        n = len(lins)
        st.refresh(lins, fine_only=True)
        stream = chain.gather(plan, lins)
        # maxima of the tensors the weight gradients multiply — x, h_1.., g_1.., g_phi — on the device (coupling_train.AMAX_WORDS each): the two launches leave them,
        # zk_wgrad_multi then forms its products from two-part f16 operands (three matrix instructions per block instead of six)
        am = None
        if chain.fused and os.environ.get("ZUKO_AMD_NO_WGRAD_HALF", "0") != "1":
            from . import coupling_train as ct

            am = torch.zeros((2 * n, ct.AMAX_WORDS), dtype=torch.int32, device=x.device)  # [x, h_1 .. h_{n-1} | g_1 .. g_{n-1}, g_phi]
        acts, phi, y, ladj = _fused_forward(st, x, plan.shapes[-1][0], uni=(uni[1], uni[2]), packed_width=chain.packed.width if chain.fused else 0,
                                            amax=None if am is None else [am[l] for l in range(n)])  # [x, h_1 ..]
        ctx.plan, ctx.n, ctx.chain, ctx.uni, ctx.st, ctx.am = plan, n, chain, uni, st, am
        ctx.save_for_backward(x, *acts, phi, stream)
        return y, ladj

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gl):
        from .autograd import _adj_any

        plan, n, chain = ctx.plan, ctx.n, ctx.chain
        kind, bound, slope, sizes = ctx.uni
        saved = ctx.saved_tensors
        hs, phi, stream = saved[:n], saved[n], saved[n + 1]
        x = hs[0]
        N, D = x.shape[0], ctx.st.plan.features
        gy = torch.zeros((N, D), dtype=torch.float32, device=x.device) if gy is None else gy.contiguous()
        gl = torch.zeros(N, dtype=torch.float32, device=x.device) if gl is None else gl.contiguous()
        am = ctx.am
        if chain.fused:
            gphi, gs, gx = chain.run_backward(plan, stream, ctx.st, ctx.uni, x, phi, gy, gl, hs,
                                              amax=None if am is None else {"g": [am[n + l] for l in range(n - 1)], "gphi": am[2 * n - 1]})
        else:
            xf = x if x.shape[1] == D else x[:, :D].contiguous()
            gxf, gphi = _adj_any((kind, bound, slope, sizes, ()), xf, phi.view(N, D, -1), gy, gl, True)
            gphi = gphi.view(N, -1)
            if x.shape[1] == D:
                gx = gxf
            else:  # (context columns: no direct term)
                gx = torch.zeros_like(x)
                gx[:, :D] = gxf
            gs, gx = chain.run(plan, stream, gphi, hs, gx_add=gx)
        res = plan.wgrad_multi([(l, gs[l] if l + 1 < n else gphi, hs[l]) for l in range(n)], packed_last=chain.packed if chain.fused else None,
                               amax=None if (am is None or not chain.fused) else {l: (am[n + l] if l + 1 < n else am[2 * n - 1], am[l]) for l in range(n)})
        grads = []
        for l in range(n):
            grads += list(res[l])
        return (None, None, None, None, None, gx if ctx.needs_input_grad[5] else None, *grads)


def autoregressive(module, uni, x: Tensor, features: int | None = None):
    """(y, ladj) of a masked autoregressive transform under autograd through AutoregressiveFn, or None when this conditioner / batch is
    not covered (then the caller composes ConditionerFn and the univariate map's own autograd node): a masked ReLU (linear, activation)*
    stack with an operand-split static-shape kernel and its one-launch dgrad chain, every weight and bias trainable.  x [N, features +
    context] fp32 — the conditioner's input cat(x, c), features first, its width a multiple of 4 — with 16-byte aligned rows; `features`
    defaults to all of its columns.  ZUKO_AMD_NO_FUSED_AR_TRAIN=1 switches it off."""
    import os

    if os.environ.get("ZUKO_AMD_NO_FUSED_AR_TRAIN", "0") == "1" or not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] > 0):
        return None
    features = x.shape[1] if features is None else features
    plan, lins = plan_for(module, x.device)
    if plan is None or plan.act != 1 or x.shape[1] != plan.shapes[0][1] or sum(uni[3]) * features != plan.shapes[-1][0] or features > x.shape[1]:
        return None
    n = len(lins)
    if not all(l.bias is not None and l.weight.requires_grad and l.bias.requires_grad and plan.cs_flag[i] is not None and plan.pairs[i].shape[0] > 0 for i, l in enumerate(lins)):
        return None
    if os.environ.get("ZUKO_AMD_EXACT_F32", "0") == "1" or os.environ.get("ZUKO_AMD_NO_WGRAD_MULTI", "0") == "1":
        return None
    if x.stride(1) != 1 or x.stride(0) % 4 != 0 or x.data_ptr() % 16 != 0:
        x = x.contiguous()
        if x.stride(0) % 4 != 0:
            return None
    st = _fused_forward_state(plan, lins, x.device, x.shape[0], features=features, total=sum(uni[3]))
    if st is None or not st.static[0].meta.get("split") or st.plan.layout.kind != uni[0] or st.plan.features != features:
        return None
    chain = _backward_kernel(plan, st, x.shape[0]) if n >= 2 else None  # the whole backward in one launch, else adjoint kernel + dgrad chain
    if chain is None:
        chain = _dgrad_chain(plan, lins, x.shape[0]) if n >= 2 else None
    if chain is None or not chain.full:
        return None
    params = []
    for l in lins:
        params += [l.weight, l.bias]
    return AutoregressiveFn.apply(plan, lins, st, chain, uni, x, *params)


def plan_for(module, device: torch.device):
    """SortedPlan of a `_FusedSequential` made of (linear, activation)* with a trainable activation, else None.  Cached on the
    module per device and mask version."""
    from .nn import Linear, MaskedLinear, _act_code

    mods = list(module)
    lins = mods[0::2]
    acts = mods[1::2]
    if not mods or not all(isinstance(m, (Linear, MaskedLinear)) for m in lins) or any(isinstance(m, (Linear, MaskedLinear)) for m in acts):
        return None, None
    if len(mods) != 2 * len(lins) - 1:
        return None, None
    codes = {_act_code(a) for a in acts}
    if len(codes) > 1 or (codes and (None in codes or next(iter(codes)) not in TRAIN_ACTS)):
        return None, None
    if any(l.weight.dtype != torch.float32 for l in lins):
        return None, None
    key = (str(device),) + tuple((m.mask._version, m.mask.data_ptr()) for m in lins if hasattr(m, "mask"))
    cache = _PLAN_CACHE.setdefault(module, {})  # (kept off the module: its __dict__ is pickled / deep-copied with it)
    if cache.get("key") != key:
        cache.clear()
        cache["key"] = key
        cache["plan"] = SortedPlan(lins, codes.pop() if codes else 0, device)
    return cache["plan"], lins


def conditioner(module, x: Tensor):
    """Differentiable `module(x)` through the HIP training kernels, or None when the module is not covered."""
    if not (x.is_cuda and x.dtype == torch.float32):
        return None
    plan, lins = plan_for(module, x.device)
    if plan is None:
        return None
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1 or (x2.shape[0] > 1 and x2.stride(0) % 4 != 0) or x2.data_ptr() % 16 != 0:
        x2 = x2.contiguous()
    params = []
    for l in lins:
        params += [l.weight, l.bias]
    out = ConditionerFn.apply(plan, lins, x2, *params)
    return out.reshape(lead + (out.shape[-1],))
