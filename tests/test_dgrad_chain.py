"""Tables of the generated dgrad-chain kernel (zuko_amd/static_ar.py: chain_tables; csrc/fused_ar_static_impl.h: ars_dgrad_kernel)
walked on the CPU: the tile stream + step tables must reproduce g -> (g W_l) * gate, layer after layer, for the masked conditioners
of the benchmark flows in both feature orders (what torch.autograd computes for zuko/nn.py:117-129)."""

import numpy as np
import pytest
import torch

from zuko_amd import static_ar
from zuko_amd.train import SortedPlan


def _walk(t, gathers, weights, masks, g_last, gates):
    """The kernel's loop nest in numpy.  weights / masks: module-order W_l, mask_l of the chain's layers l = 0 .. n-2."""
    n1 = t["NH"]
    vec = np.zeros((t["TMAX"] * 16,))
    vec[: g_last.shape[0]] = g_last
    outs = []
    soff = 0
    for c in range(n1):
        l = n1 - 1 - c
        idx = gathers[c]
        flat = (weights[l] * masks[l]).reshape(-1)
        vals = np.where(idx >= 0, flat[np.maximum(idx, 0)], 0.0).reshape(-1, 64, 4)
        assert vals.shape[0] % 24 == 0 and t["BASE"][c] * 256 == sum(len(g) for g in gathers[:c])
        out = np.zeros((t["TMAX"] * 16,))
        k = 0
        for s in range(soff, soff + t["NS"][c]):
            otg, it, m4 = t["S_OTG"][s], t["S_IT"][s], t["S_MASK"][s]
            for b in range(4):
                if m4 >> b & 1:
                    A = np.zeros((16, 16))
                    for lane in range(64):
                        A[lane % 16, 4 * (lane // 16) : 4 * (lane // 16) + 4] = vals[k, lane]
                    out[(otg * 4 + b) * 16 : (otg * 4 + b + 1) * 16] += A @ vec[it * 16 : (it + 1) * 16]
                    k += 1
        assert not vals[k:].any(), "padding tiles of the last chunk are zero"
        soff += t["NS"][c]
        width = t["HT"][c] * 16
        if c + 1 < n1:
            out[:width] *= gates[c]
        outs.append(out[:width].copy())
        vec = np.zeros_like(vec)
        vec[:width] = out[:width]
    return outs


@pytest.mark.parametrize("cfg", [("rqs", 64, 0, (256, 256, 256), 8), ("affine", 64, 0, (256, 256, 256), 0), ("rqs", 3, 5, (128, 128, 128), 8), ("affine", 16, 0, (128, 128), 0),
                                 ("rqs", 12, 0, (48,), 8)])
def test_chain_tables_reproduce_the_layerwise_dgrad(cfg):
    rng = np.random.default_rng(3)
    for _, _, lins in static_ar._plans_for(*cfg):
        n = len(lins)
        sp = SortedPlan(lins, 1, torch.device("cpu"))
        tg = static_ar.chain_tables(sp.mask_s_cpu[: n - 1], sp.rows_cpu[: n - 1], sp.cols_cpu[: n - 1])
        assert tg is not None
        t, gathers = tg
        assert t["NH"] == n - 1 and t["DIN"] == lins[n - 2].weight.shape[0] and t["DOUT"] == lins[0].weight.shape[1]
        W = [l.weight.detach().double().numpy() for l in lins[: n - 1]]
        M = [l.mask.detach().double().numpy() for l in lins[: n - 1]]
        g_last_sorted = rng.standard_normal(t["DIN"])
        gates = [(rng.random(t["HT"][c] * 16) > 0.4).astype(float) for c in range(n - 2)]
        outs = _walk(t, gathers, W, M, g_last_sorted, gates)
        # reference in module order: g_{l-1} = (g_l (W_l * mask_l)) * gate_{l-1}, the sorted order only permutes the units
        g = np.zeros(t["DIN"])
        g[sp.rows_cpu[n - 2]] = g_last_sorted
        for c in range(n - 1):
            l = n - 2 - c
            g = g @ (W[l] * M[l])
            if c + 1 < n - 1:
                gate_m = np.zeros(g.shape[0])
                gate_m[sp.rows_cpu[l - 1]] = gates[c][: g.shape[0]]
                g = g * gate_m
                np.testing.assert_allclose(outs[c][: g.shape[0]], g[sp.rows_cpu[l - 1]], rtol=1e-12, atol=1e-12)
            else:
                np.testing.assert_allclose(outs[c][: g.shape[0]], g, rtol=1e-12, atol=1e-12)


def test_chain_tables_decline_what_the_kernel_does_not_cover():
    (_, _, lins), _ = static_ar._plans_for("rqs", 20, 3, (100, 72), 8)   # widths that are not multiples of 16
    assert static_ar.chain_tables_for(lins) is None
    (_, _, lins), _ = static_ar._plans_for("rqs", 32, 0, (512, 512), 8)  # wider than a wavefront's 16 register tiles
    assert static_ar.chain_tables_for(lins) is None


def _bf16(x):
    return torch.from_numpy(np.asarray(x, dtype=np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def _split3(x):
    x = np.asarray(x, dtype=np.float32)
    h = _bf16(x)
    m = _bf16(x - h)
    return h, m, _bf16(x - h - m)


@pytest.mark.parametrize("cfg", [("rqs", 64, 0, (256, 256, 256), 8), ("affine", 64, 0, (256, 256, 256), 0), ("affine", 16, 0, (128, 128), 0), ("rqs", 12, 0, (48,), 8)])
def test_split_chain_tables_reproduce_the_whole_dgrad(cfg):
    """static_ar.chain_split_tables (the operand-split chain over ALL layers, csrc/fused_ar_split_impl.h: arxd_kernel) walked on the CPU:
    blocks of (out tile, in pair) as three bf16 images, six partial products per block, layer 0 in-pair major with its list of live
    pairs — must give d loss / d (hidden pre-activations) and d loss / d x of the masked ReLU network to f32 accuracy."""
    _walk_split_chain(cfg, packed=False)


@pytest.mark.parametrize("cfg", [("rqs", 64, 0, (256, 256, 256), 8), ("affine", 64, 0, (256, 256, 256), 0), ("affine", 16, 0, (128, 128), 0), ("rqs", 8, 0, (48,), 8)])
def test_packed_chain_tables_reproduce_the_whole_dgrad(cfg):
    """The tables of the one-launch backward (chain_split_tables(packed=...), arxb_kernel): the first chain layer contracts over the FORWARD
    kernel's packed order of phi (zuko_amd/fused.py: build_plan — unit 16 (g NT + t) + i of the stream = parameter 4 t + i % 4 of the features
    of lane i / 4 in group g), in which a lane owns its features' parameters; PB lists the first block of every pair of packed tiles."""
    _walk_split_chain(cfg, packed=True)


def _walk_split_chain(cfg, packed):
    rng = np.random.default_rng(5)
    for plan, layout, lins in static_ar._plans_for(*cfg):
        n = len(lins)
        sp = SortedPlan(lins, 1, torch.device("cpu"))
        pk = {"uni": layout.kind, "featmap": plan.featmap, "nt": layout.nt, "fpl": layout.fpl, "total": layout.total} if packed else None
        tg = static_ar.chain_split_tables(sp.mask_s_cpu, sp.rows_cpu, sp.cols_cpu, packed=pk)
        assert tg is not None
        t, gathers = tg
        dphi = lins[-1].weight.shape[0]
        if packed:
            # packed unit -> module row, as the forward plan lays the last layer's rows out (fused.py: build_plan)
            ng = len(plan.featmap) // (4 * layout.fpl)
            mod_row = -np.ones(ng * layout.nt * 16, dtype=np.int64)
            for g in range(ng):
                for tt in range(layout.nt):
                    for i in range(16):
                        fi, k = divmod(4 * tt + (i & 3), layout.total)
                        if fi < layout.fpl:
                            f = plan.featmap[g * 4 * layout.fpl + (i >> 2) * layout.fpl + fi]
                            if f >= 0:
                                mod_row[(g * layout.nt + tt) * 16 + i] = f * layout.total + k
            assert t["chain"] == 3 and t["MODROW"] == mod_row.tolist() and t["NG"] == ng and t["DIN0"] == len(mod_row) and sorted(mod_row[mod_row >= 0]) == list(range(dphi))
            ips_ = np.asarray(t["B_IP"][: t["NB"][0]])
            assert t["PB"] == [int((ips_ < pp).sum()) for pp in range(t["DIN0"] // 32 + 1)] and t["PB"][-1] == t["NB"][0]
        else:
            assert t["DIN0"] == dphi
        assert t["NH"] == n and t["DOUT"] == lins[0].weight.shape[1]
        assert t["BASE"] == [3 * sum(t["NB"][:c]) for c in range(n)] and t["NCHUNK"] == -(-3 * sum(t["NB"]) // t["CH"])
        assert sorted(set(t["B_IP"][: t["NB"][0]])) == t["P0"] and len(t["P0"]) == t["NP0"]
        ips = t["B_IP"][: t["NB"][0]]
        assert all(ips[i] <= ips[i + 1] for i in range(len(ips) - 1)), "layer 0 is in-pair major"
        W = [(l.weight.detach().numpy() * l.mask.numpy()).astype(np.float32) for l in lins]
        g_phi = rng.standard_normal(dphi).astype(np.float32)
        gates = [(rng.random(lins[l].weight.shape[0]) > 0.4).astype(np.float64) for l in range(n - 1)]  # module order, per hidden layer
        # reference (module order)
        ref, g = [], g_phi.astype(np.float64)
        for l in range(n - 1, -1, -1):
            g = g @ W[l].astype(np.float64)
            if l > 0:
                g = g * gates[l - 1]
            ref.append(g)
        # the kernel's walk (sorted unit order)
        vec = np.zeros(max(t["DIN0"], t["TMAX"] * 16) + 32, dtype=np.float32)
        vec[: t["DIN0"]] = np.where(mod_row >= 0, g_phi[np.maximum(mod_row, 0)], 0.0) if packed else g_phi  # (unpacked: the last layer's rows are in module order)
        boff = 0
        for c in range(n):
            l = n - 1 - c
            idx = gathers[c].reshape(-1, 64, 8)
            vals = np.where(idx >= 0, W[l].reshape(-1)[np.maximum(idx, 0)], 0.0).astype(np.float32)
            ah, am, al = _split3(vals)
            vh, vm, vl = _split3(vec)
            out = np.zeros(t["HT"][c] * 16)
            for s in range(t["NB"][c]):
                ot, ip = t["B_OT"][boff + s], t["B_IP"][boff + s]
                for lane in range(64):
                    i, kq = lane % 16, lane // 16
                    units = np.concatenate([np.arange(4) + (2 * ip) * 16 + 4 * kq, np.arange(4) + (2 * ip + 1) * 16 + 4 * kq])
                    for a_, b_ in ((al, vh), (ah, vl), (am, vm), (am, vh), (ah, vm), (ah, vh)):
                        out[ot * 16 + i] += float(np.dot(a_[s, lane].astype(np.float64), b_[units].astype(np.float64)))
            boff += t["NB"][c]
            width = W[l].shape[1]
            if l > 0:
                out[:width] *= gates[l - 1][sp.rows_cpu[l - 1]]
                want = ref[c][sp.rows_cpu[l - 1]]
            else:
                want = ref[c]
            scale = max(1.0, np.abs(want).max())
            assert np.abs(out[:width] - want).max() <= 4e-6 * scale, (c, np.abs(out[:width] - want).max(), scale)
            vec = np.zeros_like(vec)
            vec[:width] = out[:width].astype(np.float32)


@pytest.mark.parametrize("cfg", [("rqs", 64, 0, (256, 256, 256), 8), ("affine", 12, 0, (64, 64), 0), ("rqs", 3, 5, (128, 128, 128), 8)])
def test_packed_rows_tables_give_the_module_order_weight_gradient(cfg):
    """zuko_amd/train.py:PackedRows — the row table, sorted-domain mask, live 128 x 128 blocks and column-sum flags the weight-gradient kernels use when
    d loss / d phi arrives in the fused kernels' packed order (padding slots included): walked in numpy, they must give dW and db of the LAST linear
    layer in the module's order (what autograd derives from zuko/nn.py:217-218), with every padding slot skipped and every live weight inside a listed block."""
    from zuko_amd.train import PackedRows

    rng = np.random.default_rng(11)
    for plan, layout, lins in static_ar._plans_for(*cfg):
        sp = SortedPlan(lins, 1, torch.device("cpu"))
        tg = static_ar.chain_split_tables(sp.mask_s_cpu, sp.rows_cpu, sp.cols_cpu, packed={"uni": layout.kind, "featmap": plan.featmap, "nt": layout.nt, "fpl": layout.fpl, "total": layout.total})
        assert tg is not None
        pk = PackedRows(sp, tg[0]["MODROW"], torch.device("cpu"))
        last = len(lins) - 1
        out_f, in_f = sp.shapes[last]
        assert pk.ok and pk.width == len(tg[0]["MODROW"]) and pk.width % 16 == 0
        rows, cols = pk.rows.numpy(), sp.cols_cpu[last]
        assert sorted(rows[rows >= 0].tolist()) == list(range(out_f)), "every row of the layer has exactly one packed slot"
        N = 50
        g_mod = rng.standard_normal((N, out_f))
        h_mod = rng.standard_normal((N, in_f))        # the layer's input in the MODULE's unit order
        mask = lins[last].mask.numpy().astype(bool)
        dw_ref, db_ref = mask * (g_mod.T @ h_mod), g_mod.sum(axis=0)
        g_packed = np.where(rows >= 0, g_mod[:, np.maximum(rows, 0)], 0.0)  # what the backward launch writes: padding slots are zero
        h_sorted = h_mod[:, cols]                                             # what the forward stores: units in the plan's sorted order
        ms = pk.mask_s.numpy().astype(bool)
        dw, db = np.zeros((out_f, in_f)), np.zeros(out_f)
        live = np.zeros_like(ms)
        for ob, ib in pk.pairs.numpy():
            live[ob * 128 : (ob + 1) * 128, ib * 128 : (ib + 1) * 128] = True
        assert not (ms & ~live).any(), "a non-zero weight outside the listed blocks"
        full = g_packed.T @ h_sorted  # [width, in] in the sorted domain
        for u in range(pk.width):
            if rows[u] < 0:
                assert not ms[u].any()
                continue
            dw[rows[u], cols] = np.where(ms[u] & live[u], full[u], 0.0)
            db[rows[u]] = g_packed[:, u].sum()
        assert np.abs(dw - dw_ref).max() < 1e-9 and np.abs(db - db_ref).max() < 1e-9
        flagged = {int(ob) for (ob, _), f in zip(pk.pairs.numpy(), pk.cs_flag.numpy()) if f}
        assert flagged == set(range(-(-pk.width // 128))), "one column-sum block per 128 packed columns (the bias gradient rides on the weight pass)"
