"""CPU: the host-side plan of the fused autoregressive kernel (unit permutation, tile skipping,
weight-stream order, chunk padding, last-layer regrouping) reproduces the conditioner exactly when
walked by a numpy model of the kernel's data flow."""

import numpy as np
import pytest
import torch

from oracle import zuko_oracle as O
from plan_emulators import simulate_ar, simulate_coupling, simulate_inc


@pytest.mark.parametrize(
    "name,kind,total,bins,D,C,kw",
    [
        ("nsf64", "rqs", 23, 8, 64, 0, dict(hidden_features=[256] * 3)),
        ("nsf3c5", "rqs", 23, 8, 3, 5, dict(hidden_features=[128] * 3)),
        ("maf64", "affine", 2, 0, 64, 0, dict(hidden_features=[256] * 3)),
        ("maf10", "affine", 2, 0, 10, 3, dict(hidden_features=[40, 72])),
        ("nsf_p2", "rqs", 23, 8, 12, 0, dict(hidden_features=[64, 64], passes=2)),
        ("nsf_k4", "rqs", 11, 4, 8, 2, dict(hidden_features=[48, 48], bins=4)),
        ("nsf_k16", "rqs", 47, 16, 8, 0, dict(hidden_features=[64], bins=16)),
    ],
)
def test_plan_simulation_matches_masked_mlp(name, kind, total, bins, D, C, kw):
    from zuko_amd import fused
    from zuko_amd.flows import MAF, NSF

    torch.manual_seed(0)
    flow = (NSF if kind == "rqs" else MAF)(D, C, transforms=2, **kw)
    for t in flow.transform.transforms:
        lins = [m for m in t.hyper if hasattr(m, "mask")]
        masks = [m.mask for m in lins]
        plan = fused.build_plan(masks, D, fused.uni_layout(kind, total, bins))
        assert plan is not None and plan.n_blocks % fused.CHUNK == 0
        W = [m.weight.detach().double().numpy() for m in lins]
        B = [m.bias.detach().double().numpy() for m in lins]
        x = torch.randn(19, D + C, dtype=torch.float64)
        ref = O.mlp_forward(x, [torch.tensor(w) for w in W], [torch.tensor(b) for b in B], masks).reshape(19, D, total).numpy()
        phi = simulate_ar(plan, W, B, [m.numpy() for m in masks], x.numpy(), lambda v: np.maximum(v, 0))
        assert np.abs(phi - ref).max() < 1e-12
    if name == "nsf64":
        assert plan.kept_tiles < 0.6 * plan.dense_tiles  # degree sort exposes the block-triangular structure


def test_unsupported_shapes_fall_back():
    from zuko_amd import fused
    from zuko_amd.flows import NSF

    assert fused.uni_layout("rqs", 3 * 32 - 1, 32) is None
    assert fused.uni_layout("rqs", 47, 16).nt == 12 and fused.uni_layout("rqs", 11, 4).nt == 3
    assert not fused.layout_supports(fused.uni_layout("rqs", 11, 4), 6)  # 4-bin layout needs float4 rows
    t = NSF(8, 0, transforms=1, hidden_features=[512]).transform.transforms[0]
    assert fused.build_plan([m.mask for m in t.hyper if hasattr(m, "mask")], 8, fused.uni_layout("rqs", 23, 8)) is None
    assert t._fusable_layout() is not None
    t = NSF(8, 0, transforms=1, bins=5).transform.transforms[0]
    assert t._fusable_layout() is None


def test_plan_simulation_on_random_adjacencies():
    """Free-form (DAG) adjacencies, random widths, context columns: whatever masks zuko's construction yields, the
    plan's tile skipping and regrouping must leave the conditioner's output unchanged (property-style, seeded)."""
    from zuko_amd import fused
    from zuko_amd.flows import MaskedAutoregressiveTransform

    rng = np.random.default_rng(7)
    done = 0
    for trial in range(40):
        D = int(rng.integers(2, 33))
        C = int(rng.integers(0, 4))
        perm = rng.permutation(D)
        adj = np.tril(rng.random((D, D)) < rng.uniform(0.1, 0.7), k=-1)
        adj = adj[perm][:, perm] | np.eye(D, dtype=bool)  # a DAG under a random feature order, self loops as zuko expects
        if C:
            adj = np.concatenate((adj, rng.random((D, C)) < 0.6), axis=1)
        hidden = [int(rng.choice([16, 40, 64, 100, 256])) for _ in range(int(rng.integers(1, 4)))]
        torch.manual_seed(trial)
        try:
            t = MaskedAutoregressiveTransform(D, C, adjacency=torch.from_numpy(adj), hidden_features=hidden)
        except ValueError:  # zuko refuses adjacencies that lead to a null Jacobian (nn.py:285-286)
            continue
        lins = [m for m in t.hyper if hasattr(m, "mask")]
        masks = [m.mask for m in lins]
        plan = fused.build_plan(masks, D, fused.uni_layout("affine", 2))
        assert plan is not None
        W = [m.weight.detach().double().numpy() for m in lins]
        B = [m.bias.detach().double().numpy() for m in lins]
        x = torch.randn(5, D + C, dtype=torch.float64)
        ref = O.mlp_forward(x, [torch.tensor(w) for w in W], [torch.tensor(b) for b in B], masks).reshape(5, D, 2).numpy()
        phi = simulate_ar(plan, W, B, [m.numpy() for m in masks], x.numpy(), lambda v: np.maximum(v, 0))
        assert np.abs(phi - ref).max() < 1e-12, (trial, D, C, hidden)
        done += 1
    assert done >= 25


@pytest.mark.parametrize("kind,D,hidden,bins", [("rqs", 64, [256] * 3, 8), ("affine", 64, [256] * 3, 0), ("rqs", 32, [128], 16), ("affine", 5, [24, 32], 0),
                                                 ("rqs", 40, [160, 160], 4), ("affine", 64, [256, 256], 0), ("sos", 64, [256] * 3, 0), ("bern", 64, [256] * 3, 0)])
def test_incremental_inverse_plan_simulation_matches_oracle(kind, D, hidden, bins):
    """The aligned-tile plan of the incremental inverse kernel, walked in numpy over the very tables / stream the kernel
    consumes, reproduces the oracle's `passes`-sweep inverse and the forward log-determinant at the solution (float64)."""
    import numpy as np

    import zuko_amd.flows as F
    from oracle import zuko_oracle as O
    from zuko_amd import fused, incremental as inc

    torch.manual_seed(D + len(hidden))
    ctx = 2 if D == 5 else 0
    if kind in ("sos", "bern"):  # the polynomial flows (round 6: bisection inverse in the incremental launch): 16 / 17 parameters per feature, 4 / 5 tiles per group
        from zuko_amd.flows.autoregressive import MaskedAutoregressiveTransform

        flow = (F.SOSPF if kind == "sos" else F.BPF)(D, ctx, transforms=2, hidden_features=hidden)
        lazy = [t for t in flow.transform.transforms if isinstance(t, MaskedAutoregressiveTransform)][1]
        lay = fused.UniLayout(5, 16, 1, 4) if kind == "sos" else fused.UniLayout(6, 17, 1, 5)
    else:
        flow = F.MAF(D, ctx, transforms=2, hidden_features=hidden) if kind == "affine" else F.NSF(D, ctx, transforms=2, bins=bins, hidden_features=hidden)
        lazy = flow.transform.transforms[1]  # descending order
        lay = fused.UniLayout(0, 2, 1, 1) if kind == "affine" else fused.UniLayout({8: 1, 4: 2, 16: 3}[bins], 3 * bins - 1, 1, (3 * bins - 1 + 3) // 4, bins)
    lins = [m for m in lazy.hyper if hasattr(m, "mask")]
    plan = inc.build_inc_plan([l.mask for l in lins], D, lazy.order.numpy(), lay)
    assert plan is not None and plan.n_groups <= inc.MAX_TILES and plan.n_blocks % inc.CHUNK == 0
    # stream length = the static layout the kernel assumes
    NH, NT = plan.n_hidden, plan.nt
    assert plan.n_blocks == -(-sum(inc.L1S + (NH - 1) * j + NT * j + inc.L1D + (NH - 1) + NT for j in range(plan.n_groups)) // inc.CHUNK) * inc.CHUNK
    W = [l.weight.detach().double().numpy() for l in lins]
    B = [l.bias.detach().double().numpy() for l in lins]
    Mk = [l.mask.numpy() for l in lins]
    g = torch.Generator().manual_seed(1)
    y = torch.randn(30, D, generator=g, dtype=torch.float64)
    c = torch.randn(30, ctx, generator=g, dtype=torch.float64) if ctx else None
    uni = O.UNI_AFFINE if kind == "affine" else (O.uni_sos() if kind == "sos" else (O.uni_bpf() if kind == "bern" else O.uni_rqs(bins)))
    layer = O.ARLayer(uni, [torch.from_numpy(w) for w in W], [torch.from_numpy(b) for b in B], [torch.from_numpy(m) for m in Mk], lazy.passes, D)
    xo = O.ar_inverse(layer, y, c)
    _, lo = O.ar_forward(layer, xo, c)

    def inv_fn(phi, yv):
        ph, yy = torch.from_numpy(phi)[:, None, :], torch.from_numpy(yv)[:, None]
        x = O.univariate_inverse(uni, ph, yy)
        _, l = O.univariate_forward(uni, ph, x)
        return x[:, 0].numpy(), l[:, 0].numpy()

    xs, ls = simulate_inc(plan, W, B, Mk, y.numpy(), None if c is None else c.numpy(), lambda v: np.maximum(v, 0), inv_fn)
    if kind in ("sos", "bern"):  # (a bisection: both walks stop at the same 2^-24 bracket unless a comparison sits within rounding of the target)
        assert np.abs(xs - xo.numpy()).max() < 1e-5 and np.abs(ls - lo.numpy()).max() < 1e-3
        return
    assert np.abs(xs - xo.numpy()).max() < 1e-12 and np.abs(ls - lo.numpy()).max() < 1e-11
    # the HALF stream (round 6: pulls as 16 x 32 blocks of two f16 images on the f16 matrix instruction, per-pair power-of-two scales): same groups,
    # same diagonal tiles; its static length, and the walk with the kernel's operand split agrees with the oracle to the split's 2^-22
    hs = inc.half_stream(plan, [l.mask for l in lins])
    images = sum(inc.L1S + 2 * ((NH - 1) + NT) * ((j + 1) // 2) + inc.L1D + (NH - 1) + NT for j in range(plan.n_groups))
    assert hs.n_images == -(-images // inc.CHUNK) * inc.CHUNK == hs.n_chunks * inc.CHUNK
    assert sum(len(p) for p in hs.blk_pos[1:]) == sum(((NH - 1) + NT) * ((j + 1) // 2) for j in range(plan.n_groups))
    taken = np.concatenate([np.concatenate([p, p + 1]) for p in hs.blk_pos[1:] if len(p)]) if any(len(p) for p in hs.blk_pos[1:]) else np.zeros(0, int)
    assert len(set(taken.tolist())) == len(taken) and (hs.gather_f32.reshape(-1, 256)[taken] == -1).all(), "block images are disjoint and left to the block gathers"
    from zuko_amd import fused as Fu

    wexp = [0] + [e for _, e in Fu.half_scales(lins)][1:]
    xh, lh = simulate_inc(plan, W, B, Mk, y.numpy(), None if c is None else c.numpy(), lambda v: np.maximum(v, 0), inv_fn, half=hs, wexp=wexp)
    assert np.abs(xh - xo.numpy()).max() < 1e-6 * max(1.0, np.abs(xo.numpy()).max()) and np.abs(lh - lo.numpy()).max() < 1e-5, (np.abs(xh - xo.numpy()).max(), np.abs(lh - lo.numpy()).max())


def test_incremental_plan_rejects_layouts_that_do_not_fit():
    import zuko_amd.flows as F
    from zuko_amd import fused, incremental as inc

    torch.manual_seed(0)
    lazy = F.NSF(3, 5, transforms=1, hidden_features=[128] * 3).transform.transforms[0]  # 64 units per degree: no 16-unit aligned tiles
    lins = [m for m in lazy.hyper if hasattr(m, "mask")]
    assert inc.build_inc_plan([l.mask for l in lins], 3, lazy.order.numpy(), fused.UniLayout(1, 23, 1, 6, 8)) is None


@pytest.mark.parametrize("D,ctx,hidden", [(256, 0, [512] * 3), (5, 3, [32, 32]), (12, 2, [40, 70, 24])])
def test_coupling_plan_simulation_matches_oracle(D, ctx, hidden):
    """The fused coupling kernel's stream / index maps, walked in numpy, reproduce the oracle's coupling layer (float64)."""
    import numpy as np

    import zuko_amd.flows as F
    from oracle import zuko_oracle as O
    from zuko_amd import coupling_plan as cp

    torch.manual_seed(D)
    t = F.RealNVP(D, ctx, transforms=2, hidden_features=hidden).transform.transforms[1]
    lins = list(t.hyper)[0::2]
    idx_a, idx_b = t.mask.nonzero().squeeze(-1).numpy(), (~t.mask).nonzero().squeeze(-1).numpy()
    plan = cp.build_coupling_plan([tuple(l.weight.shape) for l in lins], idx_a, idx_b, D, ctx)
    assert plan is not None and plan.n_blocks % cp.CHUNK == 0
    W = [l.weight.detach().double().numpy() for l in lins]
    B = [l.bias.detach().double().numpy() for l in lins]
    g = torch.Generator().manual_seed(2)
    n = 8 if D > 64 else 40
    x = torch.randn(n, D, generator=g, dtype=torch.float64)
    c = torch.randn(n, ctx, generator=g, dtype=torch.float64) if ctx else None
    layer = O.CouplingLayer(O.UNI_AFFINE, [torch.from_numpy(w) for w in W], [torch.from_numpy(b) for b in B], t.mask)
    yo, lo = O.coupling_forward(layer, x, c)
    ys, ls = simulate_coupling(plan, W, B, x.numpy(), None if c is None else c.numpy(), lambda v: np.maximum(v, 0), float(np.log(1e-3)))
    assert np.abs(ys - yo.numpy()).max() < 1e-12 and np.abs(ls - lo.numpy()).max() < 1e-12


def _walk_static_tables(t, stream, bias_img, inp, rev_l0=None):
    """Pure-numpy walk of the weight stream exactly as csrc/fused_ar_static_impl.h does it from the GENERATED tables (step lists,
    stream positions from the popcounts of the tile masks, last-layer groups): returns the accumulators of every feature group
    [n, NG, NT, 16].  A wrong table shows up as a difference from simulate_ar(), which walks the plan itself."""
    n = inp.shape[0]
    tiles = stream.reshape(-1, 64, 4)  # [tile][lane][r]
    A = lambda b: tiles[b].reshape(4, 16, 4).transpose(1, 0, 2).reshape(16, 16)  # lane (i, q), r -> A[i][4 q + r]
    T = t["TMAX"]
    cur = np.zeros((n, T * 16))
    cur[:, : inp.shape[1]] = inp
    soff = np.concatenate([[0], np.cumsum(t["NS"])])
    for l in range(t["NH"]):
        out = np.zeros((n, T * 16))
        out[:, : t["BIAS_STRIDE"]] += bias_img[l * t["BIAS_STRIDE"] : (l + 1) * t["BIAS_STRIDE"]][None, : T * 16] if T * 16 <= t["BIAS_STRIDE"] else 0
        pos = t["BASE"][l]
        for s_ in range(t["NS"][l]):
            otg, it, m = t["S_OTG"][soff[l] + s_], t["S_IT"][soff[l] + s_], t["S_MASK"][soff[l] + s_]
            if l == 0 and rev_l0 is not None:
                it = rev_l0[s_]
            for tt in range(4):
                if m >> tt & 1:
                    out[:, (otg * 4 + tt) * 16 : (otg * 4 + tt + 1) * 16] += cur[:, it * 16 : it * 16 + 16] @ A(pos).T
                    pos += 1
        cur = np.maximum(out, 0)
    nt = {0: 1, 1: 6, 2: 3, 4: 6}[t["uni"]]
    acc = np.zeros((n, t["NG"], nt, 16))
    bl = bias_img[t["NH"] * t["BIAS_STRIDE"] :].reshape(t["NG"], nt, 16)
    pos = t["LAST_BASE"]
    for g in range(t["NG"]):
        acc[:, g] += bl[g][None]
        for it in t["G_IT"][t["GOFF"][g] : t["GOFF"][g + 1]]:
            for tt in range(nt):
                acc[:, g, tt] += cur[:, it * 16 : it * 16 + 16] @ A(pos).T
                pos += 1
    return acc


@pytest.mark.parametrize("cfg", [("rqs", 64, 0, (256, 256, 256), 8), ("affine", 64, 0, (256, 256, 256), 0), ("rqs", 3, 5, (128, 128, 128), 8), ("rqs", 32, 0, (512, 512), 8),
                                 ("rqs", 20, 3, (100, 72), 8), ("affine", 7, 0, (40,), 0)])
def test_static_kernel_tables_describe_the_plan(cfg):
    """zuko_amd/static_ar.py:tables() — what a generated static-shape kernel is compiled from — against the plan it came from, for
    both feature orders zuko alternates between (zuko/flows/autoregressive.py:121-125): the table-driven walk of the per-tile stream
    reproduces phi of simulate_ar() (which walks the plan) and of the masked network itself, including widths that are not
    multiples of 16 / 64, a context, a single hidden layer and the 512-wide plan only the static kernels cover."""
    from zuko_amd import fused, static_ar

    kind, D, C, hidden, bins = cfg
    plans = [pl[:2] for pl in static_ar._plans_for(kind, D, C, hidden, bins)]
    tabs = [static_ar.tables(p, lay.kind) for p, lay in plans]
    assert all(t is not None for t in tabs)
    (ca, la), (cd, ld) = static_ar._split(tabs[0]), static_ar._split(tabs[1])
    if cfg[:4] in (("rqs", 64, 0, (256, 256, 256)), ("affine", 64, 0, (256, 256, 256))):
        assert ca == cd and la != ld, "cfg2 / cfg3: one kernel serves both orders through its alternative first-layer pattern"
        assert tabs[0]["NCHUNK"] == (48 if kind == "rqs" else 17) and tabs[0]["NS"] == [10, 40, 40]
    rng = np.random.default_rng(5)
    for (plan, lay), t in zip(plans, tabs):
        assert t["WAVES"] == (4 if max(hidden) > 256 else 8) and t["NCHUNK"] == plan.fine_n_chunks
        masks = None
    # numerical walk: random weights on the plan's masks
    import torch
    from zuko_amd.flows.autoregressive import MaskedAutoregressiveTransform
    from zuko_amd.nn import MaskedLinear
    from zuko_amd.transforms import MonotonicAffineTransform, MonotonicRQSTransform

    for i, order in enumerate((torch.arange(D), torch.flipud(torch.arange(D)))):
        if kind == "affine":
            tr = MaskedAutoregressiveTransform(D, C, order=order, hidden_features=list(hidden), univariate=MonotonicAffineTransform, shapes=[(), ()])
        else:
            tr = MaskedAutoregressiveTransform(D, C, order=order, hidden_features=list(hidden), univariate=MonotonicRQSTransform, shapes=[(bins,), (bins,), (bins - 1,)])
        lins = [m for m in tr.hyper if isinstance(m, MaskedLinear)]
        M = [m.mask.numpy().astype(bool) for m in lins]
        W = [rng.standard_normal(m.shape) for m in M]
        B = [rng.standard_normal(m.shape[0]) for m in M]
        plan, lay = plans[i]
        t = tabs[i]
        x = rng.standard_normal((5, D + C))
        fine = np.concatenate([np.where(g >= 0, (W[l] * M[l]).reshape(-1)[np.maximum(g, 0)], 0.0) for l, g in enumerate(plan.fine_gather)])
        bias_img = np.concatenate([np.where(g >= 0, B[l][np.maximum(g, 0)], 0.0) for l, g in enumerate(plan.bias_gather)])
        acc = _walk_static_tables(t, fine, bias_img, x)
        h = x
        for l in range(len(M)):
            h = h @ (W[l] * M[l]).T + B[l]
            if l + 1 < len(M):
                h = np.maximum(h, 0)
        phi = h.reshape(5, D, lay.total)
        per_group = 4 * lay.fpl
        for g in range(plan.n_groups):
            for tt in range(lay.nt):
                for r in range(16):
                    m = 4 * tt + (r & 3)
                    fi, p_ = divmod(m, lay.total)
                    if fi < lay.fpl:
                        f = plan.featmap[g * per_group + (r >> 2) * lay.fpl + fi]
                        if f >= 0:
                            assert np.allclose(acc[:, g, tt, r], phi[:, f, p_], rtol=1e-9, atol=1e-9), (cfg, i, g, tt, r)
    # the emitted source carries exactly these tables
    src = static_ar.emit(tabs[0], None if la == ld or ca != cd else ld)
    assert f"NCHUNK = {tabs[0]['NCHUNK']}" in src and "zk_ars_launch" in src and ("HAS_ALT = true" in src) == (ca == cd and la != ld)


def test_training_plan_sorts_units_by_their_true_dependency_count():
    """zuko_amd/train.py:SortedPlan must sort every hidden layer by the dependency count computed in the MODULE's unit order
    (a regression here keeps the reparametrisation exact but leaves the sorted masks dense: no k-tile or block is skipped)."""
    import zuko_amd.flows as F
    from zuko_amd import fused, train
    from zuko_amd.nn import MaskedLinear

    torch.manual_seed(4)
    flow = F.NSF(64, 0, transforms=2, bins=8, hidden_features=[256] * 3)
    for lazy in flow.transform.transforms:
        lins = [m for m in lazy.hyper if isinstance(m, MaskedLinear)]
        plan = train.SortedPlan(lins, 1, torch.device("cpu"))
        assert plan.kept[1] == 0.75 and plan.kept[2] == 0.75 and plan.kept[3] < 0.8  # block lower-triangular at 128 x 128
        live = [[bin(int(np.uint64(v))).count("1") for v in k.numpy().view(np.uint64)] for k in plan.kskip_f]
        assert live[1] == [4, 8] and live[2] == [4, 8] and live[3] in (sorted(live[3]), sorted(live[3], reverse=True)) and min(live[3]) == 1
        # the same permutations as the fused kernels' plan (stable sort on the same keys): their activations are interchangeable
        deps = fused._deps([m.mask.numpy().astype(bool) for m in lins])
        for l in range(3):
            assert np.array_equal(plan.perms[l].numpy(), np.argsort(deps[l].sum(axis=1), kind="stable"))


def _bf16_round(x: np.ndarray) -> np.ndarray:
    return torch.from_numpy(np.asarray(x, dtype=np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def _split3(x: np.ndarray):
    x = np.asarray(x, dtype=np.float32)
    h = _bf16_round(x)
    m = _bf16_round(x - h)
    l = _bf16_round(x - h - m)
    return h, m, l


@pytest.mark.parametrize("cfg", [("rqs", 64, 0, (256, 256, 256), 8), ("affine", 64, 0, (256, 256, 256), 0), ("rqs", 3, 5, (128, 128, 128), 8), ("rqs", 20, 3, (100, 72), 8),
                                 ("affine", 7, 2, (40,), 0)])
def test_split_kernel_tables_describe_the_plan(cfg):
    """Tables + gathers of the operand-split kernels (static_ar.split_tables; csrc/fused_ar_split_impl.h) walked on the CPU exactly as
    the kernel walks them — blocks of (out tile, in pair) as three bf16 images, six partial products per block — must give the masked
    MLP's output to f32 accuracy, both feature orders, including widths that are not multiples of 16 / 32 and a context."""
    from zuko_amd import static_ar

    kind, D, C, hidden, bins = cfg
    rng = np.random.default_rng(11)
    for plan, lay, lins in static_ar._plans_for(kind, D, C, hidden, bins):
        ts = static_ar.split_tables(plan, lay.kind, 1)
        assert ts is not None
        t, gathers = ts
        NH = t["NH"]
        images = 3 * (sum(t["NB"]) + t["GOFF"][-1] * lay.nt)
        assert t["NCHUNK"] == -(-images // t["CH"]), "no chunk of nothing but padding: the ring moves on when a chunk's first image is read"
        assert t["NCHUNK"] * t["CH"] <= 3 * sum(len(g) // 512 for g in gathers) <= t["STREAM_IMAGES"] < t["NCHUNK"] * t["CH"] + 3
        assert t["BASE"] == [3 * sum(t["NB"][:l]) for l in range(NH)] and t["LAST_BASE"] == 3 * sum(t["NB"]), "layers follow each other without padding"
        assert (t["WAVES"], t["CH"]) == static_ar.split_geometry()
        W = [l.weight.detach().numpy().astype(np.float32) * l.mask.numpy() for l in lins]
        Bv = [l.bias.detach().numpy().astype(np.float64) for l in lins]
        x = rng.standard_normal(plan.din).astype(np.float32)
        # reference: the masked MLP in float64 (module order)
        h = x.astype(np.float64)
        for l in range(NH + 1):
            h = W[l].astype(np.float64) @ h + Bv[l]
            if l < NH:
                h = np.maximum(h, 0.0)
        ref = h
        # the kernel's walk: activations as tiles of 16 in the plan's sorted unit order
        def images(l):
            idx = gathers[l].reshape(-1, 64, 8)
            flat = W[l].reshape(-1)
            vals = np.where(idx >= 0, flat[np.maximum(idx, 0)], 0.0).astype(np.float32)
            return _split3(vals)
        vec = np.zeros(t["TMAX"] * 16, dtype=np.float32)
        vec[: plan.din] = x
        boff = 0
        for l in range(NH):
            ah, am, al = images(l)
            bias_img = np.where(plan.bias_gather[l] >= 0, Bv[l][np.maximum(plan.bias_gather[l], 0)], 0.0)
            out = bias_img[: t["TMAX"] * 16].astype(np.float64).copy()
            vh, vm, vl = _split3(vec)
            for s in range(t["NB"][l]):
                ot, ip = t["B_OT"][boff + s], t["B_IP"][boff + s]
                for lane in range(64):
                    i, kq = lane % 16, lane // 16
                    units = np.concatenate([np.arange(4) + (2 * ip) * 16 + 4 * kq, np.arange(4) + (2 * ip + 1) * 16 + 4 * kq])
                    acc = 0.0
                    for a_, b_ in ((al, vh), (ah, vl), (am, vm), (am, vh), (ah, vm), (ah, vh)):
                        acc += float(np.dot(a_[s, lane].astype(np.float64), b_[units].astype(np.float64)))
                    out[ot * 16 + i] += acc
            assert not ah[t["NB"][l] :].any()
            boff += t["NB"][l]
            vec = np.zeros_like(vec)
            vec[: t["HT"][l] * 16] = np.maximum(out[: t["HT"][l] * 16], 0.0).astype(np.float32)
        # last layer: group g, tile b, tile row i <-> feature slot / parameter (fused.build_plan)
        ah, am, al = images(NH)
        vh, vm, vl = _split3(vec)
        nt, total, fpl = lay.nt, lay.total, lay.fpl
        got = np.full(D * total, np.nan)
        blk = 0
        bias_last = plan.bias_gather[NH].reshape(-1, nt, 16)
        for g in range(t["NG"]):
            acc = np.zeros((nt, 16))
            for b in range(nt):
                acc[b] = np.where(bias_last[g, b] >= 0, Bv[NH][np.maximum(bias_last[g, b], 0)], 0.0)
            for st in range(t["GOFF"][g], t["GOFF"][g + 1]):
                ip = t["G_IP"][st]
                for b in range(nt):
                    for lane in range(64):
                        i, kq = lane % 16, lane // 16
                        units = np.concatenate([np.arange(4) + (2 * ip) * 16 + 4 * kq, np.arange(4) + (2 * ip + 1) * 16 + 4 * kq])
                        for a_, b_ in ((al, vh), (ah, vl), (am, vm), (am, vh), (ah, vm), (ah, vh)):
                            acc[b, i] += float(np.dot(a_[blk, lane].astype(np.float64), b_[units].astype(np.float64)))
                    blk += 1
            for b in range(nt):
                for i in range(16):
                    r = bias_last[g, b, i]
                    if r >= 0:
                        got[r] = acc[b, i]
        assert not np.isnan(got).any()
        scale = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 2e-6 * scale, (np.abs(got - ref).max(), scale)


def _split2_f16(x: np.ndarray):
    x = np.asarray(x, dtype=np.float32)
    h = x.astype(np.float16).astype(np.float32)
    return h, (x - h).astype(np.float16).astype(np.float32)


def _pow2_scale(amax: float) -> float:
    """2^e with amax 2^e in [2^14, 2^15) — the scale of csrc/fused_ar_half_impl.h (arh_scale) and zuko_amd/fused.py (half_scales); 1 for zero."""
    import math

    return 1.0 if amax == 0 else 2.0 ** (15 - math.frexp(float(amax))[1])


@pytest.mark.parametrize("cfg", [("rqs", 64, 0, (256, 256, 256), 8), ("affine", 64, 0, (256, 256, 256), 0), ("rqs", 3, 5, (128, 128, 128), 8), ("rqs", 20, 3, (100, 72), 8),
                                 ("affine", 7, 2, (40,), 0)])
def test_half_kernel_tables_describe_the_plan(cfg):
    """Tables + gathers of the TWO-PART operand-split kernels (static_ar.half_tables; csrc/fused_ar_half_impl.h) walked on the CPU as the kernel
    walks them — the blocks of split_tables as two f16 images of the layer's weights times a power of two, the sample's activations times its own
    power of two, three partial products per block, one fma(acc, descale, bias) per output — must give the masked MLP's output to f32 accuracy;
    the weight scales and the eligibility verdict come from fused.half_scales as in the product."""
    from zuko_amd import fused, static_ar

    kind, D, C, hidden, bins = cfg
    rng = np.random.default_rng(12)
    for plan, lay, lins in static_ar._plans_for(kind, D, C, hidden, bins):
        th, ts = static_ar.half_tables(plan, lay.kind, 1), static_ar.split_tables(plan, lay.kind, 1)
        assert th is not None
        t, gathers = th
        t3 = ts[0]
        NH, nt = t["NH"], lay.nt
        assert (t["NB"], t["B_OT"], t["B_IP"], t["GOFF"], t["G_IP"]) == (t3["NB"], t3["B_OT"], t3["B_IP"], t3["GOFF"], t3["G_IP"]), "same blocks, same order as the three-part kernel"
        images = 2 * (sum(t["NB"]) + t["GOFF"][-1] * nt)
        assert t["NCHUNK"] == -(-images // t["CH"]) and t["BASE"] == [2 * sum(t["NB"][:l]) for l in range(NH)] and t["LAST_BASE"] == 2 * sum(t["NB"])
        assert t["NCHUNK"] * t["CH"] <= 2 * sum(len(g) // 512 for g in gathers) <= t["STREAM_IMAGES"] < t["NCHUNK"] * t["CH"] + 2
        for l in range(NH):
            assert np.array_equal(gathers[l], ts[1][l])
        scales = fused.half_scales(lins)
        assert all(ok for ok, _ in scales), "random-init weights are eligible"
        W = [l.weight.detach().numpy().astype(np.float32) * l.mask.numpy() for l in lins]
        for (ok, e), w in zip(scales, W):
            assert 2.0 ** 14 <= np.abs(w).max() * 2.0 ** e < 2.0 ** 15
        Bv = [l.bias.detach().numpy().astype(np.float64) for l in lins]
        x = rng.standard_normal(plan.din).astype(np.float32) * 3.0
        h = x.astype(np.float64)
        for l in range(NH + 1):
            h = W[l].astype(np.float64) @ h + Bv[l]
            if l < NH:
                h = np.maximum(h, 0.0)
        ref = h

        def images_of(l):
            idx = gathers[l].reshape(-1, 64, 8)
            flat = W[l].reshape(-1)
            vals = np.where(idx >= 0, flat[np.maximum(idx, 0)], 0.0).astype(np.float32) * np.float32(2.0 ** scales[l][1])
            return _split2_f16(vals)

        vec = np.zeros(t["TMAX"] * 16, dtype=np.float32)
        vec[: plan.din] = x
        boff = 0

        def operands(v):
            s_ = np.float32(_pow2_scale(np.abs(v).max()))
            return _split2_f16(v * s_), 1.0 / float(s_)

        for l in range(NH):
            ah, al = images_of(l)
            bias_img = np.where(plan.bias_gather[l] >= 0, Bv[l][np.maximum(plan.bias_gather[l], 0)], 0.0)[: t["TMAX"] * 16]
            (vh, vl), inv_s = operands(vec)
            acc = np.zeros(t["TMAX"] * 16)
            for s in range(t["NB"][l]):
                ot, ip = t["B_OT"][boff + s], t["B_IP"][boff + s]
                for lane in range(64):
                    i, kq = lane % 16, lane // 16
                    units = np.concatenate([np.arange(4) + (2 * ip) * 16 + 4 * kq, np.arange(4) + (2 * ip + 1) * 16 + 4 * kq])
                    for a_, b_ in ((al, vh), (ah, vl), (ah, vh)):
                        acc[ot * 16 + i] += float(np.dot(a_[s, lane].astype(np.float64), b_[units].astype(np.float64)))
            out = acc * (2.0 ** -scales[l][1] * inv_s) + bias_img
            boff += t["NB"][l]
            vec = np.zeros_like(vec)
            vec[: t["HT"][l] * 16] = np.maximum(out[: t["HT"][l] * 16], 0.0).astype(np.float32)
        ah, al = images_of(NH)
        (vh, vl), inv_s = operands(vec)
        total = lay.total
        got = np.full(D * total, np.nan)
        blk = 0
        bias_last = plan.bias_gather[NH].reshape(-1, nt, 16)
        for g in range(t["NG"]):
            acc = np.zeros((nt, 16))
            for st in range(t["GOFF"][g], t["GOFF"][g + 1]):
                ip = t["G_IP"][st]
                for b in range(nt):
                    for lane in range(64):
                        i, kq = lane % 16, lane // 16
                        units = np.concatenate([np.arange(4) + (2 * ip) * 16 + 4 * kq, np.arange(4) + (2 * ip + 1) * 16 + 4 * kq])
                        for a_, b_ in ((al, vh), (ah, vl), (ah, vh)):
                            acc[b, i] += float(np.dot(a_[blk, lane].astype(np.float64), b_[units].astype(np.float64)))
                    blk += 1
            for b in range(nt):
                for i in range(16):
                    r = bias_last[g, b, i]
                    if r >= 0:
                        got[r] = acc[b, i] * (2.0 ** -scales[NH][1] * inv_s) + Bv[NH][r]
        assert not np.isnan(got).any()
        scale = np.abs(ref).max()
        assert np.abs(got - ref).max() <= 4e-6 * scale, (np.abs(got - ref).max(), scale)


def test_half_eligibility_follows_the_weights():
    """fused.half_scales: per-layer power of two from the largest masked magnitude; a layer whose magnitudes spread beyond what two f16 parts
    carry (many weights 2^14 below the largest, an out unit 2^12 below, a magnitude outside 2^+-40, a non-finite weight) is not eligible."""
    import zuko_amd.flows as F
    from zuko_amd import fused
    from zuko_amd.nn import MaskedLinear

    def lins_of(seed=0):
        torch.manual_seed(seed)
        lazy = F.NSF(8, 0, transforms=1, hidden_features=[64, 64]).transform.transforms[0]
        return [m for m in lazy.hyper if isinstance(m, MaskedLinear)]

    lins = lins_of()
    base = fused.half_scales(lins)
    assert all(ok for ok, _ in base)
    with torch.no_grad():
        lins[1].weight.mul_(2.0 ** -12)  # a whole layer far down: its own scale follows, still eligible
    moved = fused.half_scales(lins)
    assert moved[1] == (True, base[1][1] + 12) and moved[0] == base[0] and moved[2] == base[2]
    with torch.no_grad():
        lins = lins_of()
        g = torch.Generator().manual_seed(1)
        lins[0].weight.mul_(torch.exp2(torch.rand(lins[0].weight.shape, generator=g) * 24.0 - 20.0))  # the "wide-range" regime of tests/test_gpu_flows.py
    assert [ok for ok, _ in fused.half_scales(lins)] == [False, True, True]
    with torch.no_grad():
        lins = lins_of()
        row = int(lins[2].mask.any(dim=1).nonzero()[0])
        lins[2].weight[row].mul_(2.0 ** -16)  # one out unit's weights far below the layer's
    assert [ok for ok, _ in fused.half_scales(lins)] == [True, True, False]
    with torch.no_grad():
        lins = lins_of()
        lins[1].weight.mul_(2.0 ** -30)
        lins[2].weight.mul_(2.0 ** 45)
    assert [ok for ok, _ in fused.half_scales(lins)] == [True, False, False]
    with torch.no_grad():
        lins = lins_of()
        lins[0].weight[lins[0].mask][0] = float("inf")
        w = lins[0].weight
        idx = lins[0].mask.nonzero()[0]
        w[idx[0], idx[1]] = float("nan")
    assert fused.half_scales(lins)[0][0] is False
    with torch.no_grad():
        lins = lins_of()
        lins[1].weight.zero_()
    assert fused.half_scales(lins)[1] == (True, 0)


def test_matmul_precision_switch():
    import zuko_amd
    from zuko_amd import fused

    keep = fused.matmul_precision()
    try:
        for name, mode in (("highest", "bf16x3"), ("bf16x3", "bf16x3"), ("high", "f16x2"), ("F16x2", "f16x2")):
            zuko_amd.set_matmul_precision(name)
            assert zuko_amd.matmul_precision() == mode
        with pytest.raises(ValueError):
            zuko_amd.set_matmul_precision("tf32")
    finally:
        zuko_amd.set_matmul_precision(keep)


def test_coupling_split_stream_is_the_f32_stream_regrouped():
    """coupling_plan.build_coupling_plan's operand-split stream (cfg4's shape: 128 inputs, hidden [512] * 3): block (out tile, in pair) =
    the f32 stream's images of in tiles 2 ip and 2 ip + 1 side by side, in the kernel's step order (4 out tiles x 1 in pair; the last layer
    group by group), every layer and every group on a chunk boundary."""
    import zuko_amd.flows as F
    from zuko_amd import coupling_plan as cp

    torch.manual_seed(1)
    t = F.RealNVP(256, 0, transforms=2, hidden_features=[512] * 3).transform.transforms[1]
    lins = list(t.hyper)[0::2]
    idx_a, idx_b = t.mask.nonzero().squeeze(-1).numpy(), (~t.mask).nonzero().squeeze(-1).numpy()
    plan = cp.build_coupling_plan([tuple(l.weight.shape) for l in lins], idx_a, idx_b, 256, 0)
    assert plan.split_gather is not None and plan.split_chunks == (128 + 2 * 512 + 256) * 3 // cp.CHUNK
    f32 = plan.gather.reshape(-1, 64, 4)
    sp = plan.split_gather.reshape(-1, 64, 8)
    pos32 = poss = 0
    for l in range(3):
        n_in = 8 if l == 0 else 32
        for otg in range(8):
            for ip in range(n_in // 2):
                for tt in range(4):
                    for half in range(2):
                        src = f32[pos32 + (otg * n_in + 2 * ip + half) * 4 + tt]
                        assert np.array_equal(sp[poss][:, 4 * half : 4 * half + 4], src)
                    poss += 1
        pos32 = -(-(pos32 + 8 * n_in * 4) // cp.CHUNK) * cp.CHUNK  # (the f32 stream pads every layer to whole chunks)
        assert (3 * poss) % cp.CHUNK == 0
    for g in range(plan.n_groups):
        for ip in range(16):
            for half in range(2):
                assert np.array_equal(sp[poss][:, 4 * half : 4 * half + 4], f32[pos32 + g * 32 + 2 * ip + half])
            poss += 1
        assert (3 * poss) % cp.CHUNK == 0
    assert poss == sp.shape[0]
    # small or odd shapes keep the f32 stream only
    t2 = F.RealNVP(12, 2, transforms=1, hidden_features=[40, 70, 24]).transform.transforms[0]
    l2 = list(t2.hyper)[0::2]
    p2 = cp.build_coupling_plan([tuple(l.weight.shape) for l in l2], t2.mask.nonzero().squeeze(-1).numpy(), (~t2.mask).nonzero().squeeze(-1).numpy(), 12, 2)
    assert p2.split_gather is None


def test_polynomial_flows_are_recognised_at_their_default_sizes_only():
    """`_fusable_layout` of a lazy autoregressive transform: SOSPF / BPF at the reference's default sizes (zuko/flows/polynomial.py:51-53, :97) map to the
    fused kernels' uni kinds 5 / 6 with the maps' own constants; other sizes (and Bernstein maps given a slope) stay on the layer-wise path."""
    import zuko_amd.flows as F

    lz = F.SOSPF(6, 2, transforms=1, hidden_features=[32]).transform.transforms[0]
    lay, bound, slope = lz._fusable_layout()
    assert (lay.kind, lay.total, lay.nt, lay.fpl) == (5, 16, 4, 1) and bound == 10.0 and slope == 1e-3
    lz = F.SOSPF(6, 0, transforms=1, hidden_features=[32], slope=1e-2).transform.transforms[0]
    assert lz._fusable_layout()[2] == 1e-2
    assert F.SOSPF(6, 0, degree=3, transforms=1, hidden_features=[32]).transform.transforms[0]._fusable_layout() is None
    assert F.SOSPF(6, 0, polynomials=2, transforms=1, hidden_features=[32]).transform.transforms[0]._fusable_layout() is None
    lz = F.BPF(6, 0, transforms=1, hidden_features=[32]).transform.transforms[0]
    lay, bound, eps = lz._fusable_layout()
    assert (lay.kind, lay.total, lay.nt, lay.fpl) == (6, 17, 5, 1) and bound == 5.0 and eps == 1e-6
    assert F.BPF(6, 0, degree=8, transforms=1, hidden_features=[32]).transform.transforms[0]._fusable_layout() is None
    # the inverse of these kinds stays layer-wise: no group-aligned fused state for the sweeps
    assert F.BPF(6, 0, transforms=1, hidden_features=[32]).transform.transforms[0].fused_state(__import__("torch").device("cpu"), inverse=True) is None


def test_lane_owned_spline_panels_follow_the_header_formula():
    """zk_linear_bf16_rqs_lanes (include/zuko_amd.h) wants the last layer's rows in panels of 192 in which the 16 outputs a lane of the matrix
    instruction owns per 32-row block hold whole features: _Bf16Plan.spline_lane_panels against a loop restatement of the header's formula, for
    both bin counts, a feature count that leaves the last panel partly empty, and the 192 x 32 liveness words (host logic: runs on the CPU)."""
    import torch

    from zuko_amd.nn import MaskedMLP, _Bf16Plan

    for K, features in ((16, 10), (8, 21)):
        total = 3 * K - 1
        ts = 48 if K == 16 else 24
        fpl = 48 // ts
        fpp = 4 * fpl
        torch.manual_seed(K)
        order = torch.arange(features)
        adjacency = (order[:, None] > order[None, :]).repeat_interleave(total, dim=0)  # row f * total + j (zuko/flows/autoregressive.py:188-190)
        net = MaskedMLP(adjacency, hidden_features=[64, 128])
        lins = list(net)[0::2]
        plan = _Bf16Plan(lins)
        plan.refresh(lins)
        wp, bp, live = plan.spline_lane_panels(lins, K, features)
        w, b, mp = plan.weights[-1], plan.biases[-1], plan.masks_p[-1]
        panels = -(-features // fpp)
        assert wp.shape == (panels * 192, w.shape[1]) and bp.shape == (panels * 192,) and live.shape == (panels,)
        seen = set()
        for p in range(panels):
            for o in range(192):
                wn, c = divmod(o, 96)
                j, q, kg, t = c // 32, (c % 32) // 8, (c % 8) // 4, c % 4
                slot = 16 * j + 4 * q + t
                feat, par = p * fpp + wn * 2 * fpl + kg * fpl + slot // ts, slot % ts
                row = p * 192 + o
                if par < total and feat < features:
                    src = feat * total + par
                    seen.add(src)
                    assert torch.equal(wp[row], w[src]) and bp[row] == b[src], (K, p, o)
                else:
                    assert not wp[row].any() and bp[row] == 0, (K, p, o)
        assert seen == set(range(features * total))  # every parameter row of the reference's layout sits in exactly one slot
        kt = w.shape[1] // 32
        for p in range(panels):
            for k in range(kt):
                assert bool((int(live[p]) >> k) & 1) == bool(_panel_mask_any(plan, lins, K, features, p, k)), (K, p, k)


def _panel_mask_any(plan, lins, K, features, p, k):
    """Does panel p of the lane-owned layout hold a row whose MASK is non-zero in inputs [32 k, 32 k + 32)?  (liveness follows the mask, not the
    weights' values: a weight that happens to be zero today may be trained tomorrow without the plan being rebuilt)"""
    total = 3 * K - 1
    ts = 48 if K == 16 else 24
    fpl = 48 // ts
    fpp = 4 * fpl
    mp = plan.masks_p[-1].bool()
    for o in range(192):
        wn, c = divmod(o, 96)
        j, q, kg, t = c // 32, (c % 32) // 8, (c % 8) // 4, c % 4
        slot = 16 * j + 4 * q + t
        feat, par = p * fpp + wn * 2 * fpl + kg * fpl + slot // ts, slot % ts
        if par < total and feat < features and bool(mp[feat * total + par, 32 * k : 32 * k + 32].any()):
            return True
    return False


@pytest.mark.parametrize("cfg", [("rqs", 64, 0, (256, 256, 256), 8), ("affine", 10, 3, (40, 72), 0), ("rqs", 3, 5, (128, 128, 128), 8), ("rqs", 20, 3, (100, 72), 8), ("rqs", 8, 2, (48, 48), 4),
                                 ("rqs", 8, 0, (64,), 16), ("affine", 2, 0, (24, 24), 0)])
def test_generic_split_stream_follows_the_kernels_walk(cfg):
    """The stream of the generic operand-split kernel (fused.gsplit_gather; csrc/fused_ar_gsplit.hip) walked as that kernel walks it — the
    plan's own skip words, an in-PAIR live when either of its tiles is, 4 blocks per (out-group, pair) / NT per (feature group, pair), the ring
    moving on in whole chunks of 8 blocks at every layer end — must consume exactly n_chunks chunks and give the masked MLP (float64: the
    bf16 x 3 arithmetic is tests/test_fused_plan.py::test_split_kernel_tables_describe_the_plan's subject)."""
    from zuko_amd import fused, static_ar

    kind, D, C, hidden, bins = cfg
    rng = np.random.default_rng(5)
    for plan, lay, lins in static_ar._plans_for(kind, D, C, hidden, bins):
        got = fused.gsplit_gather(plan)
        assert got is not None
        gathers, offsets, n_chunks = got
        assert all(len(g) % (512 * fused.GS_BLOCKS_PER_CHUNK) == 0 for g in gathers) and offsets[0] == 0
        assert [o // 768 for o in offsets] == [sum(len(g) // 512 for g in gathers[:l]) for l in range(len(gathers))]
        W = [(l.weight.detach().double().numpy() * l.mask.numpy()) for l in lins]
        Bv = [l.bias.detach().double().numpy() for l in lins]
        x = rng.standard_normal(plan.din)
        h = x.copy()
        for l in range(len(W)):
            h = W[l] @ h + Bv[l]
            if l + 1 < len(W):
                h = np.maximum(h, 0.0)
        ref = h
        stream = np.concatenate([np.where(g >= 0, W[l].reshape(-1)[np.maximum(g, 0)], 0.0) for l, g in enumerate(gathers)]).reshape(-1, 64, 8)
        pos = 0  # block cursor

        def block(ip, vec):  # one 16 x 32 block times the pair's 32 activations -> 16 outputs
            nonlocal pos
            blk = stream[pos]
            pos += 1
            out = np.zeros(16)
            for lane in range(64):
                i, kq = lane % 16, lane // 16
                units = np.concatenate([np.arange(4) + (2 * ip) * 16 + 4 * kq, np.arange(4) + (2 * ip + 1) * 16 + 4 * kq])
                out[i] += blk[lane] @ vec[units]
            return out

        def end_layer():
            nonlocal pos
            pos = -(-pos // fused.GS_BLOCKS_PER_CHUNK) * fused.GS_BLOCKS_PER_CHUNK

        vec = np.zeros(256)
        vec[: plan.din] = x
        for l in range(plan.n_layers - 1):
            assert pos * 768 == offsets[l]
            out = np.where(plan.bias_gather[l] >= 0, Bv[l][np.maximum(plan.bias_gather[l], 0)], 0.0)
            for otg in range(4):
                bits = int(plan.skip[l * 4 + otg])
                for ip in range(8):
                    if (bits | bits >> 1) & 0x5555 & (1 << 2 * ip):
                        for t in range(4):
                            out[(otg * 4 + t) * 16 : (otg * 4 + t + 1) * 16] += block(ip, vec)
            end_layer()
            vec = np.maximum(out, 0.0)
        assert pos * 768 == offsets[-1]
        nt = lay.nt
        bias_last = plan.bias_gather[-1].reshape(-1, nt, 16)
        res = np.full(D * lay.total, np.nan)
        for g in range(plan.n_groups):
            acc = np.where(bias_last[g] >= 0, Bv[-1][np.maximum(bias_last[g], 0)], 0.0)
            bits = int(plan.skip[(plan.n_layers - 1) * 4 + g])
            for ip in range(8):
                if (bits | bits >> 1) & 0x5555 & (1 << 2 * ip):
                    for t in range(nt):
                        acc[t] += block(ip, vec)
            for t in range(nt):
                for i in range(16):
                    if bias_last[g, t, i] >= 0:
                        res[bias_last[g, t, i]] = acc[t, i]
        end_layer()
        assert pos == max(0, n_chunks * fused.GS_BLOCKS_PER_CHUNK) or (pos == 0 and n_chunks == 1)
        assert not np.isnan(res).any() and np.abs(res - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


def test_generic_split_kernel_is_not_offered_where_it_does_not_exist():
    from zuko_amd import fused
    from zuko_amd.flows import NSF

    t = NSF(8, 0, transforms=1, hidden_features=[64, 64]).transform.transforms[0]
    masks = [m.mask for m in t.hyper if hasattr(m, "mask")]
    lay = fused.uni_layout("rqs", 23, 8)
    assert fused.gsplit_gather(fused.build_plan(masks, 8, lay)) is not None
    assert fused.gsplit_gather(fused.build_plan(masks, 8, lay, align_groups=True)) is None  # (the wavefront inverse's plans: f32 kernel)


def test_generic_split_kernel_selection(monkeypatch):
    """Which kernel FusedAR.run() launches (host logic, no GPU): the generic operand-split kernel exactly when no generated kernel is held, unless
    ZUKO_AMD_GSPLIT=0 / ZUKO_AMD_EXACT_F32=1 keep the f32 matrix instruction; `force` prefers it even beside a generated kernel; its stream is sized
    in whole chunks of 8 blocks of three 1 KiB images."""
    from zuko_amd import fused
    from zuko_amd.flows import NSF

    monkeypatch.setenv("ZUKO_AMD_JIT", "0")
    monkeypatch.delenv("ZUKO_AMD_GSPLIT", raising=False)
    monkeypatch.delenv("ZUKO_AMD_EXACT_F32", raising=False)
    cpu = torch.device("cpu")
    st = NSF(12, 3, transforms=1, hidden_features=[88, 120]).transform.transforms[0].fused_state(cpu)  # (not a prebuilt shape)
    assert st is not None and st.static is None and st.generic_ok
    gs = st._gsplit()
    assert gs is not None
    gathers, offsets, n_chunks, stream = gs
    assert stream.numel() == n_chunks * fused.GS_BLOCKS_PER_CHUNK * 768 and sum(g.numel() for g in gathers) == n_chunks * fused.GS_BLOCKS_PER_CHUNK * 512
    assert offsets == [768 * sum(g.numel() // 512 for g in gathers[:l]) for l in range(len(gathers))]
    st.gs_mode = "0"
    assert st._gsplit() is None
    st.gs_mode = "1"
    monkeypatch.setenv("ZUKO_AMD_EXACT_F32", "1")
    assert st._gsplit() is None
    monkeypatch.delenv("ZUKO_AMD_EXACT_F32")
    st2 = NSF(64, 0, transforms=1, hidden_features=[256] * 3).transform.transforms[0].fused_state(cpu)  # the headline conditioner: prebuilt
    if st2.static is not None:
        assert st2._gsplit() is None
        st2.gs_mode = "force"
        assert st2._gsplit() is not None
