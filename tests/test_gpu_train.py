"""GPU tests of the training GEMMs (csrc/train.hip: tile-skipping forward / dgrad, split-K wgrad, column sums) and of
the sorted-domain conditioner (zuko_amd/train.py) against plain torch ops on the same device."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def _tri_mask(out_f, in_f, gen):
    """A MADE-like mask: row r may see column c iff deg_in[c] <= deg_out[r] (random degrees)."""
    dout = torch.randint(0, 64, (out_f,), generator=gen)
    din = torch.randint(0, 64, (in_f,), generator=gen)
    return dout[:, None] >= din[None, :]


@pytest.mark.parametrize("N,IN,OUT", [(1000, 256, 256), (77, 64, 200), (4096, 256, 1472), (513, 1472, 256), (300, 37, 50)])
def test_gemm_skip_wgrad_colsum(dev, N, IN, OUT):
    from zuko_amd.train import SortedPlan

    gen = torch.Generator().manual_seed(N + IN)
    x = torch.randn(N, IN, generator=gen).to(dev)
    w = (torch.randn(OUT, IN, generator=gen) / IN**0.5).to(dev)
    b = torch.randn(OUT, generator=gen).to(dev)
    mask = _tri_mask(OUT, IN, gen)
    # sort rows / columns by degree so that whole tiles vanish, as the plan does
    ro, co = torch.argsort(mask.sum(1), stable=True), torch.argsort(-mask.sum(0), stable=True)
    mask = mask[ro][:, co].to(dev)
    wm = (w * mask).contiguous()
    ks = SortedPlan._kskip(mask.cpu())
    ks = None if ks is None else ks.to(dev)

    class P:  # just enough of a plan to call the wrappers
        pass

    plan = SortedPlan.__new__(SortedPlan)
    ref = torch.relu(x @ wm.t() + b)
    y = SortedPlan.gemm(plan, x, wm, ks, b, 1)
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5 * IN**0.5)
    assert torch.equal(SortedPlan.gemm(plan, x, wm, None, b, 1), y), "skipping all-zero tiles must not change a single bit"
    gate = torch.relu(torch.randn(N, OUT, generator=gen)).to(dev)
    yg = SortedPlan.gemm(plan, x, wm, ks, None, 0, gate, 1)
    assert torch.allclose(yg, (x @ wm.t()) * (gate > 0), rtol=1e-5, atol=1e-5 * IN**0.5)
    # wgrad: dW = mask * g^T h over the live blocks
    g = torch.randn(N, OUT, generator=gen).to(dev)
    plan.shapes = [(OUT, IN)]
    ob, ib = -(-OUT // 128), -(-IN // 128)
    pad = torch.zeros((ob * 128, ib * 128), dtype=torch.bool)
    pad[:OUT, :IN] = mask.cpu()
    plan.pairs = [pad.reshape(ob, 128, ib, 128).any(3).any(1).nonzero().to(torch.int32).contiguous().to(dev)]
    plan.mask_s = [mask.to(torch.uint8).contiguous()]
    plan.idx_b, plan.cols_dev = [None], [None]  # identity: dW in the operands' own order
    dw = plan.wgrad(0, g, x)
    refw = (g.double().t() @ x.double()) * mask
    assert torch.allclose(dw.double(), refw, rtol=1e-5, atol=2e-5 * refw.abs().max().item())
    assert torch.equal(plan.wgrad(0, g, x), dw), "wgrad must be deterministic"
    # row / column permutations: element (o, c) is written to dW[rows[o], cols[c]] (the module's unit order), bit for bit the same values
    rp, cp_ = torch.randperm(OUT, generator=gen), torch.randperm(IN, generator=gen)
    plan.idx_b, plan.cols_dev = [rp.to(torch.int32).to(dev)], [cp_.to(torch.int32).to(dev)]
    dwp = plan.wgrad(0, g, x)
    assert torch.equal(dwp[rp.to(dev)][:, cp_.to(dev)], dw)
    plan.idx_b, plan.cols_dev = [None], [None]
    cs = plan.colsum(g)
    assert torch.allclose(cs.double(), g.double().sum(0), rtol=1e-5, atol=1e-4)
    pr, seen = plan.pairs[0].cpu(), set()
    flag = torch.zeros(pr.shape[0], dtype=torch.uint8)
    for k in range(pr.shape[0]):
        if int(pr[k, 0]) not in seen:
            seen.add(int(pr[k, 0]))
            flag[k] = 1
    plan.cs_flag = [flag.to(dev) if len(seen) == ob else None]
    dw2, db2 = plan.wgrad(0, g, x, want_bias=True)  # bias gradient from the same pass over g
    assert torch.equal(dw2, dw)
    if plan.cs_flag[0] is not None:
        assert torch.allclose(db2.double(), g.double().sum(0), rtol=1e-5, atol=1e-4)
        assert torch.equal(plan.wgrad(0, g, x, want_bias=True)[1], db2), "deterministic"
        plan.idx_b, plan.cols_dev = [rp.to(torch.int32).to(dev)], [cp_.to(torch.int32).to(dev)]
        assert torch.equal(plan.wgrad(0, g, x, want_bias=True)[1][rp.to(dev)], db2), "the bias gradient follows the row permutation like dW"
        plan.idx_b, plan.cols_dev = [None], [None]
    else:
        assert db2 is None


@pytest.mark.parametrize("kind", ["nsf64", "maf_ctx", "coupling"])
def test_conditioner_gradients_match_torch(dev, kind):
    """hyper(x) through ConditionerFn (sorted domain, skipping) against autograd through plain torch ops on the module's own
    (unsorted) parameters: phi, d phi / d x and every parameter gradient."""
    import zuko_amd.flows as F
    from zuko_amd.nn import MaskedLinear

    torch.manual_seed(3)
    if kind == "nsf64":
        t = F.NSF(64, 0, transforms=1, bins=8, hidden_features=[256] * 3).transform.transforms[0]
        din = 64
    elif kind == "maf_ctx":
        t = F.MAF(10, 3, transforms=1, hidden_features=[48, 40], activation=torch.nn.Tanh).transform.transforms[0]
        din = 13
    else:
        t = F.RealNVP(12, 2, transforms=1, hidden_features=[64, 64]).transform.transforms[0]
        din = t.hyper[0].weight.shape[1]
    net = t.hyper.to(dev)
    N = 700
    x = torch.randn(N, din, generator=torch.Generator().manual_seed(5)).to(dev)
    gphi = None

    def run(use_hip):
        nonlocal gphi
        for p in net.parameters():
            p.grad = None
        xr = x.clone().requires_grad_()
        if use_hip:
            out = net(xr)
        else:
            h = xr
            mods = list(net)
            for m in mods:
                if hasattr(m, "weight"):
                    wgt = m.weight * m.mask if isinstance(m, MaskedLinear) else m.weight
                    h = h @ wgt.t() + m.bias
                else:
                    h = m(h)
            out = h
        if gphi is None:
            gphi = torch.randn(out.shape, generator=torch.Generator().manual_seed(9)).to(dev)
        (out * gphi).sum().backward()
        return out.detach(), xr.grad.clone(), [p.grad.clone() for p in net.parameters()]

    o1, gx1, gp1 = run(True)
    o0, gx0, gp0 = run(False)
    assert torch.allclose(o1, o0, rtol=1e-5, atol=2e-5)
    assert torch.allclose(gx1, gx0, rtol=1e-4, atol=1e-4 * gx0.abs().max().item())
    for a, b in zip(gp1, gp0):
        assert torch.allclose(a, b, rtol=1e-4, atol=2e-5 * b.abs().max().clamp_min(1e-6).item()), (a - b).abs().max().item()
    for m in net.modules():
        if isinstance(m, MaskedLinear):
            assert (m.weight.grad[~m.mask] == 0).all()


@pytest.mark.parametrize("kind", ["nsf", "maf"])
def test_fused_training_forward(dev, kind, monkeypatch):
    """zk_ar_forward_train (the static-shape kernel as the conditioner's forward under autograd) must be selected for the cfg2 / cfg3
    conditioner and give the layer-wise kernels' phi, hidden activations and gradients (ragged batch, a poisoned row)."""
    import zuko_amd.flows as F
    from zuko_amd import train
    from zuko_amd.nn import MaskedLinear

    torch.manual_seed(4)
    flow = (F.NSF(64, 0, transforms=1, bins=8, hidden_features=[256] * 3) if kind == "nsf" else F.MAF(64, 0, transforms=1, hidden_features=[256] * 3)).to(dev)
    net = flow.transform.transforms[0].hyper
    plan, lins = train.plan_for(net, dev)
    st = train._fused_forward_state(plan, lins, dev)
    assert st is not None, "the fused training forward must be available for this conditioner"
    N = 1000 + 37
    x = torch.randn(N, 64, generator=torch.Generator().manual_seed(6)).to(dev)
    x[11, 3] = float("nan")
    hs_f, phi_f = train._fused_forward(st, x, lins[-1].weight.shape[0])
    ws, _, bs = plan.gather(lins)
    h = x
    for l in range(4):
        h = plan.gemm(h, ws[l], plan.kskip_f[l], bs[l], 1 if l < 3 else 0)
        ref = hs_f[l] if l < 3 else phi_f
        ok = torch.ones(N, dtype=torch.bool, device=dev)
        ok[11] = False
        assert torch.allclose(ref[ok], h[ok], rtol=1e-5, atol=2e-5), (l, (ref[ok] - h[ok]).abs().max().item())
    assert torch.isnan(phi_f[11]).all(), "a non-finite input makes every parameter of its sample NaN (zuko/nn.py:217-218)"

    xg = x.clone()
    xg[11, 3] = 0.5
    gphi = torch.randn(N, lins[-1].weight.shape[0], generator=torch.Generator().manual_seed(9)).to(dev)

    def run():
        for p in net.parameters():
            p.grad = None
        xr = xg.clone().requires_grad_()
        out = net(xr)
        (out * gphi).sum().backward()
        return out.detach(), xr.grad.clone(), [p.grad.clone() for p in net.parameters()]

    o1, gx1, gp1 = run()
    monkeypatch.setenv("ZUKO_AMD_NO_FUSED_TRAIN", "1")
    o0, gx0, gp0 = run()
    assert torch.allclose(o1, o0, rtol=1e-5, atol=2e-5)
    assert torch.allclose(gx1, gx0, rtol=1e-4, atol=1e-4 * gx0.abs().max().item())
    for a, b in zip(gp1, gp0):
        assert torch.allclose(a, b, rtol=1e-4, atol=2e-5 * b.abs().max().clamp_min(1e-6).item()), (a - b).abs().max().item()


@pytest.mark.parametrize("kind", ["nsf_asc", "nsf_desc", "maf_desc", "cfg1", "two_layers"])
def test_dgrad_chain_matches_the_layerwise_backward(dev, kind, monkeypatch):
    """zk_ar_dgrad_chain (one generated kernel for the dgrad of every layer but the last) must be selected for the benchmark
    conditioners and give the layer-wise GEMMs' gradients: d/dx and every parameter gradient, ragged batch, both feature orders."""
    import zuko_amd.flows as F
    from zuko_amd import train

    torch.manual_seed(8)
    if kind.startswith("nsf"):
        flow = F.NSF(64, 0, transforms=2, bins=8, hidden_features=[256] * 3)
    elif kind == "maf_desc":
        flow = F.MAF(64, 0, transforms=2, hidden_features=[256] * 3)
    elif kind == "cfg1":
        flow = F.NSF(3, 5, transforms=2, bins=8, hidden_features=[128] * 3)
    else:
        flow = F.MAF(16, 0, transforms=2, hidden_features=[128, 128])
    net = flow.to(dev).transform.transforms[1 if kind.endswith("desc") else 0].hyper
    plan, lins = train.plan_for(net, dev)
    N = 1000 + 37
    assert train._dgrad_chain(plan, lins, N) is not None, "the dgrad chain must be available (prebuilt) for this conditioner"
    din = lins[0].weight.shape[1]
    x = torch.randn(N, din, generator=torch.Generator().manual_seed(6)).to(dev)
    gphi = torch.randn(N, lins[-1].weight.shape[0], generator=torch.Generator().manual_seed(9)).to(dev)

    def run():
        for p in net.parameters():
            p.grad = None
        xr = x.clone().requires_grad_()
        out = net(xr)
        (out * gphi).sum().backward()
        return xr.grad.clone(), [p.grad.clone() for p in net.parameters()]

    gx1, gp1 = run()
    monkeypatch.setenv("ZUKO_AMD_NO_DGRAD_CHAIN", "1")
    gx0, gp0 = run()
    assert torch.allclose(gx1, gx0, rtol=1e-5, atol=1e-5 * gx0.abs().max().item()), (gx1 - gx0).abs().max().item()
    for a, b in zip(gp1, gp0):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * b.abs().max().clamp_min(1e-6).item()), (a - b).abs().max().item()
    # the forward's saved stream, not the parameters at backward time, enters the gradient (as autograd's saved tensors would)
    xr = x.clone().requires_grad_()
    monkeypatch.delenv("ZUKO_AMD_NO_DGRAD_CHAIN")
    out = net(xr)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(3.0)
    (out * gphi).sum().backward()
    with torch.no_grad():
        for p in net.parameters():
            p.div_(3.0)
    assert torch.allclose(xr.grad, gx0, rtol=1e-5, atol=1e-5 * gx0.abs().max().item())


@pytest.mark.parametrize("kind", ["nsf", "maf"])
def test_wgrad_multi_matches_the_per_layer_launches(dev, kind, monkeypatch):
    """zk_wgrad_multi (the weight / bias gradients of all layers of a conditioner in two launches) against one zk_wgrad_bias_f32 per
    layer: the same kernel body on the same slices, so every parameter gradient must be bit-identical; ragged batch."""
    import zuko_amd.flows as F

    torch.manual_seed(12)
    flow = (F.NSF(64, 0, transforms=2, bins=8, hidden_features=[256] * 3) if kind == "nsf" else F.MAF(64, 0, transforms=2, hidden_features=[256] * 3)).to(dev)
    net = flow.transform.transforms[1].hyper
    N = 3000 + 11
    x = torch.randn(N, 64, generator=torch.Generator().manual_seed(6)).to(dev)
    gphi = torch.randn(N, net[-1].weight.shape[0], generator=torch.Generator().manual_seed(9)).to(dev)

    def run():
        for p in net.parameters():
            p.grad = None
        xr = x.clone().requires_grad_()
        (net(xr) * gphi).sum().backward()
        return xr.grad.clone(), [p.grad.clone() for p in net.parameters()]

    gx1, gp1 = run()
    monkeypatch.setenv("ZUKO_AMD_NO_WGRAD_MULTI", "1")
    gx0, gp0 = run()
    assert torch.equal(gx1, gx0)
    for a, b in zip(gp1, gp0):
        assert torch.equal(a, b), (a - b).abs().max().item()


def test_gather_multi_equals_the_single_gathers(dev):
    """zk_gather_multi (up to eight f32 / operand-split gathers per launch) writes exactly what zk_gather_f32 / zk_gather_split_bf16 write,
    including masked entries, -1 indices, an empty gather and a list longer than eight."""
    from zuko_amd import _C
    from zuko_amd.ops import _ptr, _stream

    gen = torch.Generator().manual_seed(4)
    lib = _C.lib()
    items, want = [], []
    for k in range(11):
        n_src = 500 + 37 * k
        src = torch.randn(n_src, generator=gen).to(dev)
        mask = (torch.rand(n_src, generator=gen) < 0.7).to(torch.uint8).to(dev) if k % 3 else None
        split = k % 2
        count = 0 if k == 5 else (3 + k if split else 1000 + 13 * k)
        n_idx = count * 512 if split else count
        idx = torch.randint(-1, n_src, (max(n_idx, 1),), generator=gen, dtype=torch.int32).to(dev)
        n_out = count * 3 * 256 if split else count
        dst, ref = torch.full((max(n_out, 1),), 7.0, device=dev), torch.full((max(n_out, 1),), 7.0, device=dev)
        if count:
            if split:
                _C.check(lib.zk_gather_split_bf16(_ptr(src), _ptr(mask), _ptr(idx), count, _ptr(ref), _stream()), "zk_gather_split_bf16")
            else:
                _C.check(lib.zk_gather_f32(_ptr(src), _ptr(mask), _ptr(idx), count, _ptr(ref), _stream()), "zk_gather_f32")
        items.append((src, mask, idx, count, dst, split))
        want.append(ref)
    _C.gather_multi(items, _stream())
    torch.cuda.synchronize()
    for (src, mask, idx, count, dst, split), ref in zip(items, want):
        assert torch.equal(dst.view(torch.int32), ref.view(torch.int32)), (count, split)


@pytest.mark.parametrize("shape", [(300, 37, 50), (1000, 128, 256), (5, 3, 7), (2049, 260, 129)])
@pytest.mark.parametrize("masked", [False, True])
def test_layerwise_linear_backward_has_no_library_gemm(dev, shape, masked):
    """Module trees the sorted plans do not cover (residual blocks; zuko/nn.py:297-309) differentiate layer by layer through
    zuko_amd/autograd.py:LinearFn, whose dgrad / wgrad / bias gradient are the tile kernels of csrc/train.hip (until round 4: torch.mm): against
    float64 autograd of F.linear(x, mask * W, b) (zuko/nn.py:217-218)."""
    from zuko_amd import ops

    N, in_f, out_f = shape
    g = torch.Generator().manual_seed(N + in_f)
    x = torch.randn(N, in_f, generator=g)
    w = torch.randn(out_f, in_f, generator=g) / in_f**0.5
    b = torch.randn(out_f, generator=g)
    mask = (torch.rand(out_f, in_f, generator=g) > 0.4) if masked else None
    gy = torch.randn(N, out_f, generator=g)

    xr, wr, br = (t.double().requires_grad_() for t in (x, w, b))
    yr = torch.nn.functional.elu(torch.nn.functional.linear(xr, wr if mask is None else wr * mask.double(), br))
    yr.backward(gy.double())

    xd, wd, bd = (t.to(dev).requires_grad_() for t in (x, w, b))
    y = ops.linear(xd, wd, bd, None if mask is None else mask.to(dev), 2)  # (2 = ELU)
    assert type(y.grad_fn).__name__ == "LinearFnBackward"
    y.backward(gy.to(dev))
    for name, mine, ref in (("x", xd.grad, xr.grad), ("W", wd.grad, wr.grad), ("b", bd.grad, br.grad)):
        err = ((mine.cpu().double() - ref).abs().max() / ref.abs().max()).item()
        assert err < 2e-6, (name, err)
    if mask is not None:
        assert (wd.grad.cpu()[~mask] == 0).all()


def test_wgrad_bias_skips_padding_rows(dev):
    """include/zuko_amd.h: rows[o] < 0 marks a padding column of g — neither dw nor db has a destination for it.  zk_wgrad_bias_f32 called
    directly with a row table holding -1 entries: rows of dw / entries of db of the real outputs equal g^T h / colsum(g) (zuko/nn.py:217-218
    under autograd), and the guard words placed BEFORE db and dw (where index -1 would land) are untouched."""
    from zuko_amd import _C
    from zuko_amd.ops import _ptr, _stream

    lib = _C.lib()
    gen = torch.Generator().manual_seed(21)
    N, width, in_f = 700, 128, 128
    rows_h = torch.full((width,), -1, dtype=torch.int32)
    real = torch.randperm(width, generator=gen)[:96]
    rows_h[real] = torch.randperm(96, generator=gen).to(torch.int32)  # 96 module rows, 32 padding slots
    g = torch.randn(N, width, generator=gen)
    g[:, rows_h < 0] = 0.0  # (padding columns of the packed gradient are always zero)
    h = torch.randn(N, in_f, generator=gen)
    pairs = torch.tensor([[0, 0]], dtype=torch.int32).to(dev)
    flag = torch.ones(1, dtype=torch.uint8, device=dev)
    ns = max(1, lib.zk_wgrad_slices(N, 1))
    partial = torch.empty(ns * 128 * 128, device=dev)
    cs_partial = torch.empty(ns * 128, device=dev)
    guard = 12345.0
    dwbuf = torch.full((in_f + 96 * in_f,), guard, device=dev)
    dbbuf = torch.full((8 + 96,), guard, device=dev)
    dw, db = dwbuf[in_f:].view(96, in_f), dbbuf[8:]
    dw.zero_(); db.zero_()
    gd, hd, rows = g.to(dev), h.to(dev), rows_h.to(dev)
    err = lib.zk_wgrad_bias_f32(N, width, in_f, _ptr(gd), gd.stride(0), _ptr(hd), hd.stride(0), _ptr(pairs), 1, _ptr(partial), None, _ptr(dw), 0,
                                _ptr(flag), _ptr(cs_partial), _ptr(db), _ptr(rows), None, _stream())
    _C.check(err, "zk_wgrad_bias_f32")
    torch.cuda.synchronize()
    assert (dwbuf[:in_f] == guard).all() and (dbbuf[:8] == guard).all()
    want_w = torch.zeros(96, in_f, dtype=torch.float64)
    want_b = torch.zeros(96, dtype=torch.float64)
    full = g.double().T @ h.double()
    for o in range(width):
        if rows_h[o] >= 0:
            want_w[rows_h[o]] = full[o]
            want_b[rows_h[o]] = g[:, o].double().sum()
    assert (dw.cpu().double() - want_w).abs().max().item() < 1e-4 * want_w.abs().max().item()
    assert (db.cpu().double() - want_b).abs().max().item() < 1e-4 * want_b.abs().max().item()
