"""GPU: the dense training GEMM on two-part f16 operands (csrc/gemm_half.hip, C ABI zk_amax_f32 / zk_wsplit_f16 / zk_gemm_f16x2) and the
one-node training path of a coupling transform built on it (zuko_amd/coupling_train.py), against float64 products and against autograd
through the CPU oracle — which is how the reference obtains its gradients (tests/test_flows.py:22-29, zuko/flows/coupling.py:128-136)."""

import os

import pytest
import torch

from oracle import zuko_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize(
    "M,K,N,act,gate,sa,sw",
    [
        (4096, 512, 512, 1, False, 1.0, 1.0),  # 256-wide tiles
        (4096, 512, 512, 0, True, 1.0, 1.0),
        (16384, 512, 256, 0, False, 1.0, 1.0),  # 128-wide tiles, 4 wavefronts
        (16384, 512, 128, 0, True, 1.0, 1.0),  # 64-row tiles
        (1000, 72, 40, 1, False, 1.0, 1.0),  # ragged in every direction (K % 32, N % 16, M % 64)
        (129, 8, 4, 1, False, 1.0, 1.0),
        (1, 512, 512, 1, False, 1.0, 1.0),
        (300, 200, 132, 0, True, 1.0, 1.0),
        (2048, 512, 512, 1, False, 1e-6, 1e3),  # operand magnitudes far from 1: the per-tensor powers of two absorb them
        (2048, 512, 512, 0, True, 1e5, 1e-4),
    ],
)
def test_two_part_gemm_against_float64(dev, M, K, N, act, gate, sa, sw):
    """c = act(a W^T + b) (* (gate > 0)): every output within 1e-6 of max |c| of the float64 product (the f32 matrix instruction it replaces:
    5e-7 .. 9e-7 on the same inputs, profiles/r06/gemm_half_check.txt), and the maximum it leaves on the device equals max |c| bit for bit."""
    from zuko_amd import coupling_train as ct

    g = torch.Generator().manual_seed(M + K + N)
    a = (torch.randn(M, K, generator=g) * sa).to(dev)
    a = a.clamp_min(0) if act else a
    w = (torch.randn(N, K, generator=g) * sw / K**0.5).to(dev)
    b = (torch.randn(N, generator=g) * sa * sw).to(dev)
    gt = torch.randn(M, N, generator=g).to(dev) if gate else None
    am = torch.zeros(3, ct.AMAX_WORDS, dtype=torch.int32, device=dev)
    ct.amax([(a, am[0]), (w, am[1])])
    assert am[0].max().item() == torch.tensor([a.abs().max().item()]).view(torch.int32).item()
    img = torch.empty(ct.image_words(N, K), dtype=torch.int32, device=dev)
    ct.wsplit([(w, False, am[1], img)])
    c = ct.gemm(a, am[0], img, am[1], N, b, act, gt, am[2])
    ref = a.double() @ w.double().t() + b.double()
    ref = ref.clamp_min(0) if act else ref
    ref = ref * (gt > 0) if gate else ref
    err = ((c.double() - ref).abs().max() / ref.abs().max()).item()
    assert err < 1e-6, err
    assert am[2].max().item() == torch.tensor([c.abs().max().item()]).view(torch.int32).item()
    # the dgrad operand: the same weight read transposed
    imgt = torch.empty(ct.image_words(K, N), dtype=torch.int32, device=dev)
    ct.wsplit([(w, True, am[1], imgt)])
    gy = torch.randn(M, N, generator=g).to(dev)
    am2 = torch.zeros(1, ct.AMAX_WORDS, dtype=torch.int32, device=dev)
    ct.amax([(gy, am2[0])])
    if N % 4 == 0:
        gx = ct.gemm(gy, am2[0], imgt, am[1], K, None, 0, None, None)
        refx = gy.double() @ w.double()
        assert ((gx.double() - refx).abs().max() / refx.abs().max()).item() < 1e-6


def _oracle_grads(flow, features, x, c):
    sd = {k: v.detach().cpu().clone() for k, v in flow.state_dict().items() if v is not None}
    leaves = {k: v.double().requires_grad_() for k, v in sd.items() if v.is_floating_point() and ("weight" in k or "bias" in k)}
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    sd.update(leaves)
    spec = O.spec_from_state_dict(sd, "coupling", O.UNI_AFFINE, features)
    xr = x.double().clone().requires_grad_()
    cr = None if c is None else c.double().clone().requires_grad_()
    loss = -O.flow_log_prob(spec, xr, cr).mean()
    loss.backward()
    return loss.detach(), {k: v.grad for k, v in leaves.items()}, xr.grad, None if cr is None else cr.grad


@pytest.mark.parametrize(
    "ctor,kw,rows",
    [
        ("RealNVP", dict(features=16, context=0, transforms=3, hidden_features=[64, 64]), 1000),
        ("NICE", dict(features=16, context=4, transforms=2, hidden_features=[128] * 3), 4096),
        ("RealNVP", dict(features=256, context=0, transforms=16, hidden_features=[512] * 3), 96),  # BASELINE config 4
        ("RealNVP", dict(features=256, context=0, transforms=4, hidden_features=[512] * 3), 4096),
    ],
)
def test_coupling_transform_trains_as_one_autograd_node(dev, ctor, kw, rows):
    """d(-log_prob.mean()) / d(parameters, x, context) of a coupling flow whose every transform is ONE autograd node (CouplingFn) against float64
    autograd through the oracle.  Up to 1 000 rows: 2e-4 of max |grad| per tensor (the bar of test_gradients_match_reference_autograd); at 4 096 rows
    the 1-norm bar of test_gradients_over_many_tiles (2e-3: ReLU units within float32 rounding of zero) plus the max-norm gate 5e-2; the layer-wise path's
    own distance on the same rows (f32 matrix instruction) is printed beside it — which units flip is a coin toss per path (profiles/r06/cfg4_grad_noise.txt)."""
    from zuko_amd import flows as F

    torch.manual_seed(7)
    flow = getattr(F, ctor)(**kw)
    D, C = kw["features"], kw["context"]
    g = torch.Generator().manual_seed(8)
    x = torch.randn(rows, D, generator=g)
    c = torch.randn(rows, C, generator=g) if C else None
    ref_loss, ref_grads, ref_gx, ref_gc = _oracle_grads(flow, D, x, c)
    flow = flow.to(dev)

    def run(off):
        os.environ["ZUKO_AMD_NO_COUPLING_FN"] = "1" if off else "0"
        try:
            flow.zero_grad(set_to_none=True)
            xg = x.to(dev).requires_grad_()
            cg = None if c is None else c.to(dev).requires_grad_()
            loss = -flow(cg).log_prob(xg).mean()
            names = set()

            def walk(fn):
                if fn is not None and fn not in names:
                    names.add(fn)
                    for nxt, _ in fn.next_functions:
                        walk(nxt)

            walk(loss.grad_fn)
            loss.backward()
            return loss.item(), {type(f).__name__ for f in names}, {k: p.grad.detach().cpu().double() for k, p in flow.named_parameters()}, xg.grad.cpu().double(), None if cg is None else cg.grad.cpu().double()
        finally:
            os.environ.pop("ZUKO_AMD_NO_COUPLING_FN", None)

    loss, names, grads, gx, gc = run(False)
    assert "CouplingFnBackward" in names, "the one-node path did not serve this flow"
    assert abs(loss - ref_loss.item()) < 1e-5 * max(1.0, abs(ref_loss.item()))
    _, names0, grads0, gx0, _ = run(True)
    assert "CouplingFnBackward" not in names0
    mx = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()
    l1 = lambda a, b: ((a - b).abs().sum() / b.abs().sum().clamp_min(1e-12)).item()
    worst_mx = max(mx(grads[k], ref_grads[k]) for k in ref_grads)
    worst_l1 = max(l1(grads[k], ref_grads[k]) for k in ref_grads)
    old_l1 = max(l1(grads0[k], ref_grads[k]) for k in ref_grads)
    if rows <= 1000:
        assert worst_mx < 2e-4, worst_mx
        assert mx(gx, ref_gx) < 2e-4
    else:
        assert worst_l1 < 2e-3 and worst_mx < 5e-2, (worst_l1, worst_mx)
        assert l1(gx, ref_gx) < 1e-5
    if gc is not None:
        assert l1(gc, ref_gc) < 1e-5
    print(f"{ctor}{kw} rows {rows}: parameter gradients vs float64 oracle autograd: max-norm {worst_mx:.2e}, 1-norm {worst_l1:.2e} (layer-wise path {old_l1:.2e}); dx 1-norm {l1(gx, ref_gx):.2e}")


def test_coupling_node_falls_back_when_not_covered(dev):
    """Widths that are not multiples of 4, a frozen parameter, an activation outside the epilogue's set: the layer-wise autograd path serves the call."""
    from zuko_amd import flows as F

    torch.manual_seed(0)
    x = torch.randn(64, 5, device=dev)
    c = torch.randn(64, 3, device=dev)
    flow = F.NICE(5, 3, transforms=2, hidden_features=[32, 32]).to(dev)  # 2 or 3 kept features + 3 context columns: not a multiple of 4
    loss = -flow(c).log_prob(x).mean()
    loss.backward()
    assert all(p.grad is not None for p in flow.parameters())
    flow2 = F.RealNVP(16, 0, transforms=2, hidden_features=[64, 64], activation=torch.nn.GELU).to(dev)  # (its derivative is not a function of its output)
    x2 = torch.randn(64, 16, device=dev)
    (-flow2().log_prob(x2).mean()).backward()
    assert all(p.grad is not None for p in flow2.parameters())
    flow3 = F.RealNVP(16, 0, transforms=2, hidden_features=[64, 64]).to(dev)
    next(flow3.parameters()).requires_grad_(False)
    (-flow3().log_prob(x2).mean()).backward()
    assert sum(p.grad is not None for p in flow3.parameters()) == len(list(flow3.parameters())) - 1


@pytest.mark.parametrize("act", ["ELU", "Tanh", "Sigmoid", "LeakyReLU"])
def test_coupling_node_with_other_activations(dev, act):
    """ELU / Tanh / Sigmoid / LeakyReLU conditioners (activation and its derivative in the GEMM epilogues) against float64 autograd through the same formulas in
    plain torch ops (zuko/flows/coupling.py:128-136, zuko/transforms.py:436-446, 1037-1073 — the oracle's conditioner is ReLU-only).  The smooth ones have no kinks:
    2e-5 of max |grad| per tensor at 2 048 rows (the bar of test_gradients_smooth_activation_max_norm)."""
    import math

    from zuko_amd import flows as F

    torch.manual_seed(11)
    flow = F.RealNVP(16, 4, transforms=3, hidden_features=[64, 128], activation=getattr(torch.nn, act)).to(dev)
    g = torch.Generator().manual_seed(12)
    x, c = torch.randn(2048, 16, generator=g), torch.randn(2048, 4, generator=g)
    xg, cg = x.to(dev).requires_grad_(), c.to(dev).requires_grad_()
    loss = -flow(cg).log_prob(xg).mean()
    names = set()

    def walk(fn):
        if fn is not None and fn not in names:
            names.add(fn)
            for nxt, _ in fn.next_functions:
                walk(nxt)

    walk(loss.grad_fn)
    assert "CouplingFnBackward" in {type(f).__name__ for f in names}
    loss.backward()
    # float64 reference
    xs, cs = x.double().requires_grad_(), c.double().requires_grad_()
    ps = [p.detach().cpu().double().requires_grad_() for p in flow.parameters()]
    it = iter(ps)
    fn64 = {"ELU": torch.nn.functional.elu, "Tanh": torch.tanh, "Sigmoid": torch.sigmoid, "LeakyReLU": torch.nn.functional.leaky_relu}[act]
    z, ladj = xs, 0.0
    for t in flow.transform.transforms:
        mask = t.mask.cpu()
        ia, ib = mask.nonzero().squeeze(-1), (~mask).nonzero().squeeze(-1)
        a, b = z[:, ia], z[:, ib]
        h = torch.cat((a, cs), 1)
        n_lin = len(list(t.hyper)[0::2])
        for i in range(n_lin):
            w, bias = next(it), next(it)
            h = h @ w.t() + bias
            if i + 1 < n_lin:
                h = fn64(h)
        phi = h.unflatten(-1, (-1, 2))
        shift, scale = phi[..., 0], phi[..., 1]
        ls = scale / (1 + (scale / math.log(1e3)).abs())
        out = torch.empty_like(z)
        out[:, ia], out[:, ib] = a, b * ls.exp() + shift
        z, ladj = out, ladj + ls.sum(-1)
    ref = -((-0.5 * z.pow(2) - 0.5 * math.log(2 * math.pi)).sum(-1) + ladj).mean()
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1.0, abs(ref.item()))
    mx = lambda u, v: ((u.detach().cpu().double() - v).abs().max() / v.abs().max().clamp_min(1e-12)).item()
    worst = max(mx(p.grad, q.grad) for p, q in zip(flow.parameters(), ps))
    bar = 2e-5 if act != "LeakyReLU" else 2e-4  # (LeakyReLU has the kink)
    assert worst < bar, worst
    assert mx(xg.grad, xs.grad) < bar and mx(cg.grad, cs.grad) < bar
    print(f"RealNVP(16, context 4) with {act}: parameter gradients vs float64 autograd, max-norm {worst:.2e}; dx {mx(xg.grad, xs.grad):.2e}, dc {mx(cg.grad, cs.grad):.2e}")


def test_coupling_node_one_launch_equals_the_sum_of_row_chunks(dev):
    """BASELINE config 4's conditioner shape at the bench's training batch (2^14 rows): the gradients of ONE call against the float64 sum of the gradients of its 16
    chunks of 1 024 rows (the loss is a sum over rows; the chunks see other per-tensor maxima, hence other operand scales, other tiles, other reduction slices) —
    the size-independent property that holds the many-tile GEMMs and the cross-slice weight-gradient reduction of the one-node path: 1e-5 of max |grad| per tensor."""
    from zuko_amd import flows as F

    torch.manual_seed(21)
    flow = F.RealNVP(256, 0, transforms=2, hidden_features=[512] * 3).to(dev)
    N, CH = 1 << 14, 1 << 10
    x = torch.randn(N, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(22))

    def grads(rows):
        flow.zero_grad(set_to_none=True)
        xg = rows.clone().requires_grad_()
        (-flow().log_prob(xg).sum() / N).backward()
        return [p.grad.detach().double() for p in flow.parameters()], xg.grad.detach().double()

    whole, gx_whole = grads(x)
    acc, gx_parts = None, []
    for i in range(0, N, CH):
        g, gx = grads(x[i : i + CH])
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
        gx_parts.append(gx)
    worst = max(((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item() for a, b in zip(whole, acc))
    gx_err = ((gx_whole - torch.cat(gx_parts)).abs().max() / gx_whole.abs().max()).item()
    assert worst < 1e-5, worst
    assert gx_err < 1e-5, gx_err
    print(f"RealNVP(256, [512] x 3) 2^14 rows in one call vs 16 chunks: parameter gradients {worst:.2e}, dx {gx_err:.2e} of max |grad|")


def test_two_part_gemm_at_2_to_the_20_rows(dev):
    """Size-independent properties at the headline batch (2^20 rows x 512 x 512, 8 192 workgroups): rows are independent — a permutation of the rows permutes the
    outputs bit for bit (same per-tensor maximum, hence the same scale) — and three slices of 4 096 rows agree with the float64 product."""
    from zuko_amd import coupling_train as ct

    M, K, N = 1 << 20, 512, 512
    g = torch.Generator(device=dev).manual_seed(5)
    a = torch.randn(M, K, device=dev, generator=g).clamp_min(0)
    w = torch.randn(N, K, device=dev, generator=g) / K**0.5
    b = torch.randn(N, device=dev, generator=g)
    am = torch.zeros(3, ct.AMAX_WORDS, dtype=torch.int32, device=dev)
    ct.amax([(a, am[0]), (w, am[1])])
    img = torch.empty(ct.image_words(N, K), dtype=torch.int32, device=dev)
    ct.wsplit([(w, False, am[1], img)])
    c = ct.gemm(a, am[0], img, am[1], N, b, 1, None, am[2])
    perm = torch.randperm(M, device=dev, generator=g)
    cp = ct.gemm(a[perm].contiguous(), am[0], img, am[1], N, b, 1, None, None)
    assert torch.equal(cp, c[perm])
    for lo in (0, M // 2 - 2048, M - 4096):
        ref = (a[lo : lo + 4096].double() @ w.double().t() + b.double()).clamp_min(0)
        assert ((c[lo : lo + 4096].double() - ref).abs().max() / ref.abs().max()).item() < 1e-6
    assert am[2].max().item() == torch.tensor([c.abs().max().item()]).view(torch.int32).item()
