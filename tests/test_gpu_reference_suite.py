"""GPU: the reference's own test strategy (tests/test_flows.py, tests/test_nn.py,
tests/test_transforms.py — mathematical self-consistency) re-stated for zuko_amd in float32 on the device.
Gradients (log_prob, rsample, SOS / Bernstein adjoints) have their own value checks against autograd through the oracle in
tests/test_gpu_backward.py; here ladj is checked against a finite-difference Jacobian as the reference's tests do."""

from functools import partial

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _fd_jacobian(f, x, eps=1e-3):
    """Central finite differences of a vector map R^D -> R^D (float64 accumulation on the host)."""
    D = x.numel()
    cols = []
    for i in range(D):
        e = torch.zeros_like(x)
        e[i] = eps
        cols.append(((f(x + e) - f(x - e)) / (2 * eps)).double().cpu())
    return torch.stack(cols, dim=-1)


@pytest.mark.parametrize("name", ["NICE", "MAF", "NSF", "NCSF", "SOSPF", "BPF"])
def test_flows(dev, name, tmp_path):
    import zuko_amd.flows as F

    torch.manual_seed(0)
    flow = getattr(F, name)(3, 5).to(dev)
    trainable = name in ("NICE", "MAF", "NSF", "NCSF")
    x, c = torch.randn(256, 3, device=dev), torch.randn(5, device=dev)
    if name == "NCSF":
        x = x.clamp(-3, 3)

    if trainable:  # log_prob with gradients (tests/test_flows.py:17-29)
        log_p = flow(c).log_prob(x)
        assert log_p.shape == (256,) and log_p.requires_grad
        flow.zero_grad(set_to_none=True)
        (-log_p.mean()).backward()
        for pname, p in flow.named_parameters():
            assert p.grad is not None, pname
    flow.eval()
    with torch.no_grad():
        log_p_eval = flow(c).log_prob(x)
        assert log_p_eval.shape == (256,)
        if trainable:  # eval (fused inference kernel) == train (layer-wise kernels with autograd)
            assert torch.allclose(log_p, log_p_eval, rtol=1e-5, atol=1e-4)
    flow.train()

    with torch.no_grad():
        s = flow(c).sample((32,))  # sampling (:41-43)
        assert s.shape == (32, 3)
        x2, c2 = torch.randn(256, 3, device=dev), torch.randn(256, 5, device=dev)  # invertibility (:57-61)
        if name == "NCSF":
            x2 = x2.clamp(-3, 3)
        t = flow(c2).transform
        z = t.inv(t(x2))
        assert torch.allclose(x2, z, atol=1e-4 if name not in ("SOSPF", "BPF") else 2e-3)

        # Jacobian: ladj == log|det J| (:64-75), J by finite differences of the HIP forward
        x1, c1 = torch.randn(3, device=dev) * 0.7, torch.randn(5, device=dev)
        t = flow(c1).transform
        J = _fd_jacobian(lambda v: t(v), x1)
        ladj = torch.linalg.slogdet(J).logabsdet
        assert abs(t.log_abs_det_jacobian(x1, t(x1)).item() - ladj.item()) < 5e-3
        Jinv = _fd_jacobian(lambda v: t.inv(v), t(x1))
        assert torch.allclose(J, torch.linalg.inv(Jinv), atol=2e-2)

    torch.save(flow, tmp_path / "flow.pth")  # whole-module pickle (:78-91)
    flow_bis = torch.load(tmp_path / "flow.pth", weights_only=False)
    with torch.no_grad():
        assert torch.allclose(flow(c).log_prob(x[:7]), flow_bis(c).log_prob(x[:7]))
    assert repr(flow)


def test_triangular_transforms(dev):
    """tests/test_flows.py:97-144: element-wise, coupling and autoregressive layers with passes, custom
    order, custom adjacency — round trip, autograd Jacobian, ladj == slogdet == sum log|diag|."""
    from zuko_amd.flows import ElementWiseTransform, GeneralCouplingTransform, MaskedAutoregressiveTransform

    torch.manual_seed(1)
    order = torch.randperm(5)
    adjacency = torch.rand((5, 5)) < 0.25
    adjacency = adjacency + torch.eye(5, dtype=bool)
    adjacency = torch.tril(adjacency)
    adjacency[1, 0] = True
    adjacency = adjacency[order, :][:, order]
    Ts = [
        ElementWiseTransform,
        GeneralCouplingTransform,
        partial(GeneralCouplingTransform, mask=order % 2),
        MaskedAutoregressiveTransform,
        partial(MaskedAutoregressiveTransform, passes=2),
        partial(MaskedAutoregressiveTransform, order=order),
        partial(MaskedAutoregressiveTransform, adjacency=adjacency),
    ]
    for T in Ts:
        t = T(5).to(dev)
        x = torch.randn(64, 5, device=dev)
        y = t()(x)
        assert y.shape == x.shape and y.requires_grad, T
        with torch.no_grad():
            assert torch.allclose(t().inv(y), x, atol=1e-4), T
        t = T(5, 7).to(dev)
        x, c = torch.randn(64, 5, device=dev), torch.randn(7, device=dev)
        y = t(c)(x)
        assert y.shape == x.shape and y.requires_grad, T
        with torch.no_grad():
            assert torch.allclose(t(c).inv(y), x, atol=1e-4), T
        t = T(5).to(dev)
        x = torch.randn(5, device=dev)
        y = t()(x)
        J = torch.autograd.functional.jacobian(t(), x)
        ladj = torch.linalg.slogdet(J.double()).logabsdet.float()
        assert torch.allclose(t().log_abs_det_jacobian(x, y), ladj, atol=1e-4), T
        assert torch.allclose(J.diag().abs().log().sum(), ladj, atol=1e-4), T


def test_adjacency_matrix_sparsity(dev):
    """tests/test_flows.py:147-218: the Jacobian vanishes exactly outside the requested adjacency,
    with and without context columns."""
    from zuko_amd.flows import MaskedAutoregressiveTransform as T

    torch.manual_seed(2)
    order = torch.randperm(5)
    adjacency = torch.rand((5, 5)) < 0.25
    adjacency = adjacency + torch.eye(5, dtype=bool)
    adjacency = torch.tril(adjacency)
    adjacency[1, 0] = True
    adjacency = adjacency[order, :][:, order]
    t = T(5, adjacency=adjacency).to(dev)
    x = torch.randn(5, device=dev)
    J = torch.autograd.functional.jacobian(t(), x)
    assert (J[~adjacency.to(dev)] == 0).all()

    ctx = torch.rand((5, 2)) < 0.25
    t = T(features=5, context=2, adjacency=torch.cat((adjacency, ctx), dim=1)).to(dev)
    x, c = torch.randn(5, device=dev), torch.randn(2, device=dev)
    y = t(c)(x)
    with torch.no_grad():
        assert torch.allclose(t(c).inv(y), x, atol=1e-4)
    J = torch.autograd.functional.jacobian(t(c), x)
    assert (J[~adjacency.to(dev)] == 0).all()
    ladj = torch.linalg.slogdet(J.double()).logabsdet.float()
    assert torch.allclose(t(c).log_abs_det_jacobian(x, y), ladj, atol=1e-4)
    with pytest.raises(AssertionError, match="'adjacency' should have 5 or 7 columns."):
        T(features=5, context=2, adjacency=torch.cat((adjacency, ctx[:, :1]), dim=1))


@pytest.mark.parametrize("residual", [True, False])
@pytest.mark.parametrize("batch", [(), (64,)])
def test_masked_mlp_jacobian(dev, residual, batch):
    """tests/test_nn.py:41-60: output shape, grad, zero Jacobian where ~adjacency and across batch items."""
    import math

    from zuko_amd.nn import MaskedMLP

    torch.manual_seed(3)
    adjacency = torch.randn(5, 3) < 0
    adjacency[0, 0] = True
    net = MaskedMLP(adjacency, activation=nn.ELU, residual=residual).to(dev)
    x = torch.randn(*batch, 3, device=dev)
    y = net(x)
    assert y.shape == (*batch, 5) and y.requires_grad
    J = torch.autograd.functional.jacobian(net, x)
    J = J.movedim(len(batch), -2)
    mask = torch.eye(math.prod(batch), dtype=bool).reshape(batch + batch).to(dev)
    assert (J[mask][..., ~adjacency.to(dev)] == 0).all()
    assert (J[~mask] == 0).all()


@pytest.mark.parametrize("batched", [False, True])
def test_univariate_transforms(dev, batched):
    """tests/test_transforms.py:12-74 for the transforms of the hot path: shape, inverse round trip,
    diagonal autograd Jacobian (affine, RQS), ladj == log|diag J| (finite differences for SOS / Bernstein)."""
    import zuko_amd.transforms as ZT

    torch.manual_seed(4)
    b = (256,) if batched else ()
    r = lambda *s: torch.randn(s, device=dev)
    ts = [
        ZT.MonotonicAffineTransform(r(*b), r(*b)),
        ZT.MonotonicRQSTransform(r(*b, 8), r(*b, 8), r(*b, 7)),
        ZT.BernsteinTransform(r(*b, 16)),
        ZT.BoundedBernsteinTransform(r(*b, 16)),
        ZT.SOSPolynomialTransform(r(*b, 3, 5)),
    ]
    x = torch.linspace(-5.0, 5.0, 256, device=dev)
    for t in ts:
        with torch.no_grad():
            y = t(x)
            assert y.shape == x.shape, t
            z = t.inv(y)
            assert torch.allclose(x, z, atol=1e-4 if not isinstance(t, (ZT.SOSPolynomialTransform, ZT.BernsteinTransform)) else 2e-3), t
            yc, lc = t.call_and_ladj(x)
            assert torch.allclose(yc, y, atol=1e-4), t
            eps = 1e-3
            fd = ((t(x + eps) - t(x - eps)) / (2 * eps)).abs().log()
            ok = (x.abs() < 4.9) if isinstance(t, (ZT.MonotonicRQSTransform, ZT.BernsteinTransform)) else torch.ones_like(x, dtype=bool)
            assert torch.allclose(lc[ok], fd[ok], atol=2e-2), t
        if isinstance(t, (ZT.MonotonicAffineTransform, ZT.MonotonicRQSTransform)):
            J = torch.autograd.functional.jacobian(t, x)
            assert (torch.triu(J, diagonal=+1) == 0).all() and (torch.tril(J, diagonal=-1) == 0).all(), t
            assert torch.allclose(t.log_abs_det_jacobian(x, y), torch.diag(J).abs().log(), atol=1e-4), t


def test_lazy_inverse_and_unconditional(dev):
    """zuko/lazy.py:74-98, 242-330: `LazyTransform.inv` swaps the directions of the transforms it builds;
    `UnconditionalTransform` / `UnconditionalDistribution` wrap parameter-free constructors."""
    from zuko_amd.flows import MaskedAutoregressiveTransform
    from zuko_amd.lazy import Flow, LazyInverse, UnconditionalDistribution, UnconditionalTransform
    from zuko_amd.distributions import DiagNormal
    from zuko_amd.transforms import SoftclipTransform

    torch.manual_seed(6)
    t = MaskedAutoregressiveTransform(5, 3).to(dev)
    ti = t.inv
    assert isinstance(ti, LazyInverse) and ti.inv is t
    x, c = torch.randn(32, 5, device=dev), torch.randn(32, 3, device=dev)
    with torch.no_grad():
        y = t(c)(x)
        assert torch.allclose(ti(c)(y), x, atol=1e-4)
        assert torch.allclose(ti(c).inv(x), y, atol=1e-5)
        # a flow whose generative direction is the conditioner's forward pass (fast sampling, slow density)
        flow = Flow(ti, UnconditionalDistribution(DiagNormal, torch.zeros(5, device=dev), torch.ones(5, device=dev), buffer=True)).to(dev)
        s = flow(c).sample()
        assert s.shape == (32, 5) and torch.isfinite(flow(c).log_prob(s)).all()
        u = UnconditionalTransform(SoftclipTransform, bound=6.0)
        assert torch.allclose(u().inv(u()(x)), x, atol=1e-4)
