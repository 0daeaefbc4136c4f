"""INTEGRATION.md option 2, executed: the HIP-backed univariate transforms handed to the REAL zuko factories
(`zuko.flows.MAF(..., univariate=zuko_amd.transforms.MonotonicRQSTransform)`): constructor, `repr` (zuko calls the
univariate constructor on CPU tensors there, zuko/flows/autoregressive.py:188), the lazy call `flow()` that builds the
Transform objects, and a state_dict round trip with the mirrored `zuko_amd.flows` classes.  No kernel is launched
(there is no GPU here); the reference is only present in the build container, so the test skips elsewhere."""

import os
import sys
from functools import partial

import pytest
import torch

REF = "/root/reference"


@pytest.fixture(scope="module")
def zuko():
    if not os.path.isdir(os.path.join(REF, "zuko")):
        pytest.skip("the reference checkout is not mounted")
    sys.path.insert(0, REF)
    try:
        import zuko as z

        yield z
    finally:
        sys.path.remove(REF)


def test_real_zuko_maf_with_hip_univariates(zuko):
    import zuko_amd.flows as ZF
    import zuko_amd.transforms as ZT

    torch.manual_seed(0)
    ref = zuko.flows.MAF(6, 2, transforms=2, hidden_features=[32, 32], univariate=partial(ZT.MonotonicRQSTransform, slope=1e-3), shapes=[(8,), (8,), (7,)])
    text = repr(ref)
    assert "MonotonicRQSTransform" in text and "MaskedMLP" in text
    # the lazy call builds zuko's AutoregressiveTransform around OUR univariate constructor (no evaluation yet)
    dist = ref(torch.randn(2))
    assert type(dist).__name__ == "NormalizingFlow"
    t = dist.transform.transforms[0]
    assert type(t).__module__.startswith("zuko.") and t.passes == 6
    # the univariate object zuko would call is ours and honours the Transform contract zuko relies on
    u = ZT.MonotonicRQSTransform(torch.randn(8), torch.randn(8), torch.randn(7))
    assert u.bijective and u.sign == +1 and hasattr(u, "call_and_ladj") and "MonotonicRQSTransform" in repr(u)
    assert u.inv.inv is u  # an inverse view, as torch.distributions.Transform.inv

    # same module tree / keys as the mirrored package: weights move both ways
    torch.manual_seed(0)
    mine = ZF.NSF(6, 2, transforms=2, bins=8, hidden_features=[32, 32])
    sd_ref, sd_mine = ref.state_dict(), mine.state_dict()
    assert list(sd_ref.keys()) == list(sd_mine.keys())
    for k in sd_ref:
        assert torch.equal(sd_ref[k], sd_mine[k]), k  # same seed -> same initial weights, bit for bit
    mine.load_state_dict(sd_ref)
    ref.load_state_dict(mine.state_dict())


def test_real_zuko_coupling_with_hip_affine(zuko):
    import zuko_amd.transforms as ZT

    t = zuko.flows.GeneralCouplingTransform(6, 0, univariate=ZT.MonotonicAffineTransform, shapes=[(), ()], hidden_features=[16])
    assert "MonotonicAffineTransform" in repr(t)
    assert type(t()).__name__ == "CouplingTransform"


@pytest.mark.parametrize("bounded", [False, True])
def test_oracle_bernstein_follows_the_reference_for_any_eps(zuko, bounded):
    """`eps` is a constructor kwarg of the reference's Bernstein transforms (MonotonicTransform, zuko/transforms.py:594): it moves the
    linear-continuation margin and sets the bisection depth.  The oracle's eps argument (what the GPU test of the kernels' run-time eps
    is checked against) must follow the live reference bit for bit, in float64, also away from the default."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import zuko_oracle as O

    g = torch.Generator().manual_seed(3)
    theta = torch.randn(40, 3, 17 if bounded else 16, generator=g, dtype=torch.float64)
    x = torch.randn(40, 3, generator=g, dtype=torch.float64) * 3.5
    x[0, 0], x[0, 1], x[1, 0] = 4.99, -4.995, 5.2  # inside the widened margins / beyond the bound
    cls = zuko.transforms.BoundedBernsteinTransform if bounded else zuko.transforms.BernsteinTransform
    for eps in (1e-6, 1e-3, 2e-2):
        t = cls(theta, eps=eps)
        y, ladj = t.call_and_ladj(x)
        oy, ol = O.bern_forward(theta, x, bounded, eps=eps)
        assert torch.equal(y, oy) and torch.equal(ladj, ol), eps
        assert torch.equal(t.inv(y), O.bern_inverse(theta, y, bounded, eps=eps)), eps
