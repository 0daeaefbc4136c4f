"""GPU parity of the spline BIN INDEX on the product arithmetic (zuko/transforms.py:521-526).

The bin of the reference is k = #(horizontal knots < x) - 1 on ITS knots.  The product kernels (stream / general
spline kernel, epilogue of the fused autoregressive kernel) build their knots with different roundings (v_exp / v_rcp
instead of Sleef; in the fused kernel also a different GEMM summation order), so what can and must hold is

  (1) EXACT on the kernel's own knots: the index the kernel used equals #(own knots < x) - 1, for every element —
      asserted on the knots the diagnostic twins (zk_rqs_diag / zk_ar_forward_diag) write out, which run the same
      template / same per-element function as the product kernels (their y and ladj are asserted bit-identical);
  (2) every disagreement with the oracle's index is EXPLAINED by the distance between the two knot sets: the two
      indices differ by one and x lies between the kernel's knot and the oracle's knot of the same number, i.e.
      |x - oracle_knot| <= |own_knot - oracle_knot|; that distance itself is bounded in ulps of the bound B = 5.
"""

import numpy as np
import pytest
import torch

from conftest import build_flow, oracle_spec
from oracle import zuko_oracle as O

pytestmark = pytest.mark.gpu

ULP_B = 2.0 ** -21  # spacing of float32 in [4, 8): one ulp of a knot next to the bound B = 5


def _check_bins(x, k_own, knots_own, k_ref, knots_ref, what, max_knot_ulps):
    """x [..., D]; k_* int64 [..., D]; knots_* [..., D, K+1] (CPU tensors)."""
    K = knots_own.shape[-1] - 1
    # (1) exact on own knots
    recount = (knots_own < x.unsqueeze(-1)).sum(-1) - 1
    assert torch.equal(recount, k_own), f"{what}: bin index != #(own knots < x) - 1 at {int((recount != k_own).sum())} elements"
    # (2) disagreements with the oracle are explained by knot distance
    dev = (knots_own.double() - knots_ref.double()).abs()
    finite = torch.isfinite(dev)
    dmax = float(dev[finite].max()) if finite.any() else 0.0
    flips = k_own != k_ref
    nflip = int(flips.sum())
    worst = 0.0
    if nflip:
        ko, kr = k_own[flips], k_ref[flips]
        assert int((ko - kr).abs().max()) == 1, f"{what}: a bin index is off by more than one"
        j = torch.maximum(ko, kr).clamp(0, K)  # the knot that separates the two bins
        own_j = knots_own[flips].gather(-1, j.unsqueeze(-1)).squeeze(-1).double()
        ref_j = knots_ref[flips].gather(-1, j.unsqueeze(-1)).squeeze(-1).double()
        xf = x[flips].double()
        gap = (xf - ref_j).abs()
        assert bool((gap <= (own_j - ref_j).abs()).all()), f"{what}: a flipped element does not lie between the two knots"
        worst = float(gap.max())
    print(f"{what}: {x.numel()} elements, {nflip} bin flips vs oracle (rate {nflip / max(1, x.numel()):.2e}), "
          f"max |own knot - oracle knot| = {dmax / ULP_B:.2f} ulp(B), max |x - knot| at a flip = {worst / ULP_B:.2f} ulp(B)")
    assert dmax <= max_knot_ulps * ULP_B, f"{what}: knots differ from the oracle's by {dmax / ULP_B:.1f} ulp(B) > {max_knot_ulps}"
    return nflip


@pytest.mark.parametrize("K", [8, 16, 4])
@pytest.mark.parametrize("inverse", [False, True])
def test_standalone_bin_index(dev, K, inverse):
    """Product spline arithmetic from packed unconstrained parameters (same phi as the oracle)."""
    from zuko_amd import ops
    from zuko_amd.utils import unpack

    gen = torch.Generator().manual_seed(100 + K + int(inverse))
    N, D = 4096, 64
    phi = torch.randn(N, D, 3 * K - 1, generator=gen)
    v = torch.randn(N, D, generator=gen) * 2.0
    w, h, d = unpack(phi, [(K,), (K,), (K - 1,)])
    hor, ver, der = O.rqs_knots(w, h, d)
    ref_knots = ver if inverse else hor
    # adversarial values: exactly ON the oracle's knots, one ulp either side, +-B, beyond, non-finite
    v[:8] = ref_knots[:8, :, 3]
    v[8:16] = torch.nextafter(ref_knots[8:16, :, 4], torch.tensor(10.0))
    v[16:24] = torch.nextafter(ref_knots[16:24, :, 2], torch.tensor(-10.0))
    v[24, :8] = torch.tensor([5.0, -5.0, 6.0, -6.0, float("nan"), float("inf"), float("-inf"), 0.0])
    k_ref = O.rqs_bin_index(ref_knots, v)
    phig, vg = phi.to(dev), v.to(dev)
    wg, hg, dg = unpack(phig, [(K,), (K,), (K - 1,)])
    with torch.no_grad():
        out, ladj, k, knots = ops.rqs_diag(vg, wg, hg, dg, inverse=inverse)
        # the diagnostic launch IS the product arithmetic: bit-identical to the stream kernel
        if inverse:
            prod = ops.rqs_inverse(vg, wg, hg, dg)
            assert torch.equal(prod.view(torch.int32), out.view(torch.int32))
        else:
            py, pl = ops.rqs_forward(vg, wg, hg, dg)
            assert torch.equal(py.view(torch.int32), out.view(torch.int32)) and torch.equal(pl.view(torch.int32), ladj.view(torch.int32))
        # second pass: values exactly on the kernel's OWN knots must fall in the bin on the left (strict <)
        v2 = knots[..., K // 2].contiguous()
        _, _, k2, knots2 = ops.rqs_diag(v2, wg, hg, dg, inverse=inverse)
    assert torch.equal(knots2, knots)
    assert torch.equal(k2.cpu().long(), torch.full((N, D), K // 2 - 1)), "x == own knot j must give k = j - 1"
    _check_bins(v, k.cpu().long(), knots.cpu(), k_ref, ref_knots, f"standalone K={K} {'inverse' if inverse else 'forward'}", max_knot_ulps=16)


@pytest.mark.parametrize("name,N", [("nsf_cfg2", 4096), ("nsf_ctx", 1000)])
def test_fused_bin_index(dev, name, N):
    """The fused conditioner + spline kernel (the headline path): per transform, on that transform's own input."""
    import zuko_amd.flows as F
    from zuko_amd.nn import MaskedLinear

    if name == "nsf_ctx":  # context + ragged batch + 16 bins
        entry = (F.NSF, dict(features=8, context=5, transforms=2, bins=16, hidden_features=[64, 64]), 11, "ar", O.uni_rqs(16), {})
        torch.manual_seed(entry[2])
        flow = entry[0](**entry[1])
    else:
        flow, entry = build_flow(name)
    spec = oracle_spec(flow, entry)
    feats = entry[1]["features"]
    ctx = entry[1].get("context", 0)
    K = entry[1]["bins"]
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(N, feats, generator=gen)
    c = torch.randn(N, ctx, generator=gen) if ctx else None
    flow = flow.to(dev)
    total_flips = 0
    cur = x
    with torch.no_grad():
        for li, (lazy, layer) in enumerate(zip(flow.transform.transforms, spec.layers)):
            st = lazy.fused_state(dev)
            assert st is not None, "layer is expected to run on the fused kernel"
            st.refresh([m for m in lazy.hyper if isinstance(m, MaskedLinear)])
            din = feats + ctx
            dinp = -(-din // 4) * 4
            inp = torch.zeros(N, dinp, device=dev)
            inp[:, :feats] = cur.to(dev)
            if c is not None:
                inp[:, feats:din] = c.to(dev)
            y, ladj = torch.empty(N, feats, device=dev), torch.empty(N, device=dev)
            yd, ld = torch.empty_like(y), torch.empty_like(ladj)
            bins = torch.empty(N, feats, dtype=torch.int32, device=dev)
            knots = torch.empty(N, feats, K + 1, device=dev)
            st.run(inp, y, ladj, False)
            st.run_diag(inp, yd, ld, bins, knots)
            assert torch.equal(y.view(torch.int32), yd.view(torch.int32)) and torch.equal(ladj.view(torch.int32), ld.view(torch.int32)), "diagnostic twin differs from the product launch"
            phi = O._ar_phi(layer, cur, c)
            w, h, d = O.split_packed(phi, layer.uni.shapes)
            hor, _, _ = O.rqs_knots(w, h, d)
            k_ref = O.rqs_bin_index(hor, cur)
            # GEMM summation order moves the parameters by ~1e-6 relative, hence the wider knot allowance
            total_flips += _check_bins(cur, bins.cpu().long(), knots.cpu(), k_ref, hor, f"{name} transform {li}", max_knot_ulps=64)
            cur, _ = O.ar_forward(layer, cur, c)
    print(f"{name}: {total_flips} flips over {len(spec.layers)} transforms x {N} x {feats} elements")
