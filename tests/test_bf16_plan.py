"""CPU: host logic of the bf16 conditioner path (zuko_amd/nn.py:_Bf16Plan) — the hidden-unit reordering is
an exact reparametrisation of the masked MLP and the live-tile map never marks a non-zero tile dead."""

import pytest
import torch


@pytest.mark.parametrize("D,total,hidden,ctx", [(64, 11, [128, 128, 128], 0), (128, 47, [256, 192], 0), (64, 2, [64], 64)])
def test_bf16_plan_is_an_exact_reparametrisation(D, total, hidden, ctx):
    from zuko_amd.nn import MaskedMLP, _Bf16Plan

    torch.manual_seed(D + total)
    order = torch.randperm(D)
    adjacency = (order[:, None] > order).repeat_interleave(total, dim=0)
    if ctx:
        adjacency = torch.cat((adjacency, torch.ones(D * total, ctx, dtype=bool)), dim=1)
    mlp = MaskedMLP(adjacency, hidden_features=hidden).double()
    lins = list(mlp)[0::2]
    plan = _Bf16Plan(lins)
    plan.refresh(lins)
    x = torch.randn(7, D + ctx, dtype=torch.float64)
    ref, h = x, x
    for i, l in enumerate(lins):
        ref = torch.nn.functional.linear(ref, l.weight * l.mask, l.bias)
        h = torch.nn.functional.linear(h, plan.weights[i], plan.biases[i])
        if i + 1 < len(lins):
            ref, h = ref.relu(), h.relu()
    assert torch.allclose(h, ref, rtol=0, atol=1e-12)
    for w, live in zip(plan.weights, plan.live):
        if live is None:
            continue
        out_f, in_f = w.shape
        nz = torch.nn.functional.pad(w != 0, (0, 0, 0, (-out_f) % 256)).reshape(-1, 256, in_f // 64, 64).any(dim=3).any(dim=1)
        bits = ((live.unsqueeze(-1) >> torch.arange(in_f // 64)) & 1).bool()
        assert live.dtype == torch.int64 and bits.shape == nz.shape
        assert not (nz & ~bits).any(), "a tile holding non-zero weights is marked dead"
    # a parameter update invalidates the cached masked weights
    with torch.no_grad():
        lins[0].weight.add_(1.0)
    plan.refresh(lins)
    assert torch.equal(plan.weights[0], ((lins[0].weight if plan.perms[0] is None else lins[0].weight[plan.perms[0]]) * plan.masks_p[0]))


def test_bf16_plan_skips_half_of_the_last_layer_at_cfg5_shape():
    """NSF(1024, K=16, H=[1024]^3): after degree sorting about 40 % of the 256 x 64 tiles of the 48128 x 1024
    last layer are dead (the mask itself is 50 % zeros)."""
    from zuko_amd.nn import _Bf16Plan, masked_mlp_masks

    D, total = 1024, 47
    order = torch.arange(D)
    adjacency = (order[:, None] > order).repeat_interleave(total, dim=0)
    masks = masked_mlp_masks(adjacency, [1024] * 3)

    class L:  # the plan only needs .mask at construction
        def __init__(self, m):
            self.mask = m

    plan = _Bf16Plan([L(m) for m in masks])
    frac = plan.live_fraction()
    assert 0.55 < frac[-1] < 0.65 and all(0.5 < f <= 0.7 for f in frac[:-1])


@pytest.mark.parametrize("D,K,hidden", [(64, 16, [128, 128]), (70, 8, [64])])
def test_spline_panels_regroup_the_last_layer(D, K, hidden):
    """zk_linear_bf16_rqs wants the last layer's rows in panels of 256 = the 3K-1 parameters of 256 // (3K-1) whole
    features: panel p, row (f % FP) * total + j == original row f * total + j; padding rows are zero and dead."""
    from zuko_amd.nn import MaskedMLP, _Bf16Plan

    torch.manual_seed(K)
    total = 3 * K - 1
    fp = 256 // total
    order = torch.arange(D)
    adjacency = (order[:, None] > order).repeat_interleave(total, dim=0)
    mlp = MaskedMLP(adjacency, hidden_features=hidden).double()
    lins = list(mlp)[0::2]
    plan = _Bf16Plan(lins)
    plan.refresh(lins)
    wp, bp, live = plan.spline_panels(lins, K, D)
    panels = -(-D // fp)
    assert wp.shape == (panels * 256, hidden[-1]) and bp.shape == (panels * 256,) and live.shape == (panels,)
    w, b = plan.weights[-1], plan.biases[-1]
    used = torch.zeros(panels * 256, dtype=bool)
    for f in range(D):
        r0 = (f // fp) * 256 + (f % fp) * total
        assert torch.equal(wp[r0 : r0 + total], w[f * total : (f + 1) * total]) and torch.equal(bp[r0 : r0 + total], b[f * total : (f + 1) * total])
        used[r0 : r0 + total] = True
    assert (wp[~used] == 0).all() and (bp[~used] == 0).all()
    nz = (wp != 0).reshape(panels, 256, -1, 64).any(dim=3).any(dim=1)
    bits = ((live.unsqueeze(-1) >> torch.arange(hidden[-1] // 64)) & 1).bool()
    assert not (nz & ~bits).any()
    assert plan.spline_panels(lins, K, D)[0] is wp  # cached per parameter version
