"""CPU: the schedule of the layer-wise inverse in wavefront form (zuko_amd/flows/autoregressive.py: wavefront_inverse — which hidden units and
which features each sweep evaluates, the row-gathered weight copies, scattered orders, contexts, sweeps of several features) walked with torch /
oracle stand-ins for the HIP kernels, against the reference's loop (zuko/transforms.py:994-1000: every unit, every row, every feature, every
sweep) on the same float64 parameters."""

import pytest
import torch
import torch.nn.functional as Fn

from oracle import zuko_oracle as O


def _flows():
    import zuko_amd.flows as F

    return {
        "nsf_ctx": (lambda: F.NSF(6, 2, transforms=2, hidden_features=[32, 24]), O.uni_rqs(8), 2),
        "maf_randperm_passes3": (lambda: F.MAF(12, 0, transforms=2, hidden_features=[40], randperm=True, passes=3), O.UNI_AFFINE, 0),
        "nsf_passes2_softplus": (lambda: F.NSF(8, 1, transforms=2, hidden_features=[24, 24], passes=2, activation=torch.nn.Softplus), O.uni_rqs(8), 1),
        "sospf": (lambda: F.SOSPF(5, 0, transforms=2, hidden_features=[24, 24]), O.uni_sos(), 0),
        "bpf_ctx": (lambda: F.BPF(4, 3, transforms=1, hidden_features=[16]), O.uni_bpf(), 3),
    }


@pytest.mark.parametrize("name", ["nsf_ctx", "maf_randperm_passes3", "nsf_passes2_softplus", "sospf", "bpf_ctx"])
def test_wavefront_schedule_reproduces_the_reference_loop(name):
    from zuko_amd.flows.autoregressive import MaskedAutoregressiveTransform, wavefront_inverse

    make, uni, C = _flows()[name]
    torch.manual_seed(4)
    flow = make().double()
    with torch.no_grad():
        for p in flow.parameters():
            p.mul_(1.5)
    N = 37

    def linear(h, w, b, m, act):
        out = Fn.linear(h, w * m, b)
        return out if act is None else act(out)

    for lazy in (t for t in flow.transform.transforms if isinstance(t, MaskedAutoregressiveTransform)):
        D = lazy.features
        g = torch.Generator().manual_seed(9)
        y = torch.randn(N, D, generator=g, dtype=torch.float64) * 0.7
        c = torch.randn(N, C, generator=g, dtype=torch.float64) if C else None
        mods = list(lazy.hyper)
        n_last, n_hidden = [0], [0]

        def counting_linear(h, w, b, m, act):
            (n_last if act is None else n_hidden)[0] += w.shape[0]
            return linear(h, w, b, m, act)

        inverse_of = lambda phi, ys: O.univariate_inverse(uni, phi, ys)
        with torch.no_grad():
            x_w = wavefront_inverse(lazy, y, c, lazy.passes, counting_linear, inverse_of)
            # the loop as the reference writes it, with the same stand-ins
            x_r = torch.zeros_like(y)
            for _ in range(lazy.passes):
                h = x_r if c is None else torch.cat((x_r, c), dim=-1)
                for i in range(0, len(mods) - 1, 2):
                    h = linear(h, mods[i].weight, mods[i].bias, mods[i].mask, mods[i + 1])
                phi = linear(h, mods[-1].weight, mods[-1].bias, mods[-1].mask, None).unflatten(-1, (D, lazy.total))
                x_r = inverse_of(phi, y)
        assert n_last[0] == D * lazy.total, "every feature's rows of the last layer exactly once"
        assert n_hidden[0] <= sum(m.weight.shape[0] for m in mods[0:-1:2]), "every hidden unit at most once (the loop: passes x all of them)"
        assert (x_w - x_r).abs().max().item() <= 1e-12 * max(1.0, x_r.abs().max().item()), name
        # and it inverts the forward map of the layer
        with torch.no_grad():
            h = x_w if c is None else torch.cat((x_w, c), dim=-1)
            for i in range(0, len(mods) - 1, 2):
                h = linear(h, mods[i].weight, mods[i].bias, mods[i].mask, mods[i + 1])
            phi = linear(h, mods[-1].weight, mods[-1].bias, mods[-1].mask, None).unflatten(-1, (D, lazy.total))
            y_back, _ = O.univariate_forward(uni, phi, x_w)
        assert (y_back - y).abs().max().item() < (1e-4 if name in ("sospf", "bpf_ctx") else 1e-9)


def test_every_hidden_unit_is_scheduled_at_most_once_and_before_its_readers():
    """_sweep_units: a unit's sweep is the first one at whose start all the inputs it is connected to are final; the row-gathered copies hold
    exactly those rows, in sweep order."""
    import zuko_amd.flows as F

    torch.manual_seed(2)
    for flow in (F.NSF(10, 3, transforms=1, hidden_features=[48, 40]), F.MAF(9, 0, transforms=1, hidden_features=[33], randperm=True, passes=4)):
        lazy = flow.transform.transforms[0]
        dev = torch.device("cpu")
        units = lazy._sweep_units(dev, lazy.passes)
        rows = lazy._sweep_unit_rows(dev, lazy.passes)
        mods = list(lazy.hyper)
        lins = mods[0::2]
        order = lazy.order
        ready = torch.cat((order + 1, torch.zeros(lins[0].mask.shape[1] - order.numel(), dtype=order.dtype)))
        for l, lin in enumerate(lins[:-1]):
            all_units = torch.cat([u for u in units[l] if u is not None]) if any(u is not None for u in units[l]) else torch.zeros(0, dtype=torch.long)
            assert all_units.unique().numel() == all_units.numel(), "a unit is scheduled once"
            nxt = torch.full((lin.mask.shape[0],), lazy.passes + 1, dtype=ready.dtype)
            for s_, u in enumerate(units[l]):
                if u is None:
                    continue
                for unit in u.tolist():
                    ins = lin.mask[unit].nonzero().squeeze(-1)
                    need = int(ready[ins].max()) if ins.numel() else 0
                    assert need == s_, "scheduled in the first sweep at whose start its inputs are final"
                    nxt[unit] = s_
            w, m, b, starts = rows[l]
            assert torch.equal(w, lin.weight[all_units]) and torch.equal(m, lin.mask[all_units]) and torch.equal(b, lin.bias[all_units])
            assert starts == [int(sum(0 if u is None else u.numel() for u in units[l][:t])) for t in range(lazy.passes + 1)]
            ready = nxt  # units never scheduled (they would become final only after the last sweep) must not be read by anything scheduled
        # the last layer's rows of a feature of order s read only units that are final by sweep s
        last = lins[-1]
        for f in range(lazy.features):
            ins = last.mask[f * lazy.total : (f + 1) * lazy.total].any(dim=0).nonzero().squeeze(-1)
            assert ins.numel() == 0 or int(ready[ins].max()) <= int(order[f])


def test_stateful_or_unknown_activations_do_not_take_the_per_unit_sweeps():
    """The per-sweep schedule applies an activation to a subset of a layer's units: only elementwise, state-free modules qualify.  nn.PReLU(H)
    (one slope per unit), a subclass of a known activation, or a module the list does not know must fall to the whole-layer form (round-5 advisor finding:
    MAF(6, hidden=[40], activation=lambda: nn.PReLU(40)) raised inside wavefront_inverse) — and there the schedule equals the reference's loop."""
    import torch.nn as nn

    import zuko_amd.flows as F
    from zuko_amd.flows.autoregressive import wavefront_inverse

    class MyTanh(nn.Tanh):
        pass

    torch.manual_seed(3)
    for act in (lambda: nn.PReLU(40), MyTanh, lambda: nn.Threshold(0.1, 0.0)):  # (unit-mixing modules — LayerNorm, Softmax — are not autoregressive conditioners at all)
        flow = F.MAF(6, 0, transforms=1, hidden_features=[40], activation=act).double()
        lazy = flow.transform.transforms[0]
        assert lazy._sweep_units(torch.device("cpu"), lazy.passes) is None
        assert lazy._sweep_unit_rows(torch.device("cpu"), lazy.passes) is None
        mods = list(lazy.hyper)
        y = torch.randn(11, 6, dtype=torch.float64)

        def linear(h, w, b, m, a):
            out = Fn.linear(h, w * m, b)
            return out if a is None else a(out)

        inverse_of = lambda phi, ys: O.univariate_inverse(O.UNI_AFFINE, phi, ys)
        with torch.no_grad():
            whole = lambda ms, h: linear(h, ms[0].weight, ms[0].bias, ms[0].mask, ms[1])
            x_w = wavefront_inverse(lazy, y, None, lazy.passes, linear, inverse_of, stack=whole)
            x_r = torch.zeros_like(y)
            for _ in range(lazy.passes):
                h = linear(x_r, mods[0].weight, mods[0].bias, mods[0].mask, mods[1])
                phi = linear(h, mods[2].weight, mods[2].bias, mods[2].mask, None).unflatten(-1, (6, lazy.total))
                x_r = inverse_of(phi, y)
        assert (x_w - x_r).abs().max().item() <= 1e-12
    for act in (nn.ReLU, nn.ELU, nn.Softplus, nn.SiLU):
        lazy = F.MAF(6, 0, transforms=1, hidden_features=[40], activation=act).transform.transforms[0]
        assert lazy._sweep_units(torch.device("cpu"), lazy.passes) is not None
