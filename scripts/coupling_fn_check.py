"""CouplingFn (zuko_amd/coupling_train.py) against the layer-wise autograd path (f32 MFMA GEMMs) and float64 autograd of plain torch ops (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zuko_amd.flows import RealNVP, NICE

dev = torch.device("cuda:0")


def grads(flow, x, c, off):
    os.environ["ZUKO_AMD_NO_COUPLING_FN"] = "1" if off else "0"
    flow.zero_grad(set_to_none=True)
    xg = x.detach().clone().requires_grad_()
    loss = -flow(c).log_prob(xg).mean()
    loss.backward()
    return loss.item(), [p.grad.detach().clone() for p in flow.parameters()], xg.grad.detach().clone()


def ref64(flow, x, c):
    """float64 autograd through plain torch ops of the same coupling layers (zuko/flows/coupling.py:128-136, transforms.py:436-446, 1037-1073)."""
    import math
    xs = x.double().detach().clone().requires_grad_()
    ps = [p.detach().double().requires_grad_() for p in flow.parameters()]
    it = iter(ps)
    z, ladj = xs, 0.0
    for t in flow.transform.transforms:
        lins = list(t.hyper)[0::2]
        idx_a, idx_b = t.mask.nonzero().squeeze(-1), (~t.mask).nonzero().squeeze(-1)
        a, b = z[:, idx_a], z[:, idx_b]
        h = a if c is None else torch.cat((a, c.double()), 1)
        for i, _ in enumerate(lins):
            w, bias = next(it), next(it)
            h = h @ w.t() + bias
            if i + 1 < len(lins):
                h = torch.relu(h)
        phi = h.unflatten(-1, (-1, 2))
        shift, scale = phi[..., 0], phi[..., 1]
        ls = scale / (1 + (scale / math.log(1e3)).abs())
        yb = b * ls.exp() + shift
        ladj = ladj + ls.sum(-1)
        out = torch.empty_like(z)
        out[:, idx_a], out[:, idx_b] = a, yb
        z = out
    lp = (-0.5 * z.pow(2) - 0.5 * math.log(2 * math.pi)).sum(-1) + ladj
    loss = -lp.mean()
    loss.backward()
    return loss.item(), [p.grad for p in ps], xs.grad


ok = True
for name, mk, N, C in (("RealNVP(16, T=3, [64, 64])", lambda: RealNVP(16, 0, transforms=3, hidden_features=[64, 64]), 1000, 0),
                       ("NICE(12, ctx 4, T=2, [128] x 3)", lambda: NICE(12, 4, transforms=2, hidden_features=[128] * 3), 4096, 4),
                       ("RealNVP(256, T=4, [512] x 3)", lambda: RealNVP(256, 0, transforms=4, hidden_features=[512] * 3), 4096, 0)):
    torch.manual_seed(0)
    flow = mk().to(dev)
    x = torch.randn(N, flow.transform.transforms[0].mask.numel(), device=dev)
    c = torch.randn(N, C, device=dev) if C else None
    l1, g1, gx1 = grads(flow, x, c, False)
    l0, g0, gx0 = grads(flow, x, c, True)
    l64, g64, gx64 = ref64(flow, x, c)
    rel = lambda a, b: ((a.double() - b.double()).abs().sum() / b.double().abs().sum().clamp_min(1e-300)).item()
    mx = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300)).item()
    e_new = max(rel(a, b) for a, b in zip(g1, g64)); e_old = max(rel(a, b) for a, b in zip(g0, g64))
    m_new = max(mx(a, b) for a, b in zip(g1, g64)); m_old = max(mx(a, b) for a, b in zip(g0, g64))
    print(f"{name} rows {N}: loss one-node {l1:.7f} layer-wise {l0:.7f} float64 {l64:.7f}")
    print(f"    parameter gradients vs float64 autograd: 1-norm rel one-node {e_new:.2e} layer-wise {e_old:.2e}; max-norm rel one-node {m_new:.2e} layer-wise {m_old:.2e}")
    print(f"    input gradient vs float64: 1-norm rel one-node {rel(gx1, gx64):.2e} layer-wise {rel(gx0, gx64):.2e}")
    ok = ok and abs(l1 - l64) < 1e-5 * max(1, abs(l64)) and e_new < 2e-3 and m_new < 5e-2 and rel(gx1, gx64) < 2e-3
print("ok" if ok else "FAILED")
assert ok
