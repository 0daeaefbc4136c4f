"""The hand-written reverse sweep of the rational-quadratic bin map (csrc/zk_univariate_bwd.h: rqs_local_vjp) restated line by line in torch float64 and held
to torch.autograd on the same forward expressions (zuko/transforms.py:554-567) — the derivation the fused backward kernel rests on."""

import torch


def _forward(x, x0, x1, y0, y1, d0, d1):
    w, h = x1 - x0, y1 - y0
    s, z = h / w, (x - x0) / w
    omz = 1 - z
    zz = z * omz
    den = s + (d0 + d1 - 2 * s) * zz
    num = s * z * z + d0 * zz
    y = y0 + h * num / den
    jac = s * s * (2 * s * zz + d0 * omz * omz + d1 * z * z) / (den * den)
    return y, torch.log(jac)


def _rqs_local_vjp(x, x0, x1, y0, y1, d0, d1, gy, gl):
    # (same statements, same order as the device function)
    w, h = x1 - x0, y1 - y0
    iw = 1 / w
    s, z = h * iw, (x - x0) * iw
    omz = 1 - z
    zz = z * omz
    q = d0 + d1 - 2 * s
    den, num = s + q * zz, s * z * z + d0 * zz
    iden = 1 / den
    r = num * iden
    P = 2 * s * zz + d0 * omz * omz + d1 * z * z
    Pb = gl / P
    rb = gy * h
    numb = rb * iden
    denb = -(2 * gl + rb * r) * iden
    sb = 2 * gl / s + Pb * 2 * zz + denb * (1 - 2 * zz) + numb * z * z
    d0b = Pb * omz * omz + (denb + numb) * zz
    d1b = Pb * z * z + denb * zz
    zzb = Pb * 2 * s + denb * q + numb * d0
    omzb = Pb * 2 * d0 * omz + zzb * z
    zb = Pb * 2 * d1 * z + numb * 2 * s * z + zzb * omz - omzb
    xb = zb * iw
    hb = gy * r + sb * iw
    wb = -(zb * z + sb * s) * iw
    return [xb, -xb - wb, wb, gy - hb, hb, d0b, d1b]


def test_rqs_local_vjp_is_the_reverse_sweep_of_the_bin_map():
    torch.manual_seed(0)
    n = 4000
    f = torch.float64
    x0 = torch.randn(n, dtype=f)
    x1 = x0 + torch.rand(n, dtype=f) * 3 + 0.05
    x = x0 + (x1 - x0) * torch.rand(n, dtype=f)
    y0 = torch.randn(n, dtype=f)
    y1 = y0 + torch.rand(n, dtype=f) * 3 + 0.05
    d0, d1 = torch.rand(n, dtype=f) * 4 + 0.05, torch.rand(n, dtype=f) * 4 + 0.05
    gy, gl = torch.randn(n, dtype=f), torch.randn(n, dtype=f)
    leaves = [v.clone().requires_grad_() for v in (x, x0, x1, y0, y1, d0, d1)]
    y, ladj = _forward(*leaves)
    ((y * gy).sum() + (ladj * gl).sum()).backward()
    mine = _rqs_local_vjp(x, x0, x1, y0, y1, d0, d1, gy, gl)
    for name, leaf, got in zip(("x", "x0", "x1", "y0", "y1", "d0", "d1"), leaves, mine):
        err = ((leaf.grad - got).abs() / leaf.grad.abs().clamp_min(1.0)).max().item()
        assert err < 1e-10, (name, err)
