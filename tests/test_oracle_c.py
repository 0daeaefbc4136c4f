"""CPU: the plain-C restatement (oracle/zuko_oracle_c.c, gcc + libm, no torch) against the golden vectors that
were generated from the live reference — an independent check that the fixtures mean what the reference
lines say (double precision: the two implementations differ only by libm-vs-Sleef rounding)."""

import ctypes

import numpy as np
import pytest

from conftest import golden


@pytest.fixture(scope="module")
def clib():
    from oracle.build_c import build

    lib = ctypes.CDLL(build())
    P, L, I, F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
    lib.zoc_rqs_forward.argtypes = [L, I, F, F, P, P, P, P, P, P, P]
    lib.zoc_rqs_inverse.argtypes = [L, I, F, F, P, P, P, P, P, P]
    lib.zoc_affine.argtypes = [L, F, P, P, P, P, P, P]
    for f in (lib.zoc_rqs_forward, lib.zoc_rqs_inverse, lib.zoc_affine):
        f.restype = None
    return lib


def ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def test_c_restatement_of_the_spline_matches_the_golden_vectors(clib):
    g = golden("rqs_f64.npz")
    w, h, d, x = (np.ascontiguousarray(g[k]) for k in ("widths", "heights", "derivatives", "x"))
    n, K = x.size, w.shape[-1]
    y, ladj, k = np.empty(n), np.empty(n), np.empty(n, dtype=np.int64)
    clib.zoc_rqs_forward(n, K, 5.0, 1e-3, ptr(x), ptr(w), ptr(h), ptr(d), ptr(y), ptr(ladj), ptr(k))
    gk = g["k"].reshape(-1)
    same = k == gk  # the adversarial rows sit exactly on knots: a last-bit difference in a knot may move the bin
    assert (~same).sum() <= 8, f"{(~same).sum()} bin indices differ"
    np.testing.assert_allclose(y[same], g["y"].reshape(-1)[same], rtol=1e-11, atol=1e-11, equal_nan=True)
    np.testing.assert_allclose(ladj[same], g["ladj"].reshape(-1)[same], rtol=1e-9, atol=1e-9, equal_nan=True)
    assert np.array_equal(np.isnan(y), np.isnan(g["y"].reshape(-1)))
    yin = np.ascontiguousarray(g["y_in"])
    xi, ki = np.empty(n), np.empty(n, dtype=np.int64)
    clib.zoc_rqs_inverse(n, K, 5.0, 1e-3, ptr(yin), ptr(w), ptr(h), ptr(d), ptr(xi), ptr(ki))
    same_i = ki == g["k_inv"].reshape(-1)
    assert (~same_i).sum() <= 8
    np.testing.assert_allclose(xi[same_i], g["x_inv"].reshape(-1)[same_i], rtol=1e-9, atol=1e-9, equal_nan=True)


def test_c_restatement_of_the_affine_map_matches_the_golden_vectors(clib):
    g = golden("affine_f64.npz")
    x, shift, scale = (np.ascontiguousarray(g[k]) for k in ("x", "shift", "scale"))
    n = x.size
    y, ladj, back = np.empty(n), np.empty(n), np.empty(n)
    clib.zoc_affine(n, 1e-3, ptr(x), ptr(shift), ptr(scale), ptr(y), ptr(ladj), ptr(back))
    np.testing.assert_allclose(y, g["y"].reshape(-1), rtol=1e-13, atol=1e-13, equal_nan=True)
    np.testing.assert_allclose(ladj, g["ladj"].reshape(-1), rtol=1e-13, atol=1e-13, equal_nan=True)
    ok = np.isfinite(y)
    np.testing.assert_allclose(back[ok], x.reshape(-1)[ok], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name,D,C,K", [("nsf_cfg1", 3, 5, 8), ("nsf_cfg2", 64, 0, 8), ("maf_cfg3", 64, 0, 0), ("maf_doc", 3, 4, 0)])
def test_c_restatement_of_a_whole_flow_matches_the_flow_golden(clib, name, D, C, K):
    """Autoregressive flows (NSF: K-bin spline, MAF: affine map, K = 0) in double-precision C from the module's weights and
    mask buffers, against z / ladj / log_prob the reference produced in float32."""
    import torch

    from conftest import build_flow

    flow, entry = build_flow(name)  # (seeded reconstruction, state_dict SHA-256 checked against the fixture)
    g = golden(f"flow_{name}.npz")
    ts = list(flow.transform.transforms)
    lins = [[m for m in t.hyper if hasattr(m, "mask")] for t in ts]
    T_, L = len(ts), len(lins[0])
    dims = np.array([lins[0][0].weight.shape[1]] + [l.weight.shape[0] for l in lins[0]], dtype=np.int32)
    W = [np.ascontiguousarray(l.weight.detach().double().numpy()) for tl in lins for l in tl]
    M = [np.ascontiguousarray(l.mask.numpy().astype(np.uint8)) for tl in lins for l in tl]
    B = [np.ascontiguousarray(l.bias.detach().double().numpy()) for tl in lins for l in tl]
    arr = lambda xs: (ctypes.c_void_p * len(xs))(*[a.ctypes.data for a in xs])
    n = min(128, g["x"].shape[0])
    x = np.ascontiguousarray(g["x"][:n].astype(np.float64))
    c = np.ascontiguousarray(g["c"][:n].astype(np.float64)) if C else np.zeros((n, 0))
    z, ladj, lp = np.empty((n, D)), np.empty(n), np.empty(n)
    clib.zoc_nsf_log_prob.restype = None
    clib.zoc_nsf_log_prob.argtypes = [ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double] + [ctypes.c_void_p] * 8
    clib.zoc_nsf_log_prob(n, D, C, T_, L, ptr(dims), K, 5.0, 1e-3, ptr(x), ptr(c), arr(W), arr(M), arr(B), ptr(z), ptr(ladj), ptr(lp))
    np.testing.assert_allclose(z, g["z"][:n], rtol=5e-5, atol=5e-5)
    np.testing.assert_allclose(ladj, g["ladj"][:n], rtol=5e-5, atol=2e-4)
    np.testing.assert_allclose(lp, g["log_prob"][:n], rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("name,D,C", [("nice_small", 5, 3), ("realnvp_cfg4", 256, 0)])
def test_c_restatement_of_a_coupling_flow_matches_the_flow_golden(clib, name, D, C):
    """NICE / RealNVP (dense MLP conditioner, affine map on the moved half) in double-precision C from the module's
    weights and mask buffers, against the reference's float32 z / ladj / log_prob."""
    from conftest import build_flow

    flow, entry = build_flow(name)
    g = golden(f"flow_{name}.npz")
    ts = list(flow.transform.transforms)
    lins = [[m for m in t.hyper if hasattr(m, "weight")] for t in ts]
    T_, L = len(ts), len(lins[0])
    dims = np.array([d for tl in lins for d in [tl[0].weight.shape[1]] + [l.weight.shape[0] for l in tl]], dtype=np.int32)  # per transform
    W = [np.ascontiguousarray(l.weight.detach().double().numpy()) for tl in lins for l in tl]
    B = [np.ascontiguousarray(l.bias.detach().double().numpy()) for tl in lins for l in tl]
    masks = [np.ascontiguousarray(t.mask.numpy().astype(np.uint8)) for t in ts]
    arr = lambda xs: (ctypes.c_void_p * len(xs))(*[a.ctypes.data for a in xs])
    n = min(64, g["x"].shape[0])
    x = np.ascontiguousarray(g["x"][:n].astype(np.float64))
    c = np.ascontiguousarray(g["c"][:n].astype(np.float64)) if C else np.zeros((n, 0))
    z, ladj, lp = np.empty((n, D)), np.empty(n), np.empty(n)
    clib.zoc_coupling_affine_log_prob.restype = None
    clib.zoc_coupling_affine_log_prob.argtypes = [ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_double] + [ctypes.c_void_p] * 8
    clib.zoc_coupling_affine_log_prob(n, D, C, T_, L, ptr(dims), 1e-3, ptr(x), ptr(c), arr(masks), arr(W), arr(B), ptr(z), ptr(ladj), ptr(lp))
    np.testing.assert_allclose(z, g["z"][:n], rtol=5e-5, atol=5e-5)
    np.testing.assert_allclose(ladj, g["ladj"][:n], rtol=5e-5, atol=2e-4)
    np.testing.assert_allclose(lp, g["log_prob"][:n], rtol=2e-5, atol=2e-4)


def test_c_restatement_of_the_sos_transform_matches_the_golden_vectors(clib):
    """Sum-of-squares polynomial: Gauss-Legendre integral with the reference's n = L + 1 nodes, log of the integrand,
    25-step bisection inverse of y (x_inv = inverse(forward(x)) in the fixture)."""
    import math

    g = golden("sos_f64.npz")
    a, x = np.ascontiguousarray(g["a"]), np.ascontiguousarray(g["x"])
    nodes, weights = np.ascontiguousarray(g["gl_nodes01"]), np.ascontiguousarray(g["gl_weights01"])
    n, P, L1 = x.size, a.shape[-2], a.shape[-1]
    yin = np.ascontiguousarray(g["y"])
    y, ladj, xinv = np.empty(n), np.empty(n), np.empty(n)
    clib.zoc_sos.restype = None
    clib.zoc_sos.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6
    nbis = math.ceil(math.log2(2 * 10.0 / 1e-6))
    clib.zoc_sos(n, P, L1, 1e-3, ptr(nodes), ptr(weights), len(nodes), nbis, ptr(x), ptr(a), ptr(yin), ptr(y), ptr(ladj), ptr(xinv))
    np.testing.assert_allclose(y, g["y"].reshape(-1), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ladj, g["ladj"].reshape(-1), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(xinv, g["x_inv"].reshape(-1), rtol=0, atol=2e-6)  # bisection: interval width 20 / 2^25


@pytest.mark.parametrize("name,bounded", [("bern", 0), ("bbern", 1)])
def test_c_restatement_of_the_bernstein_transforms_matches_the_golden_vectors(clib, name, bounded):
    """(Bounded) Bernstein polynomial: constrained coefficients, value with linear tails and the closed-form derivative
    (the reference obtains it by autograd) against the fixture's theta / y / ladj."""
    g = golden(f"{name}_f64.npz")
    th, x = np.ascontiguousarray(g["theta"]), np.ascontiguousarray(g["x"])
    n, M = x.size, th.shape[-1]
    nc = g["theta_constrained"].shape[-1]
    y, ladj, tc = np.empty(n), np.empty(n), np.empty((n, nc))
    clib.zoc_bernstein.restype = None
    clib.zoc_bernstein.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_double] + [ctypes.c_void_p] * 5
    clib.zoc_bernstein(n, M, bounded, 5.0, ptr(x), ptr(th), ptr(y), ptr(ladj), ptr(tc))
    np.testing.assert_allclose(tc, g["theta_constrained"].reshape(n, nc), rtol=1e-12, atol=1e-12)
    # (measured: 2e-14 / 6e-14 — the closed form IS the reference's Beta-pdf form + autograd derivative in float64)
    np.testing.assert_allclose(y, g["y"].reshape(-1), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ladj, g["ladj"].reshape(-1), rtol=1e-12, atol=1e-12)
