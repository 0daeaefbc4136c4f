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
