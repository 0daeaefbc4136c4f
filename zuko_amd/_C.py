r"""ctypes binding of libzuko_amd.so (the C-ABI declared in include/zuko_amd.h).

There is deliberately NO fallback: if the shared library is missing or a symbol cannot be
resolved, importing this module raises — the product path never degrades to PyTorch ops or to
the CPU oracle.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_double, c_int, c_int64, c_void_p

import torch  # noqa: F401  MUST precede the dlopen below: libzuko_amd.so shares torch's libamdhip64.so

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZUKO_AMD_LIB", os.path.join(_HERE, "lib", "libzuko_amd.so"))

P = c_void_p
I = c_int
L = c_int64
F = c_double

# symbol -> argument types (return type is always int = hipError_t)
SIGNATURES = {
    "zk_rqs_forward": [I, L, L, I, F, F, P, P, L, L, P, L, L, P, L, L, P, P, I, P, P],
    "zk_rqs_inverse": [I, L, L, I, F, F, P, P, L, L, P, L, L, P, L, L, P, P, P],
    "zk_rqs_diag": [I, L, L, I, F, F, P, P, L, L, P, L, L, P, L, L, P, P, P, P, P],
    "zk_rqs_from_knots": [I, I, L, L, I, P, P, P, P, L, L, P, P, P, P],
    "zk_affine_forward": [I, L, L, F, P, P, L, L, P, L, L, P, P, I, P],
    "zk_affine_inverse": [I, L, L, F, P, P, L, L, P, L, L, P, P],
    "zk_sos_forward": [I, L, L, I, I, F, POINTER(c_double), POINTER(c_double), P, P, L, L, P, L, L, P, P, I, P],
    "zk_sos_inverse": [I, L, L, I, I, F, POINTER(c_double), POINTER(c_double), I, P, P, L, L, P, L, L, P, P],
    "zk_bernstein_forward": [I, L, L, I, I, F, P, P, L, L, P, P, I, P],
    "zk_bernstein_inverse": [I, L, L, I, I, F, I, P, P, L, L, P, P],
    "zk_linear": [I, L, I, I, P, L, P, P, P, I, P, L, P],
    "zk_linear_bf16": [L, I, I, P, L, P, P, P, I, P, L, P],
    "zk_linear_bf16_rqs": [L, I, I, P, L, P, P, P, I, I, F, F, P, L, P, L, P, P, P],
    "zk_diag_normal_log_prob": [I, L, L, P, P, P, P, P, P],
    "zk_sum_f64": [I, L, P, F, P, P, P],
    "zk_gather_f32": [P, P, P, L, P, P],
    "zk_univariate_backward": [I, L, L, I, F, F, P, P, P, P, I, P, P, P],
    "zk_diag_normal_backward": [L, L, P, P, P, P, P, P],
    "zk_act_backward": [L, P, P, I, P, P],
    "zk_inverse_seed": [L, P, P, P, P, P],
    "zk_sos_backward": [L, L, I, I, F, POINTER(c_double), POINTER(c_double), I, P, P, P, P, I, P, P, P],
    "zk_bernstein_backward": [L, L, I, I, F, P, P, P, P, I, P, P, P],
    "zk_gemm_f32_skip": [L, I, I, P, L, P, P, P, I, P, L, I, P, L, P],
    "zk_wgrad_slices": [L, I],
    "zk_wgrad_f32": [L, I, I, P, L, P, L, P, I, P, P, P, I, P],
    "zk_wgrad_bias_f32": [L, I, I, P, L, P, L, P, I, P, P, P, I, P, P, P, P],
    "zk_colsum_slices": [L],
    "zk_colsum_f32": [L, I, P, L, P, P, I, P],
    "zk_ar_lds_bytes": [I, I],
    "zk_ar_forward_static": [P, I, I, L, I, I, P, L, P, L, P, I, P, P, I, P, I, I, I, F, F, P],
    "zk_ar_forward_train": [P, I, I, L, I, I, P, L, P, P, P, P, L, P, P, I, P, I, I, I, P],
    "zk_ar_forward": [I, L, I, I, P, L, P, L, P, I, P, P, I, P, P, I, I, I, I, F, F, I, P],
    "zk_ar_forward_diag": [I, L, I, I, P, L, P, L, P, P, P, I, P, P, I, I, I, I, F, F, P, P, P],
    "zk_ar_inverse_sweep": [I, L, I, I, P, L, P, L, P, L, P, P, I, P, P, I, I, I, I, F, F, I, P],
    "zk_coupling_forward": [L, I, I, P, L, P, L, P, L, P, I, P, P, I, POINTER(c_int), P, I, P, I, I, POINTER(c_int), POINTER(c_int), I, I, F, I, P],
    "zk_coupling_inverse": [L, I, I, P, L, P, L, P, L, P, I, P, P, I, POINTER(c_int), P, I, P, I, I, POINTER(c_int), POINTER(c_int), I, I, F, I, P],
    "zk_ar_inverse_incremental": [I, I, L, I, I, P, L, P, L, P, L, P, P, P, I, POINTER(c_int), P, P, I, I, I, F, F, P],
    "zk_ar_inc_lds_bytes": [I, I],
    "zk_ar_inverse_partial": [I, L, I, I, P, L, P, L, P, L, P, P, I, P, P, I, I, I, I, F, F, P, I, POINTER(c_int), I, I, I, P],
}


class _Lib:
    def __init__(self, path: str) -> None:
        if not os.path.exists(path):
            raise ImportError(
                f"zuko_amd: HIP library not found at {path}. Build it with `python zuko_amd/_build.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU / PyTorch fallback."
            )
        self.path = path
        self.cdll = ctypes.CDLL(path)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(self.cdll, name)  # AttributeError -> loud failure on a stale library
            fn.argtypes = argtypes
            fn.restype = c_int
            setattr(self, name, _timed(name, fn))


# optional per-entry-point timing with events on the launch stream (used by bench.py for the
# roofline line): PROFILE = {} enables it, PROFILE = None (default) disables it.
PROFILE: dict | None = None


def _timed(name, fn):
    def call(*args):
        if PROFILE is None:
            return fn(*args)
        import torch

        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        err = fn(*args)
        b.record()
        PROFILE.setdefault(name, []).append((a, b, args))
        return err

    return call


_lib: _Lib | None = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib(LIB_PATH)
    return _lib


def check(err: int, what: str) -> None:
    if err != 0:
        raise RuntimeError(f"zuko_amd: {what} failed with hipError_t {err}")
