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

# ---- versioned argument blocks: the ctypes.Structure classes are built by PARSING include/zuko_amd.h, so the binding
#      cannot drift from the header (field order, types, sizes) --------------------------------------------------------
_HEADER = os.path.join(os.path.dirname(_HERE), "include", "zuko_amd.h")
_CTYPES = {"uint32_t": ctypes.c_uint32, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "double": ctypes.c_double, "float": ctypes.c_float,
           "int": ctypes.c_int}


def _parse_structs(path: str) -> dict:
    import re

    text = open(path).read()
    out = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} (\w+);", text, flags=re.S):
        name, body = m.group(3), re.sub(r"/\*.*?\*/", "", m.group(2), flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            fm = re.match(r"(?:const\s+)?(\w+)\s*(\*?)\s*(\w+)$", decl)
            if not fm:
                raise ImportError(f"zuko_amd: cannot parse field '{decl}' of {name} in {path}")
            typ, star, fname = fm.groups()
            fields.append((fname, c_void_p if star else _CTYPES[typ]))
        out[name] = type(name, (ctypes.Structure,), {"_fields_": fields})
    return out


STRUCTS = _parse_structs(_HEADER)
_VERSION = {"zk_ar_args_v1": 1, "zk_coupling_args_v1": 1, "zk_ar_inc_args_v1": 1}


def args(struct: str, **fields):
    """A filled argument block: struct_size / version set, every keyword must name a field (a typo raises instead of being ignored);
    tensors' data pointers as ints / c_void_p, HOST int arrays as ctypes arrays (kept alive on the returned object)."""
    cls = STRUCTS[struct]
    a = cls()
    a.struct_size, a.version = ctypes.sizeof(cls), _VERSION[struct]
    names = {f for f, _ in cls._fields_}
    keep = []
    for k, v in fields.items():
        if k not in names:
            raise KeyError(f"zuko_amd: {struct} has no field '{k}'")
        if isinstance(v, ctypes.Array):
            keep.append(v)
            v = ctypes.cast(v, c_void_p)
        if isinstance(v, c_void_p):
            v = v.value
        setattr(a, k, 0 if v is None else v)
    a._keep = keep
    return a


def stream() -> int:
    """The current HIP stream's handle.  (torch.cuda.current_stream() builds a Stream object through three Python layers: 9 us per call,
    1 ms of a RealNVP training step; the raw getter is 0.3 us.)"""
    import torch

    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def gather_multi(items, stream) -> None:
    """items: [(src, mask, idx, count, dst, split)] of tensors / None — zk_gather_multi in launches of up to eight gathers."""
    cls = STRUCTS["zk_gather_desc_v1"]
    items = [it for it in items if it[3] > 0]
    for i0 in range(0, len(items), 8):
        part = items[i0 : i0 + 8]
        arr = (cls * len(part))()
        for d, (src, mask, idx, count, dst, split) in zip(arr, part):
            d.struct_size, d.split, d.count = ctypes.sizeof(cls), int(split), int(count)
            d.src, d.mask, d.idx, d.dst = src.data_ptr(), (None if mask is None else mask.data_ptr()), idx.data_ptr(), dst.data_ptr()
        check(lib().zk_gather_multi(len(part), ctypes.cast(arr, c_void_p), stream), "zk_gather_multi")


_AR, _CP, _INC = POINTER(STRUCTS["zk_ar_args_v1"]), POINTER(STRUCTS["zk_coupling_args_v1"]), POINTER(STRUCTS["zk_ar_inc_args_v1"])

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
    "zk_bernstein_forward": [I, L, L, I, I, F, F, P, P, L, L, P, P, I, P],
    "zk_bernstein_inverse": [I, L, L, I, I, F, F, I, P, P, L, L, P, P],
    "zk_linear": [I, L, I, I, P, L, P, P, P, I, P, L, P],
    "zk_linear_bf16": [L, I, I, P, L, P, P, P, I, P, L, P],
    "zk_linear_bf16_rqs": [L, I, I, P, L, P, P, P, I, I, F, F, P, L, P, L, P, P, P],
    "zk_linear_bf16_rqs_lanes": [L, I, I, P, L, P, P, P, I, I, F, F, P, L, P, L, P, P, P],
    "zk_diag_normal_log_prob": [I, L, L, P, P, P, P, P, P],
    "zk_sum_f64": [I, L, P, F, P, P, P],
    "zk_gather_f32": [P, P, P, L, P, P],
    "zk_gather_split_bf16": [P, P, P, L, P, P],
    "zk_gather_split_f16": [P, P, P, L, P, F, P],
    "zk_gather_multi": [I, P, P],
    "zk_univariate_backward": [I, L, L, I, F, F, P, P, P, P, I, P, P, P],
    "zk_diag_normal_backward": [L, L, P, P, P, P, P, P],
    "zk_act_backward": [L, P, P, I, P, P],
    "zk_inverse_seed": [L, P, P, P, P, P],
    "zk_sos_backward": [L, L, I, I, F, POINTER(c_double), POINTER(c_double), I, P, P, P, P, I, P, P, P],
    "zk_bernstein_backward": [L, L, I, I, F, F, P, P, P, P, I, P, P, P],
    "zk_gemm_f32_skip": [L, I, I, P, L, P, P, P, I, P, L, I, P, L, P],
    "zk_wgrad_slices": [L, I],
    "zk_wgrad_f32": [L, I, I, P, L, P, L, P, I, P, P, P, I, P, P, P],
    "zk_wgrad_bias_f32": [L, I, I, P, L, P, L, P, I, P, P, P, I, P, P, P, P, P, P],
    "zk_colsum_slices": [L],
    "zk_colsum_f32": [L, I, P, L, P, P, I, P],
    "zk_amax_f32": [I, P, P],
    "zk_wsplit_f16": [I, P, P],
    "zk_gemm_f16x2": [L, I, I, P, L, P, P, P, P, I, P, L, I, P, L, P, P],
    "zk_coupling_split": [L, I, I, P, L, P, L, P, I, P, I, P, P, P, P],
    "zk_coupling_merge": [L, I, P, L, P, I, P, L, P, P, P],
    "zk_ar_lds_bytes": [I, I],
    "zk_ar_forward_static": [_AR, P],
    "zk_ar_forward_train": [_AR, P],
    "zk_ar_dgrad_chain": [_AR, P],
    "zk_ar_dgrad_full": [_AR, P],
    "zk_ar_backward_full": [_AR, P],
    "zk_wgrad_multi": [I, P, L, P],
    "zk_ar_forward": [_AR, P],
    "zk_ar_forward_split": [_AR, P],
    "zk_ar_forward_diag": [_AR, P],
    "zk_ar_inverse_sweep": [_AR, P],
    "zk_coupling_forward": [_CP, P],
    "zk_coupling_inverse": [_CP, P],
    "zk_ar_inverse_incremental": [_INC, P],
    "zk_ar_inc_lds_bytes": [I, I],
    "zk_ar_inverse_partial": [_AR, P],
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
