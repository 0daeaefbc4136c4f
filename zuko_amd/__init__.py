r"""zuko_amd — MI355X-native (gfx950) implementation of the zuko transform hot path.

`zuko_amd.flows.{NSF, MAF, RealNVP, ...}`, `zuko_amd.transforms.*` and `zuko_amd.nn.*` mirror the
reference's API for this path; all arithmetic runs in hand-written HIP kernels reached through
the C-ABI in include/zuko_amd.h (ctypes, `zuko_amd._C`).  Importing the package requires the built
shared library: there is no CPU or eager fallback.
"""

from . import _C

_C.lib()  # fail loudly at import time if libzuko_amd.so is missing or stale

from . import distributions, flows, lazy, nn, ops, transforms, utils  # noqa: E402
from .fused import matmul_precision, set_matmul_precision  # noqa: E402
from .graph import capture, capture_step  # noqa: E402

__version__ = "0.2.0"
__all__ = ["capture", "capture_step", "distributions", "flows", "invalidate", "lazy", "matmul_precision", "nn", "ops", "set_matmul_precision", "transforms", "utils"]


def invalidate(module) -> None:
    """Drop every device-side table derived from `module`'s parameters / buffers (fused weight streams, bf16 `mask * W`
    tables, sweep schedules).  The tables follow `Tensor._version`, which optimizers, `copy_` and `load_state_dict` bump;
    call this after writing parameters through `.data` or raw pointers (EMA swaps, legacy initialisers), which do not."""
    from .flows import autoregressive as _ar

    for m in module.modules():
        _ar._FUSED_CACHE.pop(m, None)
        m.__dict__.pop("_bf16_plan_cache", None)
        m.__dict__.pop("_coupling_cache", None)
        m.__dict__.pop("_split_cache", None)
        for k in ("_sweep_cache", "_unit_cache", "_unit_rows_cache"):  # wavefront form of the layer-wise inverse (flows/autoregressive.py)
            m.__dict__.pop(k, None)
        from . import train as _train
        from .flows import coupling as _cp

        _cp._COUPLING_CACHE.pop(m, None)

        _train._PLAN_CACHE.pop(m, None)
