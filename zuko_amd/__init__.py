r"""zuko_amd — MI355X-native (gfx950) implementation of the zuko transform hot path.

`zuko_amd.flows.{NSF, MAF, RealNVP, ...}`, `zuko_amd.transforms.*` and `zuko_amd.nn.*` mirror the
reference's API for this path; all arithmetic runs in hand-written HIP kernels reached through
the C-ABI in include/zuko_amd.h (ctypes, `zuko_amd._C`).  Importing the package requires the built
shared library: there is no CPU or eager fallback.
"""

from . import _C

_C.lib()  # fail loudly at import time if libzuko_amd.so is missing or stale

from . import distributions, flows, lazy, nn, ops, transforms, utils  # noqa: E402

__version__ = "0.1.0"
__all__ = ["distributions", "flows", "lazy", "nn", "ops", "transforms", "utils"]
