r"""Flow factories of the hot path: MAF / NSF (autoregressive), NICE / RealNVP (coupling),
SOSPF / BPF (polynomial).  Same constructor signatures and module trees as zuko.flows."""

from .autoregressive import MAF, MaskedAutoregressiveTransform
from .coupling import NICE, GeneralCouplingTransform, RealNVP
from .elementwise import ElementWiseTransform
from .polynomial import BPF, SOSPF
from .spline import NCSF, NSF

__all__ = [
    "BPF",
    "MAF",
    "NCSF",
    "NICE",
    "NSF",
    "SOSPF",
    "ElementWiseTransform",
    "GeneralCouplingTransform",
    "MaskedAutoregressiveTransform",
    "RealNVP",
]
