r"""Batch sharding of the hot path across the GPUs of a node.

The reference has no parallelism.  The path shards naturally: every sample's
log-density depends only on its own row of x (and c) and on replicated parameters, so ranks own
contiguous row ranges and exchange NOTHING on the data path.  The only collective is one
all-reduce(sum) of the scalar negative log-likelihood (and the row count when shards are uneven)
per step — `torch.distributed` backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the
CPU tests.
"""

from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist
from torch import Tensor


def shard_bounds(n_rows: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced row range [lo, hi) of `rank` (first n_rows % world ranks get one more)."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rows(t: Tensor | None, rank: int, world: int) -> Tensor | None:
    if t is None:
        return None
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi]


def sharded_nll(log_prob_fn: Callable[[Tensor, Tensor | None], Tensor], x_local: Tensor, c_local: Tensor | None = None,
                group=None, sum_fn: Callable[[Tensor], Tensor] | None = None) -> Tensor:
    """Mean negative log-likelihood over ALL ranks' rows.

    `log_prob_fn(x_local, c_local)` evaluates this rank's shard (on the GPU: `flow(c).log_prob(x)`,
    i.e. the HIP kernels); `sum_fn` reduces it to an f64 scalar (default: `zuko_amd.ops.sum_f64` on
    HIP tensors, plain f64 sum otherwise).  One all-reduce of [sum, count] (16 bytes)."""
    lp = log_prob_fn(x_local, c_local)
    if sum_fn is None:
        if lp.is_cuda:
            from . import ops

            total = ops.sum_f64(lp, 1.0)
        else:
            total = lp.double().sum()
    else:
        total = sum_fn(lp)
    pack = torch.stack((total.reshape(()).double(), torch.tensor(float(lp.numel()), dtype=torch.float64, device=total.device)))
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(pack, op=dist.ReduceOp.SUM, group=group)
    return -pack[0] / pack[1]
