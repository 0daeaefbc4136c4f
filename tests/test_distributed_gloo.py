"""CPU, world_size 2, gloo: the batch-sharded NLL (zuko_amd.distributed) equals the single-process
value.  The per-shard log_prob is injected (the CPU oracle stands in for the HIP kernels, which need
a GPU); what is tested is the sharding arithmetic and the one-scalar all-reduce."""

import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank: int, world: int, port: int, n_rows: int, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from conftest import build_flow, oracle_spec
    from oracle import zuko_oracle as O
    from zuko_amd.distributed import shard_rows, sharded_nll

    flow, entry = build_flow("nsf_cfg1")
    spec = oracle_spec(flow, entry)
    g = torch.Generator().manual_seed(1)
    x, c = torch.randn(n_rows, 3, generator=g), torch.randn(n_rows, 5, generator=g)
    with torch.no_grad():
        nll = sharded_nll(lambda xs, cs: O.flow_log_prob(spec, xs, cs), shard_rows(x, rank, world), shard_rows(c, rank, world))
        if rank == 0:
            ref = -O.flow_log_prob(spec, x, c).double().mean()
            out.put((nll.item(), ref.item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [1000, 1001])
def test_sharded_nll_two_ranks(n_rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_rows) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_rows, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, ref = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(got - ref) < 1e-9 * max(1.0, abs(ref))


def test_shard_bounds_cover_everything():
    sys.path.insert(0, ROOT)
    from zuko_amd.distributed import shard_bounds

    for n in (0, 1, 7, 8, 1001, 1 << 20):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
