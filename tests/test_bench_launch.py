"""bench.py --gpus N must create the N ranks itself (torch.distributed.run on 127.0.0.1) when it is not already
running under a launcher, and must report the world size it actually ran with.

CPU: the launch path alone (ZUKO_BENCH_LAUNCH_SELFTEST=1: ranks rendezvous over gloo, all-reduce a one, rank 0 prints).
GPU: the real benchmark as 2 ranks sharing the box's single GPU (dry-run environment), small batch."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout):
    env = dict(os.environ)
    env.update(env_extra)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"expected ONE json line, got {len(lines)}:\n{p.stdout[-2000:]}"
    assert p.stdout.rstrip().splitlines()[-1] == lines[0], "the JSON line is the LAST stdout line"
    assert len(lines[0]) < 8192, f"the printed line must stay under 8 KB (round 5's 20 KB line was not parsed by the driver): {len(lines[0])} bytes"
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_gpus_flag_spawns_ranks(n):
    """1, 2, 3 and the full node's 8 ranks (gloo): every rank joins, and every rank pins itself to its own slice of the host's cores
    (on a GPU box: the cores local to its GPU's NUMA node)."""
    out = _run(["--gpus", str(n)], {"ZUKO_BENCH_LAUNCH_SELFTEST": "1"}, 300)
    assert out["n_gpus"] == n and out["rccl_world_size"] == n and out["allreduce_of_ones"] == n
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= n:
        assert out["ranks_bound"] == n and len(set(out["first_cpu_per_rank"])) == n, out


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, ZUKO_BENCH_LAUNCH_SELFTEST="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    out = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch-log2", "14", "--no-cpu-baseline"],
               {"ZUKO_BENCH_SINGLE_DEVICE": "1", "ZUKO_BENCH_BACKEND": "gloo"}, 900)
    assert out["n_gpus"] == 2 and out["rccl_world_size"] == 2 and len(out["per_rank_ms_per_step"]) == 2
    assert out["config"]["global_batch"] == 2 << 14 and out["value"] > 0
    assert out["roofline"]["frac"] <= 1.0
    # the multi-rank line: rank 0's solo run before the group formed and the efficiency derived from it (the driver computes its own from per-N runs)
    assert out["rank0_alone_before_group"]["value"] > 0 and 0.0 < out["weak_scaling_efficiency"] < 2.0 and out["scaling"] == "weak"


@pytest.mark.gpu
def test_bench_refuses_more_ranks_than_gpus():
    import torch

    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ZUKO_BENCH_SINGLE_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--batch-log2", "10"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "visible GPU" in (p.stderr + p.stdout)
