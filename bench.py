#!/usr/bin/env python
r"""Headline benchmark: log_prob samples/s of NSF(features=64, context=0, transforms=8, bins=8,
hidden=[256]*3) at batch 2^20 per GPU (BASELINE.json configs[1]), fp32, synthetic N(0,1) inputs
resident in HBM, random-init weights from torch.manual_seed(0) (the reference's constructor order).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch-log2 20] [--no-cpu-baseline]

One "step" = one pass of the hot path over the rank's batch: flow().log_prob(x) for all 2^20 rows,
the f64 reduction to the mean NLL, and (N > 1) ONE all-reduce of that scalar over RCCL.  Multi-GPU
is weak scaling: every rank owns its own 2^20-row shard; nothing but the scalar crosses xGMI.

`--gpus N` with N > 1 and no torchrun environment re-launches this script under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per
GPU, rank -> device LOCAL_RANK, backend nccl = RCCL); under the driver's own torchrun launch the
ranks are used as they come.  It refuses to run N ranks on fewer than N visible GPUs unless the dry-run
environment ZUKO_BENCH_SINGLE_DEVICE=1 ZUKO_BENCH_BACKEND=gloo is set (code-path check on a 1-GPU box).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, measured with events on the launch
stream; `achieved` counts the ALGORITHMIC = non-zero-weight FLOPs of SURVEY 8(d), so `frac` <= 1) and
`cpu_baseline` (the CPU oracle = the reference's algorithm on PyTorch-CPU ops, bitwise-pinned to the
reference in the build container, timed on this host's cores on a bounded sample).
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FEATURES, TRANSFORMS, BINS, HIDDEN = 64, 8, 8, [256, 256, 256]
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBPS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA, same guide
# What a SIMD can issue v_mfma_f32_16x16x32_bf16 at (scripts/probes/mfma_clock_probe.hip, profiles/r05/mfma_clock_probe.txt; real shader cycles from s_memtime):
# with TWO wavefronts per SIMD — the headline kernel's geometry — 16.2-16.5 cycles per instruction = the matrix pipe's own rate (16), i.e. 0.96 of the
# 2.5 PFLOP/s peak on zero operands and 0.88 on random ones (the chip clocks down to ~2.2 GHz).  Round 4 quoted 0.61 from a ONE-wavefront probe with
# four accumulators in rotation (23.5 real cycles; one accumulator: 17.0, eight: 31.0) and called it the form's ceiling; it is not.
MFMA_16x16x32_ISSUE_CEILING = 0.88

T_START = time.perf_counter()  # (wall_s in the output line: where this invocation's minutes go)

CONFIGS = {
    # name: (constructor name, kwargs, workload string, bf16)
    "cfg2": ("NSF", dict(features=64, context=0, transforms=8, bins=8, hidden_features=[256] * 3), "NSF(features=64, context=0, transforms=8, bins=8, hidden=[256]*3) log_prob", False),
    "cfg3": ("MAF", dict(features=64, context=0, transforms=8, hidden_features=[256] * 3), "MAF(features=64, transforms=8, hidden=[256]*3) log_prob", False),
    "cfg4": ("RealNVP", dict(features=256, context=0, transforms=16, hidden_features=[512] * 3), "RealNVP(features=256, transforms=16, hidden=[512]*3) log_prob", False),
    "cfg5": ("NSF", dict(features=1024, context=0, transforms=12, bins=16, hidden_features=[1024] * 3), "NSF(features=1024, transforms=12, bins=16, hidden=[1024]*3) bf16 log_prob", True),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bin-report", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--no-side-configs", action="store_true",
                    help="default cfg2 run at N=1: do not append the cfg3 / cfg4 / cfg5 side measurements (each a short run of this script in a process of its own)")
    ap.add_argument("--side-steps", type=int, default=0, help="timed steps of every side configuration (0 = a per-config default)")
    ap.add_argument("--full-line", action="store_true", help="print the full detail object as the line (what the side-config children of a default run do); "
                                                                 "default: the compact headline line (<= 8 KB) on stdout and the full object in gpurun_out/bench_detail.json")
    ap.add_argument("--side-paths", action="store_true", help="print the training / sampling report of the cfg2 and cfg3 flows (the `side_paths` object of the default line) and exit")
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS),
                    help="cfg2 = the headline NSF workload (default, the only graded line); cfg3 = MAF(64,T=8,H=256x3); "
                         "cfg4 = RealNVP(256,T=16,H=512x3); cfg5 = NSF(1024,T=12,K=16,H=1024x3) in bf16 (use --batch-log2 19) — "
                         "side measurements quoted in DESIGN.md")
    return ap.parse_args()


# --------------------------------------------------------------------------------------------------
# launching: `--gpus N` creates the N ranks itself when no torchrun environment is present
# --------------------------------------------------------------------------------------------------


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relaunch_under_torchrun(n: int) -> int:
    """Re-exec this command line as N ranks (one per GPU) and return the launcher's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def bind_cpu_to_gpu(local: int, world: int) -> dict:
    """Pin this rank's host threads to the cores of the NUMA node its GPU hangs off (sysfs: the PCI device's local_cpulist), or,
    when that is unknown, to an even slice of the host's cores — eight ranks left unpinned share the launcher's cores and cross
    sockets for every launch.  Best effort: returns what was done for the JSON line."""
    import torch

    info = {"bound": False}
    try:
        ncpu = os.cpu_count() or 1
        cpus = None
        if torch.cuda.is_available() and local < torch.cuda.device_count():
            pr = torch.cuda.get_device_properties(local)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            path = f"/sys/bus/pci/devices/{bdf}/local_cpulist"
            info["pci"] = bdf
            if os.path.exists(path):
                cpus = set()
                for part in open(path).read().strip().split(","):
                    if part:
                        lo, _, hi = part.partition("-")
                        cpus |= set(range(int(lo), int(hi or lo) + 1))
                node = f"/sys/bus/pci/devices/{bdf}/numa_node"
                if os.path.exists(node):
                    info["numa_node"] = int(open(node).read().strip())
        if not cpus or len(cpus) >= ncpu:  # no topology information: an even slice per rank
            per = max(1, ncpu // max(world, 1))
            cpus = set(range(local * per, min(ncpu, (local + 1) * per)))
            info["source"] = "even slice"
        else:
            info["source"] = "local_cpulist"
        allowed = os.sched_getaffinity(0)
        cpus &= allowed
        if cpus:
            os.sched_setaffinity(0, cpus)
            info.update(bound=True, cpus=len(cpus), first_cpu=min(cpus))
    except Exception as exc:  # never fatal
        info["error"] = repr(exc)
    return info


def init_ranks(args):
    """(rank, world, local device index, dist module or None).  One process per GPU."""
    import torch

    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher created WORLD_SIZE={world} ranks")
    selftest = os.environ.get("ZUKO_BENCH_LAUNCH_SELFTEST") == "1"  # CPU test of the launch path: no device work at all
    if world == 1:
        if not selftest:
            torch.cuda.set_device(0)
        return rank, world, 0, None
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("ZUKO_BENCH_BACKEND", "nccl")
    single = os.environ.get("ZUKO_BENCH_SINGLE_DEVICE") == "1"
    if selftest:
        dist.init_process_group("gloo")
        return rank, world, 0, dist
    if single:
        if backend == "nccl":
            raise SystemExit("bench.py: ZUKO_BENCH_SINGLE_DEVICE=1 needs ZUKO_BENCH_BACKEND=gloo (RCCL cannot put two ranks on one GPU)")
        local = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} visible GPU(s); "
                         "set ZUKO_BENCH_SINGLE_DEVICE=1 ZUKO_BENCH_BACKEND=gloo for a single-GPU dry run of the multi-rank path")
    torch.cuda.set_device(local)

    def form_group():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    # the group is formed by main() AFTER rank 0 has measured the same workload alone (the reference point of the weak-scaling efficiency)
    dist._zuko_form_group = form_group
    return rank, world, local, dist


# --------------------------------------------------------------------------------------------------
# CPU baseline
# --------------------------------------------------------------------------------------------------


REFERENCE_ROOT = "/root/reference"  # exists in the build container only; never on the GPU box


def _reference_flow(flow_cpu):
    """zuko.flows.NSF from the reference checkout with this run's weights, or None when there is no checkout (the GPU box) or it does not import.
    Baseline leg only: nothing else in this file touches the reference."""
    if os.environ.get("ZUKO_BENCH_NO_REFERENCE") == "1" or not os.path.isdir(os.path.join(REFERENCE_ROOT, "zuko")):
        return None
    try:
        sys.path.append(REFERENCE_ROOT)
        import zuko  # noqa: F401

        ref = zuko.flows.NSF(features=FEATURES, context=0, transforms=TRANSFORMS, bins=BINS, hidden_features=HIDDEN)
        ref.load_state_dict(flow_cpu.state_dict())
        return ref
    except Exception:
        return None
    finally:
        if REFERENCE_ROOT in sys.path:
            sys.path.remove(REFERENCE_ROOT)


def cpu_baseline(flow_cpu, seconds: float):
    """The oracle (a restatement of the reference on PyTorch-CPU ops, bitwise equal to it in the build container:
    tests/golden/make_golden.py) timed on this host on chunks of 2^12, 2^14 and 2^16 rows of the same workload (SURVEY
    8d: the reference degrades at larger chunks; `value` is the best of the three).  Returns (json dict, sample) where
    `sample` holds the chunk tensors and the oracle's outputs on them — the SAME rows then go through the GPU path
    (parity_report)."""
    import torch

    from oracle import zuko_oracle as O

    ncpu = os.cpu_count() or 1
    sd = {k: v for k, v in flow_cpu.state_dict().items() if v is not None}
    spec = O.spec_from_state_dict(sd, "ar", O.uni_rqs(BINS), FEATURES)
    chunks = [1 << 12, 1 << 14, 1 << 16]
    xall = torch.randn(chunks[-1], FEATURES, generator=torch.Generator().manual_seed(1))
    keep = {}

    ref_flow = _reference_flow(flow_cpu)  # the reference itself when its checkout is mounted (the build container); the pinned port on the GPU box

    def once(n: int) -> float:
        x = xall[:n]
        t0 = time.perf_counter()
        if ref_flow is not None:  # the body of zuko.distributions.NormalizingFlow.log_prob (zuko/distributions.py:115-119), z and ladj kept for the parity block
            d = ref_flow()
            z, ladj = d.transform.call_and_ladj(x)
            lp = d.base.log_prob(z) + ladj
        else:
            z, ladj = O.flow_forward(spec, x)  # (= the body of O.flow_log_prob: the same lines restated)
            lp = O.diag_normal_log_prob(z, spec.loc, spec.scale) + ladj
        dt = time.perf_counter() - t0
        keep[n] = (z, ladj, lp)
        return dt

    # PyTorch-CPU does not scale to hundreds of threads on these small ops (measured on the 256-thread EPYC 9575F
    # host: 16 threads are fastest, all 256 are ~700x slower): pick the fastest of a bounded sweep up to 64 threads,
    # dropping a candidate as soon as its first pass is 3x off the best, then spend the rest of the budget there.
    best_t, best = None, float("inf")
    sweep = {}
    per_chunk = {}
    with torch.no_grad():
        for threads in sorted({t for t in (4, 8, 16, 32, 64) if 1 <= t <= ncpu} or {1}):
            torch.set_num_threads(threads)
            first = once(chunks[0])
            if first > 3.0 * best:
                sweep[threads] = chunks[0] / first
                continue
            t = min(once(chunks[0]), once(chunks[0]))
            sweep[threads] = chunks[0] / t
            if t < best:
                best_t, best = threads, t
        threads = best_t
        torch.set_num_threads(threads)
        # budget: 40 % on the 2^12 chunk (>= 3 passes), 25 % on 2^14 (>= 2), the rest on 2^16 (>= 1; one pass is 3-15 s)
        for n, share, least in ((chunks[0], 0.40, 3), (chunks[1], 0.25, 2), (chunks[2], 0.35, 1)):
            times = []
            t_end = time.perf_counter() + seconds * share
            while time.perf_counter() < t_end or len(times) < least:
                times.append(once(n))
            times.sort()
            per_chunk[n] = (times[len(times) // 2], len(times), times[0])
    # `value` is the BEST credible CPU figure: the fastest single pass over all chunk sizes (and never below what the thread sweep saw
    # at the chosen setting); the median of the same passes is printed next to it
    medians = {n: n / med for n, (med, _, _) in per_chunk.items()}
    rates = {n: n / fastest for n, (_, _, fastest) in per_chunk.items()}
    rates[chunks[0]] = max(rates[chunks[0]], sweep.get(threads, 0.0))
    n_best = max(rates, key=rates.get)
    out = {
        "value": rates[n_best],
        "value_is": "fastest single pass (best chunk size, fastest thread count of the sweep)",
        "median_samples_per_s": medians[n_best],
        "unit": "samples/s",
        "cores": threads,
        "host_cpus": ncpu,
        "kind": "reference" if ref_flow is not None else "port",
        "kind_note": "zuko imported from the mounted reference checkout" if ref_flow is not None else "no reference checkout on this host: the oracle port (oracle/zuko_oracle.py)",
        "pinned_bitwise": True,  # the port equals the live reference bit for bit (fixtures regenerated by tests/golden/make_golden.py)
        "thread_sweep_samples_per_s": {str(k): round(v, 1) for k, v in sweep.items()},
        "chunk_sweep_samples_per_s": {f"2^{n.bit_length() - 1}": round(r, 1) for n, r in rates.items()},
        "chunk_sweep_median_samples_per_s": {f"2^{n.bit_length() - 1}": round(r, 1) for n, r in medians.items()},
        "chunk_passes": {f"2^{n.bit_length() - 1}": per_chunk[n][1] for n in per_chunk},
        "extrapolated_seconds_for_2^20": (1 << 20) / rates[n_best],
        "sample": f"fastest of {per_chunk[n_best][1]} passes of a 2^{n_best.bit_length() - 1}-row chunk (best of the 2^12 / 2^14 / 2^16 sweep) at the fastest thread count of the sweep, same model, same rows the GPU parity block uses",
        "cpu_model": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown"),
    }
    return out, {"x": xall, "spec": spec, "out": keep}


def gpu_aten_baseline(flow_cpu, dev, ours: float) -> dict:
    """What a zuko user has on THIS machine without this library: the reference is pure PyTorch and runs unchanged on PyTorch-ROCm, so the
    oracle's `flow_log_prob` body (zuko/distributions.py:115-119; bitwise the reference on the CPU) is timed with its tensors on the GPU —
    ATen-ROCm kernels and library GEMMs, chunks of 2^14 .. 2^17 rows (at 2^18 an ATen kernel of this stack refuses its launch configuration).  Baseline leg only."""
    import torch

    from oracle import zuko_oracle as O

    sd = {k: v.to(dev) for k, v in flow_cpu.state_dict().items() if v is not None}
    spec = O.spec_from_state_dict(sd, "ar", O.uni_rqs(BINS), FEATURES)
    out = {"unit": "samples/s", "what": "the reference's algorithm (oracle/zuko_oracle.py) on ATen-ROCm kernels, same model, fp32", "chunk_sweep_samples_per_s": {}}
    best = 0.0
    with torch.no_grad():
        for lg in (14, 16, 17):
            n = 1 << lg
            x = torch.randn(n, FEATURES, generator=torch.Generator().manual_seed(1)).to(dev)

            def once():
                z, ladj = O.flow_forward(spec, x)
                return O.diag_normal_log_prob(z, spec.loc, spec.scale) + ladj

            try:
                for _ in range(2):
                    once()
                torch.cuda.synchronize()
                reps = 5
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    lp = once()
                e1.record()
                torch.cuda.synchronize()
                rate = n * reps / (e0.elapsed_time(e1) * 1e-3)
                out["chunk_sweep_samples_per_s"][f"2^{lg}"] = round(rate, 1)
                best = max(best, rate)
                del lp
            except Exception as exc:  # (out of memory at the largest chunk: keep what was measured)
                out["chunk_sweep_samples_per_s"][f"2^{lg}"] = repr(exc)[:120]
            del x
            torch.cuda.empty_cache()
    out["value"] = best
    if best > 0:
        out["speedup_vs_gpu_aten"] = ours / best
    return out


# --------------------------------------------------------------------------------------------------
# parity of the GPU path with the oracle, measured in this run on the rows the CPU baseline timed
# --------------------------------------------------------------------------------------------------


def parity_report(flow, x_dev, sample, dev) -> dict:
    """(a) The chunks the CPU baseline timed (2^12, 2^14, 2^16 rows) through the GPU flow as batches of their own;
    (b) 16 slices of 256 rows spread over the FULL benchmark batch (first rows, the middle, the last 256 rows: tail tiles
    of the persistent grid) taken from ONE evaluation of the whole batch, against the oracle on the same rows, in fp32
    (north_star's bar: log_prob within 1e-5 relative) and in float64 (how far each fp32 evaluation is from the exact one)."""
    import torch

    from oracle import zuko_oracle as O

    spec, xall, keep = sample["spec"], sample["x"], sample["out"]
    rep = {"bar": "log_prob max relative error <= 1e-5 (north_star); z / ladj reported as absolute errors next to the fp32 reference's own distance from float64"}
    worst_lp = worst_z = worst_l = 0.0
    rows = 0
    with torch.no_grad():
        chunks = {}
        for n, (z_o, l_o, lp_o) in sorted(keep.items()):
            xg = xall[:n].to(dev)
            dist = flow()
            lp = dist.log_prob(xg).cpu()
            z, ladj = dist.transform.call_and_ladj(xg)
            z, ladj = z.cpu(), ladj.cpu()
            e_lp = float(((lp - lp_o).abs() / lp_o.abs()).max())
            e_z, e_l = float((z - z_o).abs().max()), float((ladj - l_o).abs().max())
            chunks[f"2^{n.bit_length() - 1}"] = {"rows": n, "log_prob_max_rel": e_lp, "z_max_abs": e_z, "ladj_max_abs": e_l}
            worst_lp, worst_z, worst_l = max(worst_lp, e_lp), max(worst_z, e_z), max(worst_l, e_l)
            rows += n
        rep["cpu_baseline_chunks"] = chunks
        # deep-batch slices out of one evaluation of the whole batch
        B = x_dev.shape[0]
        nsl, width = 16, 256
        starts = sorted({min(B - width, (B - width) * i // (nsl - 1)) for i in range(nsl)}) if B >= nsl * width else [0]
        idx = torch.cat([torch.arange(s, min(s + width, B)) for s in starts])
        dist = flow()
        lp_full = dist.log_prob(x_dev)
        z_full, l_full = dist.transform.call_and_ladj(x_dev)
        idx_d = idx.to(dev)
        lp, z, ladj = lp_full[idx_d].cpu(), z_full[idx_d].cpu(), l_full[idx_d].cpu()
        del lp_full, z_full, l_full
        xs = x_dev[idx_d].cpu()
        z_o, l_o = O.flow_forward(spec, xs)
        lp_o = O.diag_normal_log_prob(z_o, spec.loc, spec.scale) + l_o
        e_lp = float(((lp - lp_o).abs() / lp_o.abs()).max())
        e_z, e_l = float((z - z_o).abs().max()), float((ladj - l_o).abs().max())
        deep = {"rows": int(idx.numel()), "slices": len(starts), "slice_rows": width, "first_row_of_last_slice": int(starts[-1]), "batch": int(B),
                "log_prob_max_rel": e_lp, "z_max_abs": e_z, "ladj_max_abs": e_l}
        worst_lp, worst_z, worst_l = max(worst_lp, e_lp), max(worst_z, e_z), max(worst_l, e_l)
        rows += int(idx.numel())
        try:  # float64 oracle on the same slices: error of the HIP path and of the fp32 reference against it
            import dataclasses

            def f64(o):
                if isinstance(o, torch.Tensor):
                    return o.double() if o.is_floating_point() else o
                if dataclasses.is_dataclass(o) and not isinstance(o, type):
                    return type(o)(**{f.name: f64(getattr(o, f.name)) for f in dataclasses.fields(o)})
                if isinstance(o, (list, tuple)):
                    return type(o)(f64(v) for v in o)
                return o

            s64 = f64(spec)
            z64, l64 = O.flow_forward(s64, xs.double())
            lp64 = O.diag_normal_log_prob(z64, s64.loc, s64.scale) + l64
            mx = lambda a, b: float((a.double() - b).abs().max())
            deep["vs_float64_oracle"] = {
                "z_max_abs": {"hip": mx(z, z64), "reference_fp32": mx(z_o, z64)},
                "ladj_max_abs": {"hip": mx(ladj, l64), "reference_fp32": mx(l_o, l64)},
                "log_prob_max_rel": {"hip": float(((lp.double() - lp64).abs() / lp64.abs()).max()), "reference_fp32": float(((lp_o.double() - lp64).abs() / lp64.abs()).max())},
            }
        except Exception as exc:  # the float64 side report must never break the line
            deep["vs_float64_oracle"] = {"error": repr(exc)}
        rep["deep_batch_slices"] = deep
    rep.update(rows=rows, log_prob_max_rel=worst_lp, z_max_abs=worst_z, ladj_max_abs=worst_l, ok=bool(worst_lp <= 1e-5))
    return rep


# --------------------------------------------------------------------------------------------------
# bin-index report of the product kernel at the benchmark batch
# --------------------------------------------------------------------------------------------------


def bin_report(flow, flow_cpu, x, dev) -> dict:
    """First transform of the headline flow at the FULL batch, through the diagnostic twin of the fused kernel
    (same template / arithmetic, asserted bit-identical in tests/test_gpu_bins.py): the index the kernel used vs
    #(its own knots < x) - 1 (must agree everywhere), vs a float64 evaluation of the layer-wise parameters on the
    GPU (all rows), and vs the CPU oracle's index (first 4096 rows)."""
    import torch

    from zuko_amd import ops
    from zuko_amd.nn import MaskedLinear
    from zuko_amd.utils import unpack

    lazy = flow.transform.transforms[0]
    st = lazy.fused_state(dev)
    if st is None:
        return {"error": "first transform does not run on the fused kernel"}
    K = BINS
    N, D = x.shape
    with torch.no_grad():
        st.refresh([m for m in lazy.hyper if isinstance(m, MaskedLinear)])
        y, ladj = torch.empty_like(x), torch.empty(N, device=dev)
        bins = torch.empty(N, D, dtype=torch.int32, device=dev)
        knots = torch.empty(N, D, K + 1, device=dev)
        st.run_diag(x, y, ladj, bins, knots)
        own = ((knots < x.unsqueeze(-1)).sum(-1) - 1).to(torch.int32)
        mism_own = int((own != bins).sum())
        del own
        # float64 evaluation (IEEE exp / division, max-subtracted softmax) of the layer-wise fp32 parameters
        phi = lazy.hyper(x).unflatten(-1, (D, 3 * K - 1))
        w, h, d = unpack(phi.double(), [(K,), (K,), (K - 1,)])
        k64 = ops.rqs_forward(x.double(), w, h, d, want_bins=True)[2]
        flips64 = int((k64 != bins).sum())
        del phi, w, h, d, k64
    out = {
        "elements": N * D,
        "mismatch_vs_own_knots": mism_own,
        "flips_vs_f64_layerwise": flips64,
        "flip_rate_vs_f64_layerwise": flips64 / float(N * D),
    }
    try:
        from oracle import zuko_oracle as O

        sd = {k: v for k, v in flow_cpu.state_dict().items() if v is not None}
        spec = O.spec_from_state_dict(sd, "ar", O.uni_rqs(K), D)
        n = min(N, 4096)
        xc = x[:n].cpu()
        with torch.no_grad():
            phi = O._ar_phi(spec.layers[0], xc, None)
            hor, _, _ = O.rqs_knots(*O.split_packed(phi, spec.layers[0].uni.shapes))
            kref = O.rqs_bin_index(hor, xc)
        kb = bins[:n].cpu().long()
        fl = kb != kref
        out["oracle_sample_elements"] = n * D
        out["flips_vs_oracle_sample"] = int(fl.sum())
        out["max_knot_dev_ulpB_vs_oracle_sample"] = float((knots[:n].cpu().double() - hor.double()).abs().max() / 2.0 ** -21)
    except Exception as exc:  # the report must never break the headline line
        out["oracle_sample_error"] = repr(exc)
    return out


# --------------------------------------------------------------------------------------------------
# side configurations (BASELINE.json configs[2..4]): parity block of a `--config cfgN` run, and the short runs the default
# invocation appends to the headline line
# --------------------------------------------------------------------------------------------------

SIDE_RUNS = {  # config: (batch log2 = its per-GPU share, warm-up, timed steps)
    "cfg3": (20, 3, 20),
    "cfg4": (19, 2, 6),
    "cfg5": (19, 1, 3),
}


def side_parity(config: str, flow, flow_cpu, dev) -> dict:
    """cfg3 / cfg4 (fp32): a 2^12-row chunk through the GPU flow against the CPU oracle (the pinned restatement of the reference) on
    the same rows — north_star's bar, log_prob within 1e-5 relative — with each side's distance from the float64 oracle next to it.
    cfg5 (bf16): SURVEY 9.1's bar on the FULL flow at 1 024 rows — against the fp32 oracle on the same bf16-valued weights the HIP bf16
    path must be no worse than the reference's own bf16 evaluation (the oracle run in torch.bfloat16) in mean / median / p99."""
    import dataclasses

    import torch

    from oracle import zuko_oracle as O

    ctor, kw, _, bf16 = CONFIGS[config]
    D = kw["features"]
    if not bf16:
        kind = "coupling" if ctor in ("RealNVP", "NICE") else "ar"
        uni = O.uni_rqs(kw["bins"]) if ctor == "NSF" else O.UNI_AFFINE
        sd = {k: v for k, v in flow_cpu.state_dict().items() if v is not None}
        spec = O.spec_from_state_dict(sd, kind, uni, D)
        n = 1 << 12
        xs = torch.randn(n, D, generator=torch.Generator().manual_seed(1))
        with torch.no_grad():
            z_o, l_o = O.flow_forward(spec, xs)
            lp_o = O.diag_normal_log_prob(z_o, spec.loc, spec.scale) + l_o
            dist = flow()
            xg = xs.to(dev)
            lp = dist.log_prob(xg).cpu()
            z, ladj = dist.transform.call_and_ladj(xg)
            z, ladj = z.cpu(), ladj.cpu()
        rel = float(((lp - lp_o).abs() / lp_o.abs()).max())
        rep = {"bar": "log_prob max relative error <= 1e-5 against the CPU oracle on the same rows (north_star)", "rows": n, "log_prob_max_rel": rel,
               "z_max_abs": float((z - z_o).abs().max()), "ladj_max_abs": float((ladj - l_o).abs().max()), "ok": bool(rel <= 1e-5)}
        try:
            def f64(o):
                if isinstance(o, torch.Tensor):
                    return o.double() if o.is_floating_point() else o
                if dataclasses.is_dataclass(o) and not isinstance(o, type):
                    return type(o)(**{f.name: f64(getattr(o, f.name)) for f in dataclasses.fields(o)})
                if isinstance(o, (list, tuple)):
                    return type(o)(f64(v) for v in o)
                return o

            s64 = f64(spec)
            with torch.no_grad():
                z64, l64 = O.flow_forward(s64, xs.double())
                lp64 = O.diag_normal_log_prob(z64, s64.loc, s64.scale) + l64
            mx = lambda a, b: float((a.double() - b).abs().max())
            rep["vs_float64_oracle"] = {
                "z_max_abs": {"hip": mx(z, z64), "reference_fp32": mx(z_o, z64)},
                "ladj_max_abs": {"hip": mx(ladj, l64), "reference_fp32": mx(l_o, l64)},
                "log_prob_max_rel": {"hip": float(((lp.double() - lp64).abs() / lp64.abs()).max()), "reference_fp32": float(((lp_o.double() - lp64).abs() / lp64.abs()).max())},
            }
        except Exception as exc:
            rep["vs_float64_oracle"] = {"error": repr(exc)}
        return rep
    # bf16
    K, n = kw["bins"], 1024  # (64 rows until round 5; the 16-thread oracle handles 1 024 in ~20 s)
    sdb = {k: v.detach().cpu() for k, v in flow.state_dict().items() if v is not None}
    xs = torch.randn(n, D, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16)
    with torch.no_grad():
        spec_b = O.spec_from_state_dict(sdb, "ar", O.uni_rqs(K), D)
        zb, lb = O.flow_forward(spec_b, xs)
        lpb = O.diag_normal_log_prob(zb, spec_b.loc, spec_b.scale) + lb  # (= O.flow_log_prob(spec_b, xs) without a second pass over the 630 M parameters on the CPU)
        del spec_b
        spec_32 = O.spec_from_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in sdb.items()}, "ar", O.uni_rqs(K), D)
        z32, l32 = O.flow_forward(spec_32, xs.float())
        lp32 = O.diag_normal_log_prob(z32, spec_32.loc, spec_32.scale) + l32
        del spec_32, sdb
        dist = flow()
        lp = dist.log_prob(xs.to(dev)).cpu()
        z, ladj = dist.transform.call_and_ladj(xs.to(dev))
        z, ladj = z.float().cpu(), ladj.float().cpu()

    def stats(e):
        e = e.double().flatten()
        return {"mean": float(e.mean()), "median": float(e.median()), "p99": float(torch.quantile(e, 0.99)), "max": float(e.max())}

    rep = {"bar": "bf16 (SURVEY 9.1): |hip_bf16 - fp32 oracle| <= |reference's own bf16 evaluation - fp32 oracle| in mean, median and p99 of z, ladj and log_prob "
                  "(max within 2x), same bf16-valued weights and rows", "rows": n, "ok": True}
    for what, hip, ref, gold in (("z", z, zb.float(), z32), ("ladj", ladj, lb.float(), l32), ("log_prob", lp.float(), lpb.float(), lp32)):
        e_hip, e_ref = stats((hip - gold).abs()), stats((ref - gold).abs())
        good = e_hip["mean"] <= e_ref["mean"] and e_hip["median"] <= e_ref["median"] + 1e-12 and e_hip["p99"] <= e_ref["p99"] and e_hip["max"] <= 2.0 * e_ref["max"] + 1e-6
        rep[what] = {"hip_bf16_abs_err": e_hip, "reference_bf16_abs_err": e_ref, "ok": bool(good)}
        rep["ok"] = bool(rep["ok"] and good)
    rep["log_prob_max_rel_vs_fp32_oracle"] = float(((lp.float() - lp32).abs() / lp32.abs()).max())
    # an ABSOLUTE bar against the fp32 oracle as well ("no worse than the reference's own bf16" is met 9x / 150x over): z within
    # half a bf16 ulp of [2, 4) on average after 12 transforms, ladj (a sum of 1024 terms per transform, f32 accumulation) within 0.1 on average,
    # log_prob within 1.5e-3 relative everywhere — about twice what the path measures (tests/test_gpu_flows.py::test_cfg5_full_flow_bf16 holds the same)
    ab = {"z_mean_abs_err": (rep["z"]["hip_bf16_abs_err"]["mean"], 2.0 ** -7), "z_max_abs_err": (rep["z"]["hip_bf16_abs_err"]["max"], 0.15),
          "ladj_mean_abs_err": (rep["ladj"]["hip_bf16_abs_err"]["mean"], 0.1), "ladj_max_abs_err": (rep["ladj"]["hip_bf16_abs_err"]["max"], 0.35),
          "log_prob_max_rel": (rep["log_prob_max_rel_vs_fp32_oracle"], 1.5e-3)}
    rep["absolute_bar_vs_fp32_oracle"] = {k: {"value": v, "bar": bar, "ok": bool(v <= bar)} for k, (v, bar) in ab.items()}
    rep["ok"] = bool(rep["ok"] and all(v["ok"] for v in rep["absolute_bar_vs_fp32_oracle"].values()))
    return rep


def run_side_configs(args) -> dict:
    """cfg3 / cfg4 / cfg5 for a few steps each, every one as `python bench.py --config cfgN ...` in a process of its own (own
    allocator, own plans; a failure costs that entry only), AFTER the headline's timed region.  Returns {cfgN: condensed line}."""
    out = {}
    for cfg, (blog, warm, steps) in SIDE_RUNS.items():
        steps = args.side_steps or steps
        cmd = [sys.executable, os.path.abspath(__file__), "--config", cfg, "--gpus", "1", "--batch-log2", str(blog), "--warmup", str(warm), "--steps", str(steps),
               "--no-bin-report", "--no-side-configs", "--full-line"]
        t0 = time.perf_counter()
        try:
            env = dict(os.environ)
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
            line = next((l for l in reversed(res.stdout.splitlines()) if l.startswith("{")), None)
            if res.returncode != 0 or line is None:
                out[cfg] = {"error": f"exit code {res.returncode}", "stderr_tail": res.stderr[-600:]}
                continue
            full = json.loads(line)
            roof = full.get("roofline") or {}
            out[cfg] = {
                "workload": full["config"]["workload"],
                "value": full["value"], "unit": full["unit"], "ms_per_step": full["ms_per_step"], "steps": full["steps"], "warmup": full["warmup"], "dtype": full["dtype"],
                "batch_per_gpu": full["config"]["batch_per_gpu"],
                "roofline": {k: roof.get(k) for k in ("kernel", "avg_launch_ms", "bound", "achieved", "peak", "unit", "frac", "instantiation", "note") if roof.get(k) is not None},
                "parity": full.get("parity"),
                "kernels": [{k: r.get(k) for k in ("kernel", "calls", "avg_ms", "frac") if r.get(k) is not None} for r in full.get("kernels", [])],
                "wall_s": round(time.perf_counter() - t0, 1),
            }
        except Exception as exc:  # never let a side measurement break the headline line
            out[cfg] = {"error": repr(exc)}
    return out


def side_paths_report() -> dict:
    """The paths either side of log_prob for the cfg2 / cfg3 flows: one Adam step of the README training loop at batch 2^16 (with the
    gradients checked against float64 autograd through the oracle on 4 096 of the rows) and flow().transform.inv at 2^18 (with the round trip through
    the forward).  Short, after everything else; the headline's timed region never sees it."""
    import torch

    from zuko_amd import flows as F

    # the oracle legs of this report run on the HOST: PyTorch-CPU does not scale to the box's 256 hardware threads on these sizes (cpu_baseline's
    # sweep: 16 threads are fastest, all of them hundreds of times slower) — measured in round 5: this report took 331 s at the default, most of it here
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    out = {}
    cfg1 = dict(features=3, context=5, transforms=3, bins=8, hidden_features=[128] * 3)  # BASELINE.json configs[0]: the conditional flow
    for name, ctor, kw in (("nsf_cfg2", "NSF", CONFIGS["cfg2"][1]), ("maf_cfg3", "MAF", CONFIGS["cfg3"][1]), ("nsf_cfg1_conditional", "NSF", cfg1), ("realnvp_cfg4", "RealNVP", CONFIGS["cfg4"][1])):
        entry = {}
        t_entry = time.perf_counter()
        try:
            torch.manual_seed(0)
            flow = getattr(F, ctor)(**kw).to(dev)
            coupling = ctor == "RealNVP"  # (cfg4 trains layer by layer — zuko_amd/flows/coupling.py — there is no one-node path for coupling transforms)
            B = 1 << 14 if coupling else 1 << 16
            x = torch.randn(B, kw["features"], device=dev)
            ctx = torch.randn(1 << 18, kw["context"], device=dev) if kw.get("context") else None

            def dist(rows):  # flow(c) on the first `rows` context rows / flow() without a context
                return flow() if ctx is None else flow(ctx[:rows])

            opt = torch.optim.Adam(flow.parameters(), lr=1e-3)

            def step():
                loss = -dist(B).log_prob(x).mean()
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
                return loss

            def nodes(fn, seen):
                if fn is not None and fn not in seen:
                    seen.add(fn)
                    for nxt, _ in fn.next_functions:
                        nodes(nxt, seen)
                return seen

            def grads(rows):
                flow.zero_grad()
                loss = -dist(rows).log_prob(x[:rows]).mean()
                names = {type(f).__name__ for f in nodes(loss.grad_fn, set())}
                loss.backward()
                return loss.item(), names, [p.grad.clone() for p in flow.parameters()]

            prows = 8192 if coupling else 4096  # (RealNVP cfg4 at 1 024 rows: 1-norm distances between 7e-7 and 3e-3 for BOTH training paths depending on the data seed, 5e-4 .. 1.9e-3 at 4 096 — profiles/r06/cfg4_grad_noise.txt — which units flip is a coin toss: more rows, less of it)
            l1, names, g1 = grads(prows)
            # the yardstick is autograd through the ORACLE (the reference's algorithm on PyTorch-CPU ops, float64) on the same 4 096 rows and weights —
            # which is how the reference itself obtains its gradients (tests/test_flows.py:22-29); until round 4 this block compared two HIP paths
            from oracle import zuko_oracle as O

            sd = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for k, v in flow.state_dict().items() if v is not None}
            pnames = [k for k, _ in flow.named_parameters()]
            leaves = {k: sd[k].requires_grad_() for k in pnames}
            sd.update(leaves)
            spec = O.spec_from_state_dict(sd, "coupling" if coupling else "ar", O.uni_rqs(kw["bins"]) if ctor == "NSF" else O.UNI_AFFINE, kw["features"])
            l2t = -O.flow_log_prob(spec, x[:prows].cpu().double(), None if ctx is None else ctx[:prows].cpu().double()).mean()
            l2t.backward()
            l2 = float(l2t.detach())
            g2 = [leaves[k].grad for k in pnames]
            rel = max(((a.cpu().double() - b).abs().max() / b.abs().max().clamp_min(1e-12)).item() for a, b in zip(g1, g2))
            rel1 = max(((a.cpu().double() - b).abs().sum() / b.abs().sum().clamp_min(1e-12)).item() for a, b in zip(g1, g2))
            del spec, leaves, sd
            for _ in range(3):
                l0 = step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                l = step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            entry["training"] = {"workload": f"{ctor} Adam step of -log_prob(x).mean(), batch 2^{B.bit_length() - 1}", "ms_per_step": dt * 1e3, "samples_per_s": B / dt, "loss_before_after": [float(l0), float(l)],
                                 "one_autograd_node_per_transform": ("CouplingFnBackward" if coupling else "AutoregressiveFnBackward") in names,
                                 "parity": {"rows": prows, "against": "float64 autograd through the oracle (oracle/zuko_oracle.py) on the same rows and weights",
                                            "grad_l1_rel_vs_oracle_autograd": rel1, "grad_max_rel_vs_oracle_autograd": rel, "loss_abs_diff": abs(l1 - l2),
                                            "bar": "1-norm distance per parameter tensor < 2e-3 (tests/test_gpu_backward.py::test_gradients_over_many_tiles: the max-norm moves by O(1 / rows) per "
                                                   "hidden unit whose pre-activation lies within float32 rounding of zero — profiles/r05/grad_error_probe.txt; the same test holds one launch over "
                                                   "4096 rows to 1e-5 of 32 per-tile launches)",
                                            "max_norm_gate": "grad_max_rel < 5e-2 as well (a single mis-scattered row of a large tensor moves the 1-norm by 1 / rows: the max norm catches it)",
                                            "ok": bool(rel1 < 2e-3 and rel < 5e-2 and abs(l1 - l2) < 1e-4 * max(1.0, abs(l2)))}}
            del opt
            with torch.no_grad():
                Bs = 1 << 16 if coupling else (1 << 20 if name in ("nsf_cfg2", "maf_cfg3") else 1 << 18)  # (BASELINE.json configs[2]: MAF log_prob + inverse at batch 2^20)
                z = torch.randn(Bs, kw["features"], device=dev)
                t = dist(Bs).transform
                xs = t.inv(z)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    xs = t.inv(z)
                torch.cuda.synchronize()
                ds = (time.perf_counter() - t0) / 3
                back = t(xs)
                err = (back - z).abs().max().item()
            entry["sampling"] = {"workload": f"{ctor} flow().transform.inv(z), batch 2^{Bs.bit_length() - 1}", "batch_log2": Bs.bit_length() - 1, "ms": ds * 1e3, "samples_per_s": Bs / ds, "round_trip_max_abs": err,
                                 "ok": bool(err < 1e-3)}
            if name == "nsf_cfg1_conditional":  # BASELINE.json configs[0] at ITS batch: what a call costs when the launch, not the arithmetic, is the time
                from zuko_amd import _C

                with torch.no_grad():
                    xb = x[:4096]
                    for _ in range(3):
                        lp = dist(4096).log_prob(xb)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(50):
                        lp = dist(4096).log_prob(xb)
                    torch.cuda.synchronize()
                    dl = (time.perf_counter() - t0) / 50
                    _C.PROFILE = {}
                    lp = dist(4096).log_prob(xb)
                    torch.cuda.synchronize()
                    prof, _C.PROFILE = _C.PROFILE, None
                entry["log_prob_batch_4096"] = {"workload": "NSF(3, 5, transforms=3, hidden=[128]*3) flow(c).log_prob(x), batch 4096 (BASELINE.json configs[0])", "ms": dl * 1e3, "samples_per_s": 4096 / dl,
                                                "library_calls_per_log_prob": {k: len(v) for k, v in prof.items()}}
                # the same call captured in a HIP graph (the library launches on torch's current stream, so torch.cuda.graph captures it): what is left
                # when the Python / ctypes / launch overhead of the four calls is paid once
                try:
                    with torch.no_grad():
                        cstat = ctx[:4096].clone()
                        s_ = torch.cuda.Stream()
                        s_.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(s_):
                            for _ in range(3):
                                lp_g = flow(cstat).log_prob(xb)
                        torch.cuda.current_stream().wait_stream(s_)
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph):
                            lp_g = flow(cstat).log_prob(xb)
                        graph.replay()
                        torch.cuda.synchronize()
                        same = bool(torch.equal(lp_g, lp))
                        t0 = time.perf_counter()
                        for _ in range(200):
                            graph.replay()
                        torch.cuda.synchronize()
                        dg = (time.perf_counter() - t0) / 200
                    entry["log_prob_batch_4096"]["hip_graph_replay"] = {"ms": dg * 1e3, "samples_per_s": 4096 / dg, "bitwise_equal_to_eager": same}
                except Exception as exc:
                    entry["log_prob_batch_4096"]["hip_graph_replay"] = {"error": repr(exc)[:200]}
                # one optimisation step at that batch, eager and as a replayed HIP graph (zuko_amd.capture_step) on a COPY of the flow
                try:
                    import copy

                    import zuko_amd

                    f2 = copy.deepcopy(flow)
                    o2 = torch.optim.Adam(f2.parameters(), lr=1e-3, capturable=True)
                    xs4, cs4 = x[:4096].clone(), ctx[:4096].clone()

                    def eager_step():
                        loss = -f2(cs4).log_prob(xs4).mean()
                        o2.zero_grad(set_to_none=False)
                        loss.backward()
                        o2.step()

                    for _ in range(3):
                        eager_step()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(20):
                        eager_step()
                    torch.cuda.synchronize()
                    de = (time.perf_counter() - t0) / 20
                    stepg = zuko_amd.capture_step(f2, o2, xs4, cs4)
                    stepg()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(50):
                        stepg()
                    torch.cuda.synchronize()
                    dgs = (time.perf_counter() - t0) / 50
                    entry["training_step_batch_4096"] = {"workload": "Adam(capturable) step of -flow(c).log_prob(x).mean(), batch 4096", "eager_ms": de * 1e3, "hip_graph_replay_ms": dgs * 1e3,
                                                         "loss_finite": bool(torch.isfinite(stepg.loss).item())}
                except Exception as exc:
                    entry["training_step_batch_4096"] = {"error": repr(exc)[:200]}
        except Exception as exc:  # never let a side measurement break the headline line
            entry["error"] = repr(exc)
        entry["wall_s"] = round(time.perf_counter() - t_entry, 1)
        out[name] = entry
    # the polynomial flows: fused layer kernels against the layer-wise path on the same weights
    from zuko_amd.flows import autoregressive as AR

    for name, ctor in (("sospf_d64", "SOSPF"), ("bpf_d64", "BPF")):
        entry = {}
        t_entry = time.perf_counter()
        try:
            torch.manual_seed(0)
            flow = getattr(F, ctor)(64, 0, transforms=3, hidden_features=[256] * 3).to(dev)
            Bp = 1 << 18
            x = torch.randn(Bp, 64, device=dev)
            res = {}
            for mode in ("fused", "layer_wise"):
                orig = AR.FusedAutoregressiveTransform._fused
                if mode == "layer_wise":
                    AR.FusedAutoregressiveTransform._fused = lambda self, x, need_generic=False: None
                try:
                    with torch.no_grad():
                        for _ in range(2):
                            lp = flow().log_prob(x)
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(5):
                            lp = flow().log_prob(x)
                        torch.cuda.synchronize()
                        res[mode] = ((time.perf_counter() - t0) / 5, lp)
                finally:
                    AR.FusedAutoregressiveTransform._fused = orig
            rel = ((res["fused"][1] - res["layer_wise"][1]).abs().max() / res["layer_wise"][1].abs().max()).item()
            entry["log_prob"] = {"workload": f"{ctor}(64, transforms=3, hidden=[256]*3) log_prob, batch 2^18", "ms": res["fused"][0] * 1e3, "samples_per_s": Bp / res["fused"][0],
                                 "layer_wise_ms": res["layer_wise"][0] * 1e3, "parity": {"log_prob_max_rel_vs_layer_wise_kernels": rel, "ok": bool(rel < 1e-5)}}
            # an Adam step at 2^16 rows on a COPY of the flow (round 6: the polynomial maps' adjoints are reverse-mode kernels written out; until then 17-wide
            # forward-mode duals: SOSPF 27 ms, BPF 118 ms per step).  Gradients: tests/test_gpu_backward.py against float64 autograd and the dual-number kernels.
            try:
                import copy

                f2 = copy.deepcopy(flow)
                o2 = torch.optim.Adam(f2.parameters(), lr=1e-3)
                xt = 0.8 * torch.randn(1 << 16, 64, device=dev)

                def tstep():
                    loss = -f2().log_prob(xt).mean()
                    o2.zero_grad(set_to_none=True)
                    loss.backward()
                    o2.step()
                    return loss

                for _ in range(3):
                    l0 = tstep()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    l1 = tstep()
                torch.cuda.synchronize()
                dtr = (time.perf_counter() - t0) / 5
                entry["training"] = {"workload": f"{ctor}(64, transforms=3, hidden=[256]*3) Adam step of -log_prob(x).mean(), batch 2^16", "ms_per_step": dtr * 1e3, "samples_per_s": (1 << 16) / dtr,
                                     "loss_before_after": [float(l0), float(l1)], "one_autograd_node_per_transform": False,
                                     "parity": {"ok": bool(torch.isfinite(l1).item()), "note": "finite loss only; gradient values: tests/test_gpu_backward.py"}}
                del f2, o2, xt
            except Exception as exc:
                entry["training"] = {"error": repr(exc)[:200]}
            # sampling (round 6): ONE incremental launch per autoregressive layer with the reference's bisection (zuko/transforms.py:608-617) in the kernel's group
            # epilogue — at 2^18, and at 2^14 beside what it replaces: the layer-wise wavefront form (ZUKO_AMD_NO_INCREMENTAL=1: per sweep the hidden layers, the last
            # layer's rows and the bisections of that sweep's feature) and the reference's loop itself (ZUKO_AMD_FULL_SWEEPS=1), which must agree bit for bit
            with torch.no_grad():
                tr = flow().transform
                samp = {}
                for lg, modes in ((18, ("incremental",)), (14, ("incremental", "wavefront", "reference_loop"))):
                    Bs = 1 << lg
                    x0 = 0.8 * torch.randn(Bs, 64, device=dev)
                    z = tr(x0)  # (inside the maps' invertible range: the bisection works on [-B, B], as the reference's)
                    for mode in modes:
                        env = {"wavefront": {"ZUKO_AMD_NO_INCREMENTAL": "1"}, "reference_loop": {"ZUKO_AMD_NO_INCREMENTAL": "1", "ZUKO_AMD_FULL_SWEEPS": "1"}}.get(mode, {})
                        keep = {k: os.environ.get(k) for k in env}
                        os.environ.update(env)
                        try:
                            xs = tr.inv(z)
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                            xs = tr.inv(z)
                            torch.cuda.synchronize()
                            samp[(lg, mode)] = (time.perf_counter() - t0, xs, (xs - x0).abs().max().item())
                        finally:
                            for k, v in keep.items():
                                if v is None:
                                    os.environ.pop(k, None)
                                else:
                                    os.environ[k] = v
            t18, t14 = samp[(18, "incremental")], samp[(14, "incremental")]
            entry["sampling"] = {"workload": f"{ctor}(64, transforms=3, hidden=[256]*3) flow().transform.inv(z), batch 2^18: one incremental launch per layer, bisection in the kernel", "batch_log2": 18,
                                 "ms": t18[0] * 1e3, "samples_per_s": (1 << 18) / t18[0], "round_trip_max_abs": t18[2], "ok": bool(t18[2] < 1e-3),
                                 "at_2^14": {"incremental_ms": t14[0] * 1e3, "layer_wise_wavefront_ms": samp[(14, "wavefront")][0] * 1e3, "reference_loop_ms": samp[(14, "reference_loop")][0] * 1e3,
                                             "incremental_max_abs_diff_vs_reference_loop": (t14[1] - samp[(14, "reference_loop")][1]).abs().max().item()},
                                 "bitwise_equal_to_reference_loop": bool(torch.equal(samp[(14, "wavefront")][1], samp[(14, "reference_loop")][1]))}
        except Exception as exc:
            entry["error"] = repr(exc)
        entry["wall_s"] = round(time.perf_counter() - t_entry, 1)
        out[name] = entry
    t_entry = time.perf_counter()
    out["generic_split_kernel"] = generic_split_report(dev)
    out["generic_split_kernel"]["wall_s"] = round(time.perf_counter() - t_entry, 1)
    return out


def generic_split_report(dev) -> dict:
    """The headline flow with NO kernel generated for its conditioner (a box without hipcc, a shape that was not prebuilt): zk_ar_forward_split
    (csrc/fused_ar_gsplit.hip: run-time skip tests around the operand-split arithmetic) beside the generated kernels of the headline line and the
    generic f32-instruction kernel it replaces.  Same weights, same rows; log_prob must be bit-identical to the generated kernels'."""
    import torch

    from zuko_amd import flows as F
    from zuko_amd.flows import autoregressive as AR

    entry = {"workload": CONFIGS["cfg2"][2] if len(CONFIGS["cfg2"]) > 2 else "NSF cfg2 log_prob, batch 2^20"}
    try:
        torch.manual_seed(0)
        flow = F.NSF(**CONFIGS["cfg2"][1]).to(dev)
        B = 1 << 20
        x = torch.randn(B, 64, device=dev)
        res = {}
        import zuko_amd

        keep_precision = zuko_amd.matmul_precision()
        for mode, env, precision in (("two_part_kernels", {}, "f16x2"), ("generated_split_kernels", {}, "bf16x3"), ("generic_split_kernel", {"ZUKO_AMD_NO_STATIC_AR": "1"}, "bf16x3"),
                                     ("generic_f32_kernel", {"ZUKO_AMD_NO_STATIC_AR": "1", "ZUKO_AMD_GSPLIT": "0"}, "bf16x3")):
            keep = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            zuko_amd.set_matmul_precision(precision)
            try:
                for lazy in flow.transform.transforms:
                    AR._FUSED_CACHE.pop(lazy, None)
                with torch.no_grad():
                    for _ in range(2):
                        lp = flow().log_prob(x)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(5):
                        lp = flow().log_prob(x)
                    torch.cuda.synchronize()
                    res[mode] = ((time.perf_counter() - t0) / 5, lp)
            finally:
                zuko_amd.set_matmul_precision(keep_precision)
                for k, v in keep.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
                for lazy in flow.transform.transforms:
                    AR._FUSED_CACHE.pop(lazy, None)
        for mode, (dt, _) in res.items():
            entry[mode] = {"ms": dt * 1e3, "samples_per_s": B / dt}
        entry["two_part_max_rel_diff_vs_three_part"] = ((res["two_part_kernels"][1] - res["generated_split_kernels"][1]).abs().max() / res["generated_split_kernels"][1].abs().max()).item()
        entry["generic_split_over_generated"] = res["generic_split_kernel"][0] / res["generated_split_kernels"][0]
        entry["generic_split_bitwise_equal_to_generated"] = bool(torch.equal(res["generic_split_kernel"][1], res["generated_split_kernels"][1]))
        entry["generic_f32_max_rel_diff"] = ((res["generic_f32_kernel"][1] - res["generated_split_kernels"][1]).abs().max() / res["generated_split_kernels"][1].abs().max()).item()
    except Exception as exc:
        entry["error"] = repr(exc)
    return entry


def run_side_paths() -> dict:
    cmd = [sys.executable, os.path.abspath(__file__), "--side-paths"]
    try:
        env = dict(os.environ)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        line = next((l for l in reversed(res.stdout.splitlines()) if l.startswith("{")), None)
        if res.returncode != 0 or line is None:
            return {"error": f"exit code {res.returncode}", "stderr_tail": res.stderr[-600:]}
        return json.loads(line)
    except Exception as exc:
        return {"error": repr(exc)}


def model_flops(flow) -> dict:
    """Per sample, whole flow: dense conditioner FLOPs (SURVEY 8d column 2) and FLOPs on non-zero (unmasked) weights."""
    dense = nnz = 0
    for t in flow.transform.transforms:
        hyper = getattr(t, "hyper", None)
        if hyper is None:
            continue
        for m in hyper.modules():
            w = getattr(m, "weight", None)
            if w is None or w.dim() != 2:
                continue
            dense += 2 * w.numel()
            mask = getattr(m, "mask", None)
            nnz += 2 * (int(mask.sum()) if mask is not None and mask.shape == w.shape else w.numel())
    return {"dense": dense, "nnz": nnz}


# --------------------------------------------------------------------------------------------------
# the printed line: a compact view (<= LINE_BUDGET bytes) of the full object, which goes to a file
# --------------------------------------------------------------------------------------------------

LINE_BUDGET = 8192  # bytes: round 5's 20 KB line was not picked up by the driver (BENCH_r05.json "parsed": null)
DETAIL_PATH = os.path.join("gpurun_out", "bench_detail.json")  # relative to the repo root; copied to profiles/rNN/ by the round scripts


def _sig(v, digits: int = 6):
    """Floats to `digits` significant digits, recursively (the line is for reading and parsing, the detail file keeps everything)."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.{digits}g}")
    if isinstance(v, dict):
        return {k: _sig(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, digits) for x in v]
    return v


def _pick(d, *keys):
    d = d or {}
    return {k: d[k] for k in keys if d.get(k) is not None}


def write_detail(out: dict):
    """The full object (per-kernel table, sweeps, float64 comparisons, wall-clock split, every side entry in full) as a file."""
    path = os.path.join(ROOT, DETAIL_PATH)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        return DETAIL_PATH
    except OSError as exc:  # read-only checkout: the line still prints
        return f"not written: {exc!r}"[:120]


def headline_line(out: dict) -> dict:
    """What bench.py prints: the contract's keys, `roofline`, `cpu_baseline`, the in-run parity verdicts and ONE compact entry per side
    configuration / side path.  Everything else lives in the detail file.  Guaranteed to serialise to <= LINE_BUDGET bytes: optional
    groups are dropped, least important first, should a run ever exceed it (tests/test_bench_line.py)."""
    line = _pick(out, "metric", "value", "unit", "n_gpus", "rccl_world_size", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling")
    line["vs_baseline"] = out.get("vs_baseline")
    line.update(_pick(out, "dtype", "data"))
    if out.get("dtype_note"):
        line["dtype_note"] = ("f32 in/out/accumulate; products as 2-way f16 operand splits scaled by exact powers of two, 3 partial products on the f16 matrix instruction (f32 dot-product error: see parity)"
                              if "TWO f16" in out["dtype_note"] else "f32 in/out/accumulate; products as 3-way bf16 operand splits on the bf16 matrix instruction (f32 dot-product rounding: see parity)")
    line.update(_pick(out, "matmul_precision"))
    line["config"] = _pick(out.get("config"), "workload", "batch_per_gpu", "global_batch", "parallelism")
    roof = out.get("roofline")
    line["roofline"] = None if roof is None else {
        **_pick(roof, "bound", "achieved", "peak", "unit", "frac"), "traffic": roof.get("traffic"),
        **_pick(roof, "kernel", "avg_launch_ms", "algorithmic_flop_per_launch", "executed_flop_per_launch", "frac_executed", "peak_basis", "f32_instruction_peak",
                "frac_on_the_six_product_basis_of_rounds_3_to_5", "traffic_source")}
    if out.get("cpu_baseline"):
        line["cpu_baseline"] = _pick(out["cpu_baseline"], "value", "unit", "cores", "kind", "sample", "median_samples_per_s", "host_cpus", "cpu_model", "pinned_bitwise")
        line.update(_pick(out, "speedup_vs_cpu_baseline"))
    if out.get("parity"):
        line["parity"] = _pick(out["parity"], "ok", "rows", "log_prob_max_rel", "z_max_abs", "ladj_max_abs", "error")
        line["parity"]["bar"] = "log_prob max rel <= 1e-5 vs the oracle, fp32"
    if out.get("bin_index"):
        line["bin_index"] = _pick(out["bin_index"], "elements", "mismatch_vs_own_knots", "flips_vs_oracle_sample", "oracle_sample_elements", "error")
    if out.get("gpu_aten_baseline"):
        line["gpu_aten_baseline"] = _pick(out["gpu_aten_baseline"], "value", "unit", "error")
        line.update(_pick(out, "speedup_vs_gpu_aten"))
    line.update(_pick(out, "nll", "per_rank_ms_per_step", "rank0_alone_before_group", "weak_scaling_efficiency"))
    if out.get("cpu_affinity"):
        line["cpu_affinity"] = _pick(out["cpu_affinity"], "bound", "source", "cpus", "numa_node")
    side = {}
    for cfg, e in (out.get("side_configs") or {}).items():
        if "error" in e:
            side[cfg] = {"error": str(e["error"])[:120]}
            continue
        r, p = e.get("roofline") or {}, e.get("parity") or {}
        side[cfg] = {**_pick(e, "value", "ms_per_step", "batch_per_gpu", "dtype"), **_pick(r, "frac", "kernel", "avg_launch_ms"),
                     "parity_ok": p.get("ok"), **({"parity_rows": p["rows"]} if p.get("rows") else {})}
    if side:
        line["side_configs"] = side
    paths = {}
    for name, e in (out.get("side_paths") or {}).items():
        if not isinstance(e, dict):
            continue
        if "error" in e:
            paths[name] = {"error": str(e["error"])[:120]}
            continue
        c = {}
        tr, sa, lp = e.get("training"), e.get("sampling"), e.get("log_prob")
        if tr:
            c["train"] = {**_pick(tr, "ms_per_step", "samples_per_s", "one_autograd_node_per_transform"), "ok": (tr.get("parity") or {}).get("ok")}
        if sa:
            c["sample"] = {**_pick(sa, "samples_per_s", "ms", "batch_log2", "ok", "bitwise_equal_to_reference_loop")}
        if lp:
            c["log_prob"] = {**_pick(lp, "samples_per_s", "ms"), "ok": (lp.get("parity") or {}).get("ok")}
        if e.get("log_prob_batch_4096"):
            b = e["log_prob_batch_4096"]
            c["log_prob_4096"] = {**_pick(b, "ms"), **{"graph_" + k: v for k, v in _pick(b.get("hip_graph_replay"), "ms", "bitwise_equal_to_eager").items()},
                                  **{"one_launch_" + k: v for k, v in _pick(b.get("one_launch"), "ms", "bitwise_equal_to_eager").items()}}
        if isinstance(e.get("training_step_batch_4096"), dict):
            c["train_step_4096"] = _pick(e["training_step_batch_4096"], "eager_ms", "hip_graph_replay_ms")
        for k in ("two_part_kernels", "generated_split_kernels", "generic_split_kernel", "generic_f32_kernel"):
            if isinstance(e.get(k), dict):
                c[k + "_ms"] = e[k].get("ms")
        c.update(_pick(e, "generic_split_bitwise_equal_to_generated", "two_part_max_rel_diff_vs_three_part"))
        paths[name] = c
    if paths:
        line["side_paths"] = paths
    if out.get("wall_s"):
        line["wall_s"] = _pick(out["wall_s"], "total")
    line["detail"] = out.get("detail")
    line = _sig(line)
    # the guard: never print more than the budget
    for drop in ("cpu_affinity", "gpu_aten_baseline", "side_paths", "side_configs", "dtype_note", "bin_index", "per_rank_ms_per_step"):
        if len(json.dumps(line)) <= LINE_BUDGET - 256:
            break
        if drop in line:
            line[drop] = {"dropped": "line budget", "see": out.get("detail")}
    return line


def main() -> None:
    args = parse()
    if args.side_paths:
        print(json.dumps(side_paths_report()))
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(relaunch_under_torchrun(args.gpus))

    import torch

    if args.config != "cfg2":  # side configs: model construction and the CPU-oracle parity leg (cfg5: 629 M parameters) at a thread count PyTorch-CPU scales to
        torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    rank, world, local, dist = init_ranks(args)
    if os.environ.get("ZUKO_BENCH_LAUNCH_SELFTEST") == "1":
        got = 1
        aff = bind_cpu_to_gpu(int(os.environ.get("LOCAL_RANK", 0)), world)  # (no GPU here: the even-slice branch)
        bound = torch.tensor([1.0 if aff.get("bound") else 0.0, float(aff.get("first_cpu", -1))])
        firsts = [bound.clone() for _ in range(world)]
        if dist is not None:
            t = torch.ones(1)
            dist.all_reduce(t)
            got = int(t.item())
            dist.all_gather(firsts, bound)
        if rank == 0:
            print(json.dumps({"selftest": True, "n_gpus": world, "rccl_world_size": dist.get_world_size() if dist else 1, "allreduce_of_ones": got,
                              "ranks_bound": int(sum(f[0].item() for f in firsts)), "first_cpu_per_rank": [int(f[1].item()) for f in firsts]}))
        if dist is not None:
            dist.destroy_process_group()
        return
    dev = torch.device("cuda", local)

    import zuko_amd
    import zuko_amd.flows as ZF
    from zuko_amd import _C, ops

    ctor, kw, workload, bf16 = CONFIGS[args.config]
    make = lambda: getattr(ZF, ctor)(**kw)
    features, transforms = kw["features"], kw["transforms"]
    torch.manual_seed(0)
    flow_cpu = make()
    flops = model_flops(flow_cpu)
    if bf16:  # 629 M parameters: no second copy
        flow, flow_cpu = flow_cpu.to(dev).to(torch.bfloat16), None
    else:
        flow = make()
        flow.load_state_dict(flow_cpu.state_dict())
        flow = flow.to(dev)
    B = 1 << args.batch_log2
    x = torch.randn(B, features, generator=torch.Generator().manual_seed(1 + rank)).to(dev)
    if bf16:
        x = x.to(torch.bfloat16)

    def step(collective: bool = True):
        with torch.no_grad():
            lp = flow().log_prob(x)
            nll = ops.sum_f64(lp, -1.0 / (B * world))
            if dist is not None and collective:
                dist.all_reduce(nll)  # the only collective: one f64 scalar over RCCL/xGMI
        return nll

    affinity = bind_cpu_to_gpu(local, world) if torch.cuda.device_count() >= world else {"bound": False, "source": "single-device dry run"}
    solo = None
    if dist is not None and hasattr(dist, "_zuko_form_group"):
        if rank == 0:  # the SAME per-GPU workload on rank 0 alone, before the group exists: what N-GPU weak scaling is measured against
            k = args.steps  # same warm-up and step counts as the group run (a colder, shorter solo run would flatter the efficiency)
            for _ in range(args.warmup):
                step(collective=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                step(collective=False)
            torch.cuda.synchronize()
            solo = {"steps": k, "ms_per_step": (time.perf_counter() - t0) / k * 1e3}
            solo["value"] = B / (solo["ms_per_step"] * 1e-3)
        dist._zuko_form_group()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        nll = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nll = step()
    fence()
    dt_local = time.perf_counter() - t0
    dt = dt_local
    per_rank_ms = [dt_local / args.steps * 1e3]
    if dist is not None:
        mine = torch.tensor([dt_local], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        dt = max(float(t.item()) for t in every)  # MAX over ranks
    ms = dt / args.steps * 1e3
    value = B * world / (dt / args.steps)
    nll_value = float(nll.item())  # the all-reduced mean NLL of the last timed step

    # per-kernel durations over extra (profiled) steps: events on the launch stream
    roof = None
    kernels = {}
    extra = []
    if rank == 0:
        _C.PROFILE = {}
        for _ in range(min(3, args.steps)):
            step(collective=False)  # rank-0-only pass: must not enter a collective
        torch.cuda.synchronize()
        prof, _C.PROFILE = _C.PROFILE, None
        for name, recs in prof.items():
            groups, variants = {}, {}
            for a, b, cargs in recs:
                shown = name
                if name in ("zk_ar_forward", "zk_ar_forward_static", "zk_coupling_forward"):  # argument blocks (include/zuko_amd.h)
                    blk = cargs[0]
                    sizes = (blk.N, blk.D, blk.DIN) if name != "zk_coupling_forward" else (blk.N, blk.D, blk.C)
                    shown = "zk_ar_forward" if name != "zk_coupling_forward" else name  # (generic and static-shape instantiation of the same layer kernel)
                elif name in ("zk_linear_bf16_rqs", "zk_linear_bf16_rqs_lanes"):  # N, in, panels, K, features
                    sizes = (cargs[0], cargs[1], cargs[2], cargs[8], cargs[9])
                else:
                    sizes = cargs[0:3] if name == "zk_linear_bf16" else cargs[1:4]  # (dtype-less signature)
                key = (shown,) + tuple(int(v) for v in sizes if isinstance(v, int))
                groups.setdefault(key, []).append(a.elapsed_time(b))
                if shown == "zk_ar_forward":
                    variants.setdefault(key, set()).add("static-shape" if name == "zk_ar_forward_static" else "generic")
            for key, ts in groups.items():
                kernels[" ".join(map(str, key))] = {"calls": len(ts), "avg_ms": sum(ts) / len(ts), **({"bf16": True} if bf16 else {}),
                                                    **({"instantiation": "+".join(sorted(variants[key]))} if key in variants else {})}
        # the standalone (phi-in-HBM) spline kernel is not on the fused path: time it on its own so its
        # HBM fraction (the bandwidth-bound roofline of north_star) is measured in the same run
        if args.config == "cfg2":
            try:
                gen = torch.Generator(device=dev).manual_seed(3)
                phi = torch.randn(B, features, 3 * BINS - 1, generator=gen, device=dev)
                w, h, d = phi[..., :BINS], phi[..., BINS : 2 * BINS], phi[..., 2 * BINS :]
                # phi through non-temporal loads (library default) and through plain loads, interleaved in one process: which one
                # is faster differs between boxes by a few percent (BENCH_r01 0.75 / BENCH_r02 0.65 of 8 TB/s on the same kernel)
                variants = {"nt": [], "plain": []}
                prev = os.environ.get("ZUKO_AMD_K1_NT")
                with torch.no_grad():
                    for name in ("nt", "plain"):
                        os.environ["ZUKO_AMD_K1_NT"] = "1" if name == "nt" else "0"
                        ops.rqs_forward(x, w, h, d, reduce=True)
                    for _ in range(5):
                        for name in ("nt", "plain"):
                            os.environ["ZUKO_AMD_K1_NT"] = "1" if name == "nt" else "0"
                            _C.PROFILE = {}
                            ops.rqs_forward(x, w, h, d, reduce=True)
                            torch.cuda.synchronize()
                            variants[name] += [a.elapsed_time(b) for a, b, _ in _C.PROFILE.get("zk_rqs_forward", [])]
                            _C.PROFILE = None
                if prev is None:
                    os.environ.pop("ZUKO_AMD_K1_NT", None)
                else:
                    os.environ["ZUKO_AMD_K1_NT"] = prev
                med = {k: sorted(v)[len(v) // 2] for k, v in variants.items()}
                pick = min(med, key=med.get)
                kernels[f"zk_rqs_forward {B} {features} {BINS}"] = {"calls": len(variants[pick]), "avg_ms": sum(variants[pick]) / len(variants[pick]), "standalone": True, "phi_loads": pick,
                                                                    "median_ms_by_phi_load_policy": {k: round(v, 4) for k, v in med.items()},
                                                                    "library_default": "nt (ZUKO_AMD_K1_NT=0 selects plain loads)"}
                del phi, w, h, d
            except Exception as exc:  # never let the side measurement break the headline line
                _C.PROFILE = None
                kernels["zk_rqs_forward (standalone)"] = {"calls": 0, "avg_ms": float("nan"), "error": repr(exc)}
        per_transform = {k: v / transforms for k, v in flops.items()}
        st = None
        try:
            st = flow.transform.transforms[0].fused_state(dev)
        except Exception:
            st = None
        # 16 x 16 x 2 FLOP per streamed tile per sample (the static-shape kernel streams fewer tiles than the generic one)
        executed = None if st is None or not hasattr(st.plan, "kept_tiles") else (st.plan.fine_kept_tiles if getattr(st, "static_variant", 0) else st.plan.kept_tiles) * 512.0
        split = bool(st is not None and getattr(st, "static", None) is not None and st.static[0].meta.get("split"))
        per_transform["split_coupling"] = bool(st is not None and getattr(st, "split", False) is True)
        if split:  # operand-split kernels: 16 x 32 x 2 f32-equivalent FLOP per streamed block per sample
            from zuko_amd import static_ar

            tsp = static_ar.split_tables(st.plan, st.plan.layout.kind, st.act)[0]
            executed = (sum(tsp["NB"]) + tsp["GOFF"][-1] * st.plan.layout.nt) * 1024.0
        per_transform["split"] = split
        # the launch the product makes for this flow: the two-part (f16 x 2, three partial products) kernel when the weights are eligible (zuko_amd/fused.py)
        per_transform["half"] = bool(st is not None and hasattr(st, "_half_serves") and st._half_serves(torch.empty(4, features, device=dev)))
        per_transform["half_coupling"] = bool(st is not None and getattr(st, "half_able", False) and getattr(st, "half_ok", False) and zuko_amd.matmul_precision() == "f16x2")
        last = [m for m in flow.transform.transforms[0].hyper.modules() if getattr(m, "weight", None) is not None][-1]
        lmask = getattr(last, "mask", None)
        per_transform["last_layer_nnz_frac"] = 1.0 if lmask is None else float(lmask.float().mean())
        roof, extra = zuko_amd_roofline(kernels, B, per_transform, executed, args.config)

    if rank == 0:
        out = {
            "metric": "log_prob samples/sec, NSF d=64 K=8 bins=8 batch=2^20" if args.config == "cfg2" else f"log_prob samples/sec, {args.config}",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "rccl_world_size": dist.get_world_size() if dist is not None else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "per_rank_ms_per_step": per_rank_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if bf16 else "f32",
            **({"dtype_note": ("f32 in, f32 out, f32 accumulation; the conditioner's products run on the f16 matrix instruction with every f32 operand as TWO f16 numbers scaled into range by "
                               "exact powers of two (three partial products: the error of an f32 dot product, see `parity`); zuko_amd.set_matmul_precision('bf16x3') selects three bf16 parts / six "
                               "products, ZUKO_AMD_EXACT_F32=1 the f32 matrix instruction") if "two-part" in (roof or {}).get("instantiation", "") else
                              ("f32 in, f32 out, f32 accumulation; the conditioner's products run on the bf16 matrix instruction with every f32 operand split into three bf16 numbers "
                               "(six partial products: the rounding of an f32 dot product, see `parity`); ZUKO_AMD_EXACT_F32=1 selects the f32 matrix instruction")}
               if (roof or {}).get("f32_instruction_peak") else {}),
            "matmul_precision": zuko_amd.matmul_precision(),
            "data": "synthetic",
            "config": {
                "workload": f"{workload}, batch=2^{args.batch_log2} per GPU, x~N(0,1), seed-0 init",
                "batch_per_gpu": B,
                "global_batch": B * world,
                "parallelism": f"batch-sharded x{world}, one {'RCCL' if os.environ.get('ZUKO_BENCH_BACKEND', 'nccl') == 'nccl' else os.environ.get('ZUKO_BENCH_BACKEND')} all-reduce of the scalar NLL",
            },
            "roofline": roof,
            "end_to_end": {
                "flop_per_sample_dense": flops["dense"],
                "flop_per_sample_nonzero_weights": flops["nnz"],
                "tflops_on_nonzero_weights": value / world * flops["nnz"] / 1e12,
                "frac_of_mfma_peak_on_nonzero_weights": value / world * flops["nnz"] / 1e12 / (roof["peak"] if roof and roof.get("bound") == "mfma" else (PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS)),
                "tflops_dense_equiv": value / world * flops["dense"] / 1e12,
                "note": "dense_equiv counts masked-out (structurally zero) weights the kernels never multiply; it is NOT a hardware fraction",
            },
            "kernels": extra,
            "nll": nll_value,
            "cpu_affinity": affinity,
        }
        if solo is not None:  # (the driver computes scaling efficiency from its own per-N runs; this is the same ratio from ONE invocation)
            out["rank0_alone_before_group"] = solo
            out["weak_scaling_efficiency"] = value / world / solo["value"]
        wall = {"until_headline_done": round(time.perf_counter() - T_START, 1)}  # seconds each leg of this invocation took (the timed region is `ms_per_step`)
        tw = time.perf_counter()

        def lap(name):
            nonlocal tw
            wall[name] = round(time.perf_counter() - tw, 1)
            tw = time.perf_counter()

        if world == 1 and args.config == "cfg2" and not args.no_bin_report:
            try:
                out["bin_index"] = bin_report(flow, flow_cpu, x, dev)
            except Exception as exc:
                out["bin_index"] = {"error": repr(exc)}
            lap("bin_index")
        if world == 1 and not args.no_cpu_baseline and args.config == "cfg2":
            out["cpu_baseline"], sample = cpu_baseline(flow_cpu, args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_baseline_median"] = value / out["cpu_baseline"]["median_samples_per_s"]
            lap("cpu_baseline")
            try:
                out["parity"] = parity_report(flow, x, sample, dev)
            except Exception as exc:
                out["parity"] = {"error": repr(exc), "ok": False}
            lap("parity")
            try:
                out["gpu_aten_baseline"] = gpu_aten_baseline(flow_cpu, dev, value)
                if "speedup_vs_gpu_aten" in out["gpu_aten_baseline"]:
                    out["speedup_vs_gpu_aten"] = out["gpu_aten_baseline"]["speedup_vs_gpu_aten"]
            except Exception as exc:
                out["gpu_aten_baseline"] = {"error": repr(exc)}
            lap("gpu_aten_baseline")
        elif world == 1 and not args.no_cpu_baseline:
            try:
                del x
                torch.cuda.empty_cache()
                out["parity"] = side_parity(args.config, flow, flow_cpu, dev)
            except Exception as exc:
                out["parity"] = {"error": repr(exc), "ok": False}
            lap("parity")
        if world == 1 and args.config == "cfg2" and args.batch_log2 == 20 and not args.no_side_configs and os.environ.get("ZUKO_BENCH_SINGLE_DEVICE") != "1":
            # BASELINE.json configs[2..4] at their per-GPU share, a few steps each, after the headline's timed region
            del flow, x
            torch.cuda.empty_cache()
            out["side_configs"] = run_side_configs(args)
            lap("side_configs")
            out["side_paths"] = run_side_paths()  # training step and sampling of the cfg2 / cfg3 flows, each with its own check
            lap("side_paths")
        wall["total"] = round(time.perf_counter() - T_START, 1)
        out["wall_s"] = wall
        if args.full_line:
            print(json.dumps(out))
        else:
            out["detail"] = write_detail(out)
            sys.stdout.flush()
            print(json.dumps(headline_line(out)), flush=True)  # the LAST stdout line, <= LINE_BUDGET bytes
    if dist is not None:
        dist.destroy_process_group()


def _pmc_traffic(B: int, config: str):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/rNN/traffic.json;
    FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 read-side correction as MI355X_MICROARCH.md prescribes) — valid
    only for the default 2^20 headline workload they were taken on; counters cannot be read inside this process."""
    import glob

    if B != 1 << 20 or config != "cfg2":
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f).get("hbm_bytes_per_launch"), os.path.relpath(files[-1], ROOT)


def zuko_amd_roofline(kernels: dict, B: int, flop_per_transform: dict, executed_per_sample, config: str):
    """Roofline object for the dominant kernel + a per-kernel table (all measured in this run).

    MFMA-bound kernels: `achieved` = FLOPs on the non-zero (unmasked) weights the launch has to multiply / launch time
    (SURVEY 8d's mask-aware figure; 531 568 per sample per transform at cfg2), so `frac` is a hardware fraction <= 1;
    `frac_executed` additionally counts the zeros inside the 16x16 tiles the kernel keeps (what the matrix pipe issues).
    HBM-bound kernels: algorithmic bytes / time."""
    table = []
    for name, rec in kernels.items():
        row = {"kernel": name, **rec}
        parts = name.split()
        t = rec["avg_ms"] * 1e-3
        if parts[0] in ("zk_linear_bf16_rqs", "zk_linear_bf16_rqs_lanes"):  # last layer + spline in one kernel (first / second generation)
            n, fin, k, feats = int(parts[1]), int(parts[2]), int(parts[4]), int(parts[5])
            dense = 2.0 * n * fin * feats * (3 * k - 1)
            row.update(bound="mfma", achieved=dense * flop_per_transform.get("last_layer_nnz_frac", 1.0) / t / 1e12, peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s",
                       note=f"achieved counts the unmasked weights only ({flop_per_transform.get('last_layer_nnz_frac', 1.0):.4f} of the dense last layer)")
            row["dense_equiv_tflops"] = dense / t / 1e12
        elif parts[0] == "zk_linear_bf16":
            n, fin, fout = int(parts[1]), int(parts[2]), int(parts[3])
            row.update(bound="mfma", achieved=2.0 * n * fin * fout / t / 1e12, peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s", note="dense-equivalent (mask not accounted per layer)")
        elif parts[0] == "zk_linear":
            n, fin, fout = int(parts[1]), int(parts[2]), int(parts[3])
            row.update(bound="mfma", achieved=2.0 * n * fin * fout / t / 1e12, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s")
        elif parts[0] in ("zk_ar_forward", "zk_coupling_forward"):
            split = (bool(flop_per_transform.get("split")) and parts[0] == "zk_ar_forward" and "static" in rec.get("instantiation", "")) or \
                    (bool(flop_per_transform.get("split_coupling")) and parts[0] == "zk_coupling_forward")
            # operand-split kernels: every f32 product = 6 bf16 matrix products (csrc/fused_ar_split_impl.h), so the ceiling for f32-equivalent
            # FLOP is the dense bf16 peak / 6; the f32 matrix instruction's own peak stays on the line for comparison
            half = (bool(flop_per_transform.get("half")) and parts[0] == "zk_ar_forward" and "static" in rec.get("instantiation", "")) or \
                   (bool(flop_per_transform.get("half_coupling")) and parts[0] == "zk_coupling_forward")
            split = split and not half
            peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if half else (PEAK_BF16_MFMA_TFLOPS / 6.0 if split else PEAK_F32_MFMA_TFLOPS)
            row.update(bound="mfma", achieved=float(B) * flop_per_transform["nnz"] / t / 1e12, peak=peak, unit="TFLOP/s")
            if half:
                # two-part kernels: every f32 product = 3 f16 matrix products (csrc/fused_ar_half_impl.h): ceiling = dense 16-bit peak / 3, f32-equivalent FLOP
                row["instantiation"] = "static-shape, two-part operand split (2 x f16 per f32 operand scaled by exact powers of two, 3 partial products, f32 accumulate)"
                row["peak_basis"] = f"{PEAK_BF16_MFMA_TFLOPS:g} TFLOP/s dense f16 / 3 matrix products per f32 product; f32-equivalent FLOP"
                row["f32_instruction_peak"] = PEAK_F32_MFMA_TFLOPS
                row["frac_on_the_six_product_basis_of_rounds_3_to_5"] = row["achieved"] / (PEAK_BF16_MFMA_TFLOPS / 6.0)
            if split:
                row["instantiation"] = "static-shape, operand-split (3 x bf16 per f32 operand, 6 partial products, f32 accumulate)"
                row["peak_basis"] = f"{PEAK_BF16_MFMA_TFLOPS:g} TFLOP/s dense bf16 / 6 matrix products per f32 product; f32-equivalent FLOP"
                row["f32_instruction_peak"] = PEAK_F32_MFMA_TFLOPS
                row["instruction_form_ceiling"] = {"form": "v_mfma_f32_16x16x32_bf16, two wavefronts per SIMD", "frac_of_bf16_peak": MFMA_16x16x32_ISSUE_CEILING,
                                                   "real_cycles_per_instruction_and_simd": 16.3, "pipe_cycles_per_instruction": 16,
                                                   "source": "profiles/r05/mfma_clock_probe.txt (random operands, ~2.2 GHz under that load; a property of the chip, not re-measured in this run)",
                                                   "note": "the kernel's matrix pipe is busy 0.62 of the launch (profiles/r05/traffic.json: GRBM_GUI_ACTIVE / 8 = 5.97e6 real cycles per launch, 2.25 GHz under the profiler): it is NOT at this ceiling; "
                                                           "its conversion / spline / ring phases run beside idle matrix pipes (profiles/r05/headline.md)"}
            row["algorithmic_flop_per_launch"] = float(B) * flop_per_transform["nnz"]
            row["dense_equiv_tflops"] = float(B) * flop_per_transform["dense"] / t / 1e12
            if executed_per_sample and parts[0] == "zk_ar_forward" and (split or half or not flop_per_transform.get("split")):
                row["executed_flop_per_launch"] = float(B) * executed_per_sample
                row["achieved_executed"] = float(B) * executed_per_sample / t / 1e12
                row["frac_executed"] = row["achieved_executed"] / peak
        elif parts[0] == "zk_rqs_forward" and len(parts) == 4 and parts[1].isdigit():
            n, d, k = int(parts[1]), int(parts[2]), int(parts[3])
            esz = 2 if rec.get("bf16") else 4
            byts = float(n) * (d * (esz + esz * (3 * k - 1) + esz) + 4)  # x + phi + y per element, + ladj per row
            row.update(bound="hbm", achieved=byts / t / 1e9, peak=PEAK_HBM_GBPS, unit="GB/s")
        if "achieved" in row:
            row["frac"] = row["achieved"] / row["peak"]
            if "instruction_form_ceiling" in row and row.get("frac_executed"):
                row["frac_of_form_ceiling_executed"] = row["frac_executed"] / MFMA_16x16x32_ISSUE_CEILING  # issued matrix work vs what the form can be issued at
        table.append(row)
    dom = max((r for r in table if not r.get("standalone")), key=lambda r: r["avg_ms"] * r["calls"])
    roof = None
    if "achieved" in dom:
        traffic, src = _pmc_traffic(B, config)
        roof = {"bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"],
                "traffic": traffic, "traffic_source": (f"{src}: rocprofv3 --pmc passes of this same command, not re-measured in this run" if src else None),
                "kernel": dom["kernel"], "avg_launch_ms": dom["avg_ms"]}
        for k in ("instantiation", "peak_basis", "f32_instruction_peak", "frac_on_the_six_product_basis_of_rounds_3_to_5", "instruction_form_ceiling", "frac_of_form_ceiling_executed", "algorithmic_flop_per_launch", "executed_flop_per_launch", "achieved_executed", "frac_executed", "dense_equiv_tflops", "note"):
            if k in dom:
                roof[k] = dom[k]
    return roof, table


if __name__ == "__main__":
    main()
