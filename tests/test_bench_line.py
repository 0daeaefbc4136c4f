"""bench.py prints ONE compact JSON line (<= 8 KB) as its last stdout line; the full object goes to gpurun_out/bench_detail.json.
Round 5's 20 KB line was not picked up by the driver (BENCH_r05.json "parsed": null): this holds the budget on a representative
full object — round 5's own (profiles/r05/bench.json) — and on an adversarially bloated one."""

import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _full():
    with open(os.path.join(ROOT, "profiles", "r05", "bench.json")) as f:
        out = json.load(f)
    out["detail"] = bench.DETAIL_PATH
    return out


def test_line_of_a_real_run_fits_and_keeps_the_contract():
    out = _full()
    assert len(json.dumps(out)) > 16000  # (the object that did not parse)
    line = bench.headline_line(out)
    text = json.dumps(line)
    assert len(text) < bench.LINE_BUDGET == 8192, len(text)
    assert "\n" not in text
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == float(f"{out['value']:.6g}") and line["config"]["workload"] == out["config"]["workload"]
    r = line["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms")) <= set(r)
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    c = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and c["kind"] in ("reference", "port")
    assert line["parity"]["ok"] is True and line["parity"]["rows"] == out["parity"]["rows"]
    assert line["bin_index"]["mismatch_vs_own_knots"] == 0
    # one compact entry per side configuration / side path
    assert set(line["side_configs"]) == set(out["side_configs"]) and set(line["side_paths"]) == set(out["side_paths"])
    for e in line["side_configs"].values():
        assert {"value", "frac", "parity_ok"} <= set(e)
    assert all(len(json.dumps(e)) < 400 for e in list(line["side_configs"].values()) + list(line["side_paths"].values()))


def test_line_budget_holds_whatever_the_run_adds():
    out = _full()
    for i in range(40):  # forty more side paths and configurations than any run has
        out["side_paths"][f"extra_{i}"] = copy.deepcopy(out["side_paths"]["nsf_cfg2"])
        out["side_configs"][f"cfgx{i}"] = copy.deepcopy(out["side_configs"]["cfg5"])
    out["per_rank_ms_per_step"] = [20.1234567] * 8
    line = bench.headline_line(out)
    assert len(json.dumps(line)) < bench.LINE_BUDGET
    for k in CONTRACT:
        assert k in line, k
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0


def test_errors_in_side_entries_stay_short():
    out = _full()
    out["side_configs"]["cfg4"] = {"error": "x" * 5000, "stderr_tail": "y" * 600}
    out["side_paths"]["maf_cfg3"] = {"error": "z" * 5000}
    line = bench.headline_line(out)
    assert len(json.dumps(line)) < bench.LINE_BUDGET
    assert len(line["side_configs"]["cfg4"]["error"]) <= 120


def test_multi_rank_line_carries_the_scaling_fields():
    out = _full()
    out.update(n_gpus=8, rccl_world_size=8, per_rank_ms_per_step=[20.5] * 8, rank0_alone_before_group={"steps": 100, "ms_per_step": 20.4, "value": 5.1e7}, weak_scaling_efficiency=0.99)
    for k in ("cpu_baseline", "parity", "bin_index", "side_configs", "side_paths"):
        out.pop(k)
    line = bench.headline_line(out)
    assert line["weak_scaling_efficiency"] == 0.99 and len(line["per_rank_ms_per_step"]) == 8 and line["rccl_world_size"] == 8
    assert len(json.dumps(line)) < 4096


def test_non_finite_numbers_do_not_break_json():
    out = _full()
    out["roofline"]["traffic"] = None
    out["nll"] = float("nan")
    text = json.dumps(bench.headline_line(out))
    assert "NaN" not in text and json.loads(text)["roofline"]["traffic"] is None


def test_line_of_the_round_6_run_fits():
    """The full object of a round-6 run (profiles/r06/bench_detail.json: two-part kernels, sampling at 2^20, cfg5 parity at 1 024 rows)."""
    path = os.path.join(ROOT, "profiles", "r06", "bench_detail.json")
    with open(path) as f:
        out = json.load(f)
    line = bench.headline_line(out)
    text = json.dumps(line)
    assert len(text) < bench.LINE_BUDGET
    for k in CONTRACT:
        assert k in line, k
    assert line["matmul_precision"] == "f16x2" and "3 matrix products" in line["roofline"]["peak_basis"]
    assert line["side_paths"]["nsf_cfg2"]["sample"]["batch_log2"] == 20 and line["side_configs"]["cfg5"]["parity_rows"] == 1024
