#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/final5e; mkdir -p $OUT
cd $ROOT
timeout 200 python -m pytest tests/test_gpu_flows.py tests/test_gpu_reference_suite.py -q -m gpu -k "coupling or realnvp or nice or golden or split_coupling" 2>&1 | tail -4 | tee $OUT/pytest_coupling.txt
timeout 100 python bench.py --config cfg4 --batch-log2 19 --steps 6 --warmup 2 2> $OUT/cfg4.err | tail -1 > $OUT/cfg4.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final5e/cfg4.json").read())
print("cfg4", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("parity", {}).get("ok"), d.get("parity", {}).get("log_prob_max_rel"), d.get("nll"))
PY
