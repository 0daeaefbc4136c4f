#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/final5d; mkdir -p $OUT
cd $ROOT
timeout 420 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
timeout 120 python bench.py --no-cpu-baseline --no-side-configs --no-bin-report > $OUT/bench_quick.json 2> $OUT/bench_quick.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final5d/bench_quick.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d.get("nll"))
PY
