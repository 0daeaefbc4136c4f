#!/bin/bash
# One GPU-box call: GPU test-suite (all failures, not -x), smoke, default bench line.  Outputs under gpurun_out/<tag>/
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -150 > $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
timeout 600 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json; cut -c1-600 $OUT/bench_default.json
