#!/bin/bash
# GPU call 3 of round 3: whole GPU suite on the argument-block ABI + generated static kernels, then every number of the round
TAG=${1:-r03c}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -60 > $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
timeout 600 python scripts/static_shapes_bench.py 20 > $OUT/static_shapes.jsonl 2>$OUT/static_shapes.err; cat $OUT/static_shapes.jsonl
bash scripts/round_final.sh $TAG
