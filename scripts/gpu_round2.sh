#!/bin/bash
# GPU call 2 of round 3: static-kernel tests first (fast fail), the shape sweep, then the whole GPU suite and the default bench line
TAG=${1:-r03b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_flows.py -m gpu -q -rf -k "static_shape or wide_cond or ring_kernels" 2>&1 | tail -40 > $OUT/pytest_static.txt; tail -5 $OUT/pytest_static.txt
timeout 900 python scripts/static_shapes_bench.py 20 > $OUT/static_shapes.jsonl 2>$OUT/static_shapes.err; cat $OUT/static_shapes.jsonl; tail -3 $OUT/static_shapes.err
timeout 1500 python -m pytest tests -m gpu -q -rf 2>&1 | tail -60 > $OUT/pytest_gpu.txt; tail -5 $OUT/pytest_gpu.txt
cp gpurun_out/parity_report.json $OUT/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
timeout 600 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_default.json; cut -c1-300 $OUT/bench_default.json
ls zuko_amd/lib/ars/*.so | wc -l
