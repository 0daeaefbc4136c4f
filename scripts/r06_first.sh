#!/bin/bash
# Round 6, first GPU call: full GPU suite, smoke, the default bench line (compact) + the detail file.  Outputs under gpurun_out/r06a/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06a; mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench_stdout.txt 2> $OUT/bench.err; tail -n 1 $OUT/bench_stdout.txt | wc -c; tail -n 1 $OUT/bench_stdout.txt | head -c 1500; tail -3 $OUT/bench.err
cp gpurun_out/bench_detail.json $OUT/bench_detail.json
