#!/bin/bash
# End-of-round evidence (round 5): full GPU suite, smoke, the bench line.  Outputs under gpurun_out/final5/ -> copied to profiles/r05/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/final5; mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json; tail -3 $OUT/bench.err
