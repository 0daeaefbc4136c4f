#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/final5c; mkdir -p $OUT
cd $ROOT
timeout 480 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
