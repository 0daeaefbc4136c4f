#!/bin/bash
# cfg5 last layer: second-generation kernel (zk_linear_bf16_rqs_lanes) against the first (ZUKO_AMD_BF16_PANELS256=1) and its probe builds, on ONE box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-cfg5lanes}; mkdir -p $OUT; cd $ROOT
shift
for rep in 1 2; do
  echo -n "lanes: " | tee -a $OUT/layer.txt; timeout 300 python scripts/cfg5_layer_time.py 2>&1 | tail -1 | tee -a $OUT/layer.txt
  echo -n "panels256: " | tee -a $OUT/layer.txt; ZUKO_AMD_BF16_PANELS256=1 timeout 300 python scripts/cfg5_layer_time.py 2>&1 | tail -1 | tee -a $OUT/layer.txt
  for name in "$@"; do
    echo -n "$name: " | tee -a $OUT/layer.txt
    ZUKO_AMD_LIB=$ROOT/scripts/probes/ab/lib_$name.so timeout 300 python scripts/cfg5_layer_time.py 2>&1 | tail -1 | tee -a $OUT/layer.txt
  done
done
