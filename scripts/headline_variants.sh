#!/bin/bash
# headline kernel (NSF cfg2 operand-split arx_kernel): probe builds of scripts/split_ablate.py (ABL_TAG=_<name>), one transform per launch at 2^20 rows
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-headvar}; mkdir -p $OUT; cd $ROOT; shift
for rep in 1 2; do
for name in "$@"; do
  echo -n "$name: " | tee -a $OUT/variants.txt
  ABL_TAG=_$name timeout 300 python scripts/split_ablate.py run 20 0 2>&1 | tail -1 | tee -a $OUT/variants.txt
done
done
