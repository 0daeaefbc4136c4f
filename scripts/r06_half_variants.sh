#!/bin/bash
# Round 6: the two-part (f16 x 2) headline kernel — timing ablations (ARX_ABL) and probe builds.   build (here) | run <outdir> (GPU box)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
export ABL_HALF=1
V=(
 "base|"
 "look2|-DARH_LOOK=2"
 "look3|-DARH_LOOK=3"
 "reluint|-DARH_RELU_INT=1"
 "nofence|-DARX_FENCE=0"
 "look2nofence|-DARH_LOOK=2 -DARX_FENCE=0"
)
if [ "$1" = build ]; then
  ABL_TAG=_abl python scripts/split_ablate.py build &
  n=1
  for v in "${V[@]}"; do
    IFS='|' read -r tag flags <<< "$v"
    ABL_TAG=_$tag ABL_ONLY0=1 python scripts/split_ablate.py build $flags &
    n=$((n+1)); if [ $((n % 3)) = 0 ]; then wait; fi
  done
  wait
else
  OUT=gpurun_out/${2:-halfvar}; mkdir -p $OUT
  ABL_TAG=_abl timeout 600 python scripts/split_ablate.py run 20 2>&1 | grep ARX_ABL | tee -a $OUT/ablations.txt
  for rep in 1 2 3; do
    for v in "${V[@]}"; do
      IFS='|' read -r tag flags <<< "$v"
      echo -n "$tag: " | tee -a $OUT/variants.txt
      ABL_TAG=_$tag timeout 300 python scripts/split_ablate.py run 20 0 2>&1 | tail -1 | tee -a $OUT/variants.txt
    done
  done
fi
