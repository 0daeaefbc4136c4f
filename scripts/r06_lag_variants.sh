#!/bin/bash
# Round 6: probe builds of the headline kernel with the late half of the workgroup LAG chunk barriers behind (csrc/fused_ar_static_impl.h: ArRingS LAG).
# usage: scripts/r06_lag_variants.sh build   (here)  |  run <outdir>  (GPU box)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
V=(
 "base||"
 "lag1||-DARX_LAG=1"
 "lag1p1||-DARX_LAG=1 -DARX_PRIO=1"
 "lag1p2||-DARX_LAG=1 -DARX_PRIO=2"
 "x0|XLDS=0|"
 "lag2x0|XLDS=0|-DARX_LAG=2"
 "nr2|NR=2|"
 "nr2lag1|NR=2|-DARX_LAG=1"
 "nr2lag2|NR=2|-DARX_LAG=2"
)
if [ "$1" = build ]; then
  n=0
  for v in "${V[@]}"; do
    IFS='|' read -r tag shape flags <<< "$v"
    ABL_TAG=_$tag ABL_ONLY0=1 ABL_SHAPE="$shape" python scripts/split_ablate.py build $flags &
    n=$((n+1)); if [ $((n % 4)) = 0 ]; then wait; fi
  done
  wait
else
  OUT=gpurun_out/${2:-lagvar}; mkdir -p $OUT
  for rep in 1 2; do
    for v in "${V[@]}"; do
      IFS='|' read -r tag shape flags <<< "$v"
      echo -n "$tag: " | tee -a $OUT/variants.txt
      ABL_TAG=_$tag timeout 300 python scripts/split_ablate.py run 20 0 2>&1 | tail -1 | tee -a $OUT/variants.txt
    done
  done
fi
