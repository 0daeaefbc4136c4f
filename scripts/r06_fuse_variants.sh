#!/bin/bash
# Round 6: probe builds of the headline kernel — conversions fused into the producing layer (arx_hidden_fused) against the round-5 form.
# usage: scripts/r06_fuse_variants.sh build   (here)  |  run <outdir>  (GPU box)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
V=(
 "r5form|||ZUKO_AMD_SPLIT_FUSE=0"
 "fuse|||ZUKO_AMD_SPLIT_FUSE=1"
 "fuseq1||-DARX_QMIX=1|ZUKO_AMD_SPLIT_FUSE=1"
 "fuseq2||-DARX_QMIX=2|ZUKO_AMD_SPLIT_FUSE=1"
 "fusenf||-DARX_FENCE=0|ZUKO_AMD_SPLIT_FUSE=1"
 "fuselag1||-DARX_LAG=1|ZUKO_AMD_SPLIT_FUSE=1"
)
if [ "$1" = build ]; then
  n=0
  for v in "${V[@]}"; do
    IFS='|' read -r tag shape flags envs <<< "$v"
    env $envs ABL_TAG=_$tag ABL_ONLY0=1 ABL_SHAPE="$shape" python scripts/split_ablate.py build $flags &
    n=$((n+1)); if [ $((n % 4)) = 0 ]; then wait; fi
  done
  wait
else
  OUT=gpurun_out/${2:-fusevar}; mkdir -p $OUT
  for rep in 1 2 3; do
    for v in "${V[@]}"; do
      IFS='|' read -r tag shape flags envs <<< "$v"
      echo -n "$tag: " | tee -a $OUT/variants.txt
      env $envs ABL_TAG=_$tag timeout 300 python scripts/split_ablate.py run 20 0 2>&1 | tail -1 | tee -a $OUT/variants.txt
    done
  done
fi
