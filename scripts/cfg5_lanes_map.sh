#!/bin/bash
# cfg5 last layer (second-generation kernel): XCD patch shapes of the tile walk, on ONE box.  usage: cfg5_lanes_map.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-cfg5map}; mkdir -p $OUT; cd $ROOT
for m in default 0 "8,4,4,2" "2,16,4,2" "4,8,8,1" "8,4,8,1" "4,8,2,4" "16,2,2,4"; do
  if [ "$m" = default ]; then unset ZUKO_AMD_BF16_MAP; else export ZUKO_AMD_BF16_MAP=$m; fi
  echo -n "map $m: " | tee -a $OUT/map.txt; timeout 300 python scripts/cfg5_layer_time.py 2>&1 | tail -1 | cut -c1-200 | tee -a $OUT/map.txt
done
