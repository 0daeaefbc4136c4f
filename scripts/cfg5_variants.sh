#!/bin/bash
# cfg5 last-layer kernel: time library variants built by scripts/build_tu_variant.sh (scripts/probes/ab/lib_<name>.so) on ONE box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-cfg5var}; mkdir -p $OUT; cd $ROOT
shift
for rep in 1 2; do
for name in base "$@"; do
  if [ $name = base ]; then unset ZUKO_AMD_LIB; else export ZUKO_AMD_LIB=$ROOT/scripts/probes/ab/lib_$name.so; fi
  echo -n "$name: " | tee -a $OUT/layer.txt
  timeout 300 python scripts/cfg5_layer_time.py 2>/dev/null | tail -1 | tee -a $OUT/layer.txt
done
done
unset ZUKO_AMD_LIB
