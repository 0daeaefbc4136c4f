#!/bin/bash
# builds scripts/probes/ab/lib_<name>.so = the current library with fused_ar.hip recompiled under extra flags
# usage: build_variant.sh <name> <extra hipcc flags...>
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd); OUT=$ROOT/scripts/probes/ab; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=off "$@" -c $ROOT/zuko_amd/csrc/fused_ar.hip -o $OUT/fused_ar_$NAME.o || exit 1
OBJS=$(ls $ROOT/zuko_amd/lib/*.o | grep -v fused_ar.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt $OUT/fused_ar_$NAME.o $OBJS -L/usr/local/lib/python3.10/dist-packages/torch/lib -l:libamdhip64.so -o $OUT/lib_$NAME.so && echo built $OUT/lib_$NAME.so
