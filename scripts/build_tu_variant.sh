#!/bin/bash
# builds scripts/probes/ab/lib_<name>.so = the current library with ONE translation unit recompiled under extra flags
# usage: build_tu_variant.sh <tu (e.g. linear_bf16)> <name> <extra hipcc flags...>
TU=$1; NAME=$2; shift 2
ROOT=$(cd $(dirname $0)/.. && pwd); OUT=$ROOT/scripts/probes/ab; mkdir -p $OUT
EXTRA=""; [ "$TU" = fused_coupling ] && EXTRA="-mllvm -pragma-unroll-threshold=1000000"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=off $EXTRA "$@" -c $ROOT/zuko_amd/csrc/$TU.hip -o $OUT/${TU}_$NAME.o || exit 1
OBJS=$(ls $ROOT/zuko_amd/lib/*.o | grep -v "/$TU.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt $OUT/${TU}_$NAME.o $OBJS -L/usr/local/lib/python3.10/dist-packages/torch/lib -l:libamdhip64.so -o $OUT/lib_$NAME.so && rm -f $OUT/${TU}_$NAME.o && echo built $OUT/lib_$NAME.so
