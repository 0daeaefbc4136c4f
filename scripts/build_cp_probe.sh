#!/bin/bash
# builds scripts/probes/ab/lib_cp_${NAME}.so = the current library with fused_coupling.hip recompiled with -DZK_CP_TIMING=1 (+ extra flags)
NAME=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd); OUT=$ROOT/scripts/probes/ab; mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -ffp-contract=off -mllvm -pragma-unroll-threshold=1000000 "$@" -c $ROOT/zuko_amd/csrc/fused_coupling.hip -o $OUT/fused_coupling_${NAME}.o || exit 1
OBJS=$(ls $ROOT/zuko_amd/lib/*.o | grep -v fused_coupling.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -no-hip-rt $OUT/fused_coupling_${NAME}.o $OBJS -L/usr/local/lib/python3.10/dist-packages/torch/lib -l:libamdhip64.so -o $OUT/lib_cp_${NAME}.so && echo built $OUT/lib_cp_${NAME}.so
