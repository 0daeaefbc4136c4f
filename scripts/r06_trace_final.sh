#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench command's headline workload on the final tree (run on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r06_trace_final; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-bin-report --no-side-configs > $OUT/trace.log 2>&1
tail -1 $OUT/trace.log | cut -c1-300
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -6 $OUT/kernel_stats.csv
rm -rf $OUT/trace
