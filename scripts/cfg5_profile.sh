#!/bin/bash
# cfg5 (bf16 NSF-1024) bench line + rocprofv3 kernel-trace stats (run on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/prof_cfg5; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
python $ROOT/bench.py --config cfg5 --batch-log2 19 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_cfg5_2p19.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $ROOT/bench.py --config cfg5 --batch-log2 19 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_cfg5.csv \;
rm -rf $OUT/trace
cut -c1-600 $OUT/bench_cfg5_2p19.json; head -6 $OUT/kernel_stats_cfg5.csv | cut -c1-160
