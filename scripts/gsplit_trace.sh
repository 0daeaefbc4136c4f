#!/bin/bash
# rocprofv3 kernel-trace stats of the headline flow on the generic operand-split kernel (ZUKO_AMD_NO_STATIC_AR=1) and on the generated kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for mode in generic generated; do
  rm -rf /tmp/gst
  if [ $mode = generic ]; then export ZUKO_AMD_NO_STATIC_AR=1; else unset ZUKO_AMD_NO_STATIC_AR; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gst -o t -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bin-report --no-side-configs > $OUT/gsplit_trace_$mode.log 2>&1
  f=$(find /tmp/gst -name "*kernel_stats.csv" | head -1)
  head -4 "$f" | cut -c1-260 | tee $OUT/kernel_trace_stats_headline_$mode.csv
  cp "$f" $OUT/kernel_trace_stats_headline_${mode}_full.csv
  grep -o '"value": [0-9.]*' $OUT/gsplit_trace_$mode.log | head -1
done
