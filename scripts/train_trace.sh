#!/bin/bash
# rocprofv3 kernel-trace stats of the NSF cfg2 training step (run on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-train_trace}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/scripts/train_bench.py > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_train.csv \;
rm -rf $OUT/trace
grep "^NSF\|^MAF" $OUT/trace.log
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats_train.csv")))
for r in rows[:22]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:9.1f} {r['Percentage']}%")
PY
