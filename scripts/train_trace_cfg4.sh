#!/bin/bash
# rocprofv3 kernel-trace stats of the RealNVP cfg4 training step at 2^14 rows (run on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r06t}; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/scripts/train_profile_cfg4.py > $OUT/trace_cfg4.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_train_cfg4.csv \;
rm -rf $OUT/trace
grep "whole step" $OUT/trace_cfg4.log
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats_train_cfg4.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("sum of kernel time over the 14 steps of the script: %.1f ms -> %.2f ms per step" % (tot/1e6, tot/1e6/14))
for r in rows[:25]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:9.1f} {r['Percentage']}%")
PY
