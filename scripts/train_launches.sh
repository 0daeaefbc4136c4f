#!/bin/bash
# launches per training step, by kernel (rocprofv3 --kernel-trace of STEPS identical NSF cfg2 Adam steps); run on the GPU box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r05; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
export STEPS=${STEPS:-12}
for b in ${BATCHES:-16 8}; do
  rm -rf /tmp/tl
  LOG2B=$b rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tl -o t -- python $ROOT/scripts/train_launches.py > $OUT/train_launches_$b.log 2>&1
  f=$(find /tmp/tl -name "*kernel_stats.csv" | head -1)
  python - <<PY | tee $OUT/train_launches_$b.txt
import csv
rows = list(csv.DictReader(open("$f")))
steps = $STEPS
print(f"NSF cfg2 Adam step, batch 2^$b: launches and GPU time per step over {steps} steps (incl. one-time setup in the first)")
tc = tt = 0
for r in rows:
    c, t = int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps
    tc += c; tt += t
    if c >= 0.5:
        print(f"{r['Name'][:110]:110s} {c:7.1f} launches {t:8.3f} ms")
print(f"total {tc:.0f} launches, {tt:.3f} ms of kernel time per step")
PY
done
