#!/bin/bash
# cfg5 GEMM A/B on one box: tile-walk variants of the bf16 layer kernels (run on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-cfg5ab}; mkdir -p $OUT; cd $ROOT
for m in 0 default 8,4,2,4 2,16,4,2 4,8,2,4; do
  if [ $m = default ]; then unset ZUKO_AMD_BF16_MAP; else export ZUKO_AMD_BF16_MAP=$m; fi
  echo "== MAP=$m" | tee -a $OUT/gemm.txt
  LOG2N=${LOG2N:-19} REPS=3 timeout 300 python scripts/bf16_gemm_probe.py 2>&1 | grep "N=2" | tee -a $OUT/gemm.txt
done
unset ZUKO_AMD_BF16_MAP
for m in 0 default; do
  if [ $m = default ]; then unset ZUKO_AMD_BF16_MAP; else export ZUKO_AMD_BF16_MAP=$m; fi
  timeout 600 python bench.py --config cfg5 --batch-log2 19 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg5_map_$m.json
  python - <<PY
import json
d=json.load(open("$OUT/bench_cfg5_map_$m.json"))
print("MAP=$m", d["value"], d["ms_per_step"], [(k["kernel"], round(k["avg_ms"],3)) for k in d["kernels"]])
PY
done | tee $OUT/bench.txt
