#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/final5; mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_bins.py tests/test_gpu_flows.py -q -m gpu -k "bin_index or generic_split or without_a_generated or cfg5_full" 2>&1 | tail -4 | tee $OUT/pytest_fixups.txt
date +%s > $OUT/bench_t0
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
date +%s > $OUT/bench_t1
echo "bench seconds: $(( $(cat $OUT/bench_t1) - $(cat $OUT/bench_t0) ))"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final5/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["wall_s"])
print({k: v.get("wall_s") for k, v in d["side_configs"].items()})
print({k: v.get("wall_s") for k, v in d["side_paths"].items() if isinstance(v, dict)})
PY
