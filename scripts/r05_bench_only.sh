#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/final5b; mkdir -p $OUT
cd $ROOT
date +%s > $OUT/bench_t0
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
date +%s > $OUT/bench_t1
echo "bench seconds: $(( $(cat $OUT/bench_t1) - $(cat $OUT/bench_t0) ))"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final5b/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["wall_s"])
print({k: v.get("wall_s") for k, v in d["side_configs"].items()})
print({k: v.get("wall_s") for k, v in d["side_paths"].items() if isinstance(v, dict)})
PY
