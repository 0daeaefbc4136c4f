#!/bin/bash
# Everything the round's numbers come from, in one GPU-box call (outputs under gpurun_out/final/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/final; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
python bench.py 2>/dev/null | tail -1 > $OUT/bench_default.json; cut -c1-300 $OUT/bench_default.json
bash scripts/gpu_profile.sh r01 > $OUT/gpu_profile.log 2>&1; tail -3 $OUT/gpu_profile.log
bash scripts/k1.sh 2>&1 | grep rqs_ | tee $OUT/k1.txt
python bench.py --config cfg3 --batch-log2 19 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg3_2p19.json; cut -c1-200 $OUT/bench_cfg3_2p19.json
python bench.py --config cfg4 --batch-log2 19 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg4_2p19.json; cut -c1-200 $OUT/bench_cfg4_2p19.json
python scripts/inverse_bench.py 2>&1 | grep "batch" | tee $OUT/inverse.txt
python scripts/train_bench.py 2>&1 | grep "batch" | tee $OUT/train.txt
bash scripts/cfg5_profile.sh > $OUT/cfg5.log 2>&1; tail -4 $OUT/cfg5.log | cut -c1-300
