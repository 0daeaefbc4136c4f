#!/bin/bash
# Everything the round's numbers come from, in one GPU-box call (outputs under gpurun_out/final_<tag>/)
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/final_$TAG; mkdir -p $OUT
cd $ROOT
timeout 600 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_2p20.json; cut -c1-300 $OUT/bench_2p20.json
ZUKO_BENCH_SINGLE_DEVICE=1 ZUKO_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --batch-log2 19 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_2ranks_1gpu_dryrun.json; cut -c1-200 $OUT/bench_2ranks_1gpu_dryrun.json
timeout 300 python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg3_2p20.json; cut -c1-200 $OUT/bench_cfg3_2p20.json
timeout 300 python bench.py --config cfg4 --batch-log2 19 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg4_2p19.json; cut -c1-200 $OUT/bench_cfg4_2p19.json
timeout 600 python bench.py --config cfg5 --batch-log2 19 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg5_2p19.json; cut -c1-200 $OUT/bench_cfg5_2p19.json
bash scripts/k1.sh 2>&1 | grep rqs_ | tee $OUT/k1.txt
timeout 600 python scripts/inc_check.py 2>&1 | grep -v amdgpu | tee $OUT/inverse.txt
timeout 600 python scripts/train_bench.py 2>&1 | grep "^NSF\|^MAF" | tee $OUT/train.txt
