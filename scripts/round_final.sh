#!/bin/bash
# Everything the round's numbers come from, in one GPU-box call (outputs under gpurun_out/final_<tag>/)
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/final_$TAG; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
timeout 600 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_2p20.json; cut -c1-300 $OUT/bench_2p20.json
ZUKO_AMD_EXACT_F32=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_2p20_exact_f32.json; cut -c1-200 $OUT/bench_2p20_exact_f32.json
ZUKO_BENCH_SINGLE_DEVICE=1 ZUKO_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --batch-log2 19 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_2ranks_1gpu_dryrun.json; cut -c1-200 $OUT/bench_2ranks_1gpu_dryrun.json
timeout 300 python bench.py --config cfg3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg3_2p20.json; cut -c1-200 $OUT/bench_cfg3_2p20.json
timeout 300 python bench.py --config cfg4 --batch-log2 19 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_cfg4_2p19.json; cut -c1-200 $OUT/bench_cfg4_2p19.json
timeout 600 python scripts/train_bench.py 2>&1 | grep "^NSF\|^MAF" | tee $OUT/train.txt
timeout 300 python scripts/train_profile.py 2>&1 | grep -v amdgpu | tee $OUT/train_profile.txt | tail -3
timeout 200 python scripts/split_ablate.py run 2>&1 | grep ARX | tee $OUT/split_ablations.txt
bash scripts/train_trace.sh final_$TAG/train_trace > $OUT/train_trace.txt 2>&1; head -3 $OUT/train_trace.txt
bash scripts/gpu_profile.sh $TAG > $OUT/gpu_profile.log 2>&1; tail -3 $OUT/gpu_profile.log
bash scripts/cfg5_profile.sh > $OUT/cfg5_profile.log 2>&1; cut -c1-300 gpurun_out/prof_cfg5/bench_cfg5_2p19.json
