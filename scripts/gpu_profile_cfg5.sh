#!/bin/bash
# cfg5 (bf16 NSF-1024) under rocprofv3: kernel-trace stats + LDS / SQ counter passes of `bench.py --config cfg5` at 2^19 rows.  usage: gpu_profile_cfg5.sh <tag>
set -u
TAG=${1:-r05}; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/prof_cfg5_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
BENCH="python $ROOT/bench.py --config cfg5 --gpus 1 --batch-log2 19 --warmup 1 --steps 2 --no-bin-report --no-side-configs --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -8 $OUT/kernel_stats.csv
pmc() { local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o bench -- $BENCH > $OUT/pmc_$name.log 2>&1
  find $OUT/pmc_$name -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$name.csv \;
  python $ROOT/scripts/summarize_pmc.py $OUT/pmc_$name.csv | grep "linear_bf16" | tee $OUT/pmc_$name.summary.txt; }
pmc lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU
rm -rf $OUT/trace/*/*.db $OUT/pmc_*/ $OUT/*.csv.bak 2>/dev/null; rm -f $OUT/pmc_*.csv; du -sh $OUT
