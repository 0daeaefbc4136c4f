#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of the default bench workload.
# Usage: bash scripts/gpu_profile.sh <tag>       outputs -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-bin-report --no-side-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -12 $OUT/kernel_stats.csv
pmc() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o bench -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-bin-report --no-side-configs > $OUT/pmc_$name.log 2>&1
  find $OUT/pmc_$name -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$name.csv \;
  python $ROOT/scripts/summarize_pmc.py $OUT/pmc_$name.csv | tee $OUT/pmc_$name.summary.txt
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pmc act SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_MFMA
pmc ifetch SQ_IFETCH SQ_IFETCH_LEVEL SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_CYCLES SQ_BUSY_CU_CYCLES
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
rm -rf $OUT/trace/*/*.db $OUT/pmc_*/ 2>/dev/null
du -sh $OUT
