#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/prof_f32; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
pmc() { local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -o p -- python $ROOT/scripts/f32_gemm_probe.py > $OUT/pmc_$name.log 2>&1
  find $OUT/pmc_$name -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$name.csv \;
  python $ROOT/scripts/summarize_pmc.py $OUT/pmc_$name.csv | grep linear_f32 | tee $OUT/pmc_$name.summary.txt; }
pmc sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH
rm -rf $OUT/pmc_*/
