#!/bin/bash
# PMC passes over zk_gemm_f16x2 (run on the GPU box): matrix-pipe busy share, LDS conflicts, L2 traffic
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export PMC_SETS="sq:SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS;l2:TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum;mem:FETCH_SIZE WRITE_SIZE;wait:SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU"
bash $ROOT/scripts/pmc_run.sh r06t/pmc_gemm gemm_half -- python $ROOT/scripts/gemm_half_pmc_workload.py
cat $ROOT/gpurun_out/r06t/pmc_gemm/*.summary.txt > $ROOT/gpurun_out/r06t/pmc_gemm_half.txt
