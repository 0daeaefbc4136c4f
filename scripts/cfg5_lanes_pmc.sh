#!/bin/bash
# PMC passes over the cfg5 layer script (second-generation last-layer kernel); usage: cfg5_lanes_pmc.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
export PMC_SETS="lds:SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE;sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU;tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum;tcp:TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"
LOG2N=${LOG2N:-17} REPS=2 bash scripts/pmc_run.sh ${1:-cfg5lanes_pmc} "lanes_kernel\|linear_bf16_kernel" -- python $ROOT/scripts/cfg5_layer_time.py
