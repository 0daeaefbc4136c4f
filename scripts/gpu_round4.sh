#!/bin/bash
# GPU call 4 of round 3: cfg5 last-layer kernel variants, then the rocprofv3 evidence of the headline (kernel trace + PMC passes) and of the training step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
bash scripts/cfg5_variants.sh cfg5var spread spread_prio epi3 spread_epi3 prio
bash scripts/gpu_profile.sh r03 2>&1 | tail -40
bash scripts/train_trace.sh train_trace_r03 2>&1 | tail -30
