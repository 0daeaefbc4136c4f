#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
bash scripts/cfg5_variants.sh cfg5var2 nospread rot rot_dma1
for n in rot rot_dma1; do echo "== bf16 tests on lib_$n"; ZUKO_AMD_LIB=$ROOT/scripts/probes/ab/lib_$n.so timeout 600 python -m pytest tests/test_gpu_flows.py -m gpu -q -k "bf16" 2>&1 | tail -3; done
echo "== bf16 tests on the default library"; timeout 600 python -m pytest tests/test_gpu_flows.py -m gpu -q -k "bf16" 2>&1 | tail -3
