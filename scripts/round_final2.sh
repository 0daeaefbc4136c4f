#!/bin/bash
# End-of-round re-validation after the last training changes: full GPU suite + the training artefacts (outputs under gpurun_out/final2/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/final2; mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $OUT/pytest_gpu.txt; tail -3 $OUT/pytest_gpu.txt
cp gpurun_out/parity_report.json $OUT/parity_report.json 2>/dev/null
timeout 300 python scripts/train_bench.py 2>&1 | grep "^NSF\|^MAF" | tee $OUT/train.txt
timeout 200 python scripts/train_profile.py 2>&1 | grep -v amdgpu > $OUT/train_profile.txt; tail -2 $OUT/train_profile.txt
timeout 100 python scripts/wgrad_bench.py 2>&1 | grep -v amdgpu | tee $OUT/wgrad_bench.txt
bash scripts/train_trace.sh final2/train_trace > $OUT/train_trace.txt 2>&1; head -2 $OUT/train_trace.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
