#!/bin/bash
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^wave|metric" | sed 's/{"metric.*avg_launch_ms": \([0-9.]*\).*/launch ms \1/' | sort | uniq -c | sort -rn | head -3; }
for d in 0 8 9; do echo "== dbg=$d"; ZUKO_AMD_AR_DEBUG=$d run; done
python -m pytest tests/test_gpu_flows.py -m gpu -x -q -k "fused or golden" 2>&1 | tail -2
