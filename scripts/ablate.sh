#!/bin/bash
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^wave|metric" | sed 's/{"metric.*avg_launch_ms": \([0-9.]*\).*/launch ms \1/' | sort | uniq -c | sort -rn | head -2; }
ZUKO_AMD_AR_DEBUG=8 run; ZUKO_AMD_AR_DEBUG=0 run
