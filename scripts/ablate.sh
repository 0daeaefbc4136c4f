#!/bin/bash
# ablation sweep of the fused kernel (profiling aid): dbg bit0 = no univariate math, bit1 = no DMA/barrier, bit2 = no LDS reads
run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],3), 'ms/launch', round(d['value']/1e6,2), 'M samples/s')"; }
for d in 0 7; do echo -n "skip dbg=$d  "; ZUKO_AMD_AR_DEBUG=$d run; done
for d in 0 7; do echo -n "DENSE dbg=$d  "; ZUKO_AMD_AR_DENSE=1 ZUKO_AMD_AR_DEBUG=$d run; done
