ZUKO_AMD_JIT=0 python scripts/arx3_trace.py 2>&1 | tail -3 | cut -c1-300
