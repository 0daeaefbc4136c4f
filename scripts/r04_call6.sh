mkdir -p gpurun_out/r04
ZUKO_AMD_CACHE_DIR=/root/repo/variants/8x2xARX2_TRACE ZUKO_AMD_JIT=0 python scripts/arx2_trace.py > gpurun_out/r04/arx2_trace.json 2> gpurun_out/r04/arx2_trace.err
tail -2 gpurun_out/r04/arx2_trace.err
ZUKO_AMD_CACHE_DIR=/root/repo/variants/8x2 ZUKO_AMD_JIT=0 python scripts/arx2_check.py --time-only --label 8x2 2>&1 | grep label > gpurun_out/r04/arx2_t.txt
cat gpurun_out/r04/arx2_t.txt
