mkdir -p gpurun_out/r04
ZUKO_AMD_STATIC_CXXFLAGS="-DARX3_ONLY -DARX3_TRACE -DARX3_TRACE" ZUKO_AMD_CACHE_DIR=/root/repo/variants/8x4xARX3_TRACE ZUKO_AMD_JIT=0 python scripts/arx3_trace.py > gpurun_out/r04/arx3_trace.json 2> gpurun_out/r04/arx3_trace.err
tail -2 gpurun_out/r04/arx3_trace.err
ZUKO_AMD_STATIC_CXXFLAGS="-DARX3_ONLY" ZUKO_AMD_CACHE_DIR=/root/repo/variants/8x4 ZUKO_AMD_JIT=0 python scripts/arx3_check.py --time-only --label 8x4 2>&1 | grep label | cut -c1-500
