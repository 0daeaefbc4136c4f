mkdir -p gpurun_out/r04
ZUKO_AMD_STATIC_CXXFLAGS="-DARX3_ONLY -DARX3_TRACE -DARX3_TRACE" ZUKO_AMD_CACHE_DIR=/root/repo/variants/8x4xARX3_TRACE ZUKO_AMD_JIT=0 python scripts/arx3_trace.py > gpurun_out/r04/arx3_trace.json 2> gpurun_out/r04/arx3_trace.err
tail -3 gpurun_out/r04/arx3_trace.err
