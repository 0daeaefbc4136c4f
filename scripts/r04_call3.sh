mkdir -p gpurun_out/r04
: > gpurun_out/r04/arx2_abl.txt
for v in 8x2x1 8x2x2 8x2x4 8x2x5 8x2x7 8x2x8; do
  ZUKO_AMD_ARX2_QB=8 ZUKO_AMD_ARX2_FILL=2 ZUKO_AMD_CACHE_DIR=/root/repo/variants/$v ZUKO_AMD_JIT=0 python scripts/arx2_check.py --time-only --label $v 2>&1 | grep label >> gpurun_out/r04/arx2_abl.txt
done
cat gpurun_out/r04/arx2_abl.txt
