mkdir -p gpurun_out/r04
python scripts/arx2_check.py --label default > gpurun_out/r04/arx2_check.txt 2>&1
for v in 12x3 8x0; do
  qb=${v%x*}; fill=${v#*x}
  ZUKO_AMD_ARX2_QB=$qb ZUKO_AMD_ARX2_FILL=$fill ZUKO_AMD_CACHE_DIR=/root/repo/variants/$v ZUKO_AMD_JIT=0 python scripts/arx2_check.py --time-only --label $v >> gpurun_out/r04/arx2_check.txt 2>&1
done
tail -40 gpurun_out/r04/arx2_check.txt
