mkdir -p gpurun_out/r04
for v in "$@"; do
  qb=${v%%x*}
  ZUKO_AMD_ARX2_QB=$qb ZUKO_AMD_CACHE_DIR=/root/repo/variants/$v ZUKO_AMD_JIT=0 python scripts/arx2_check.py --time-only --label $v 2>&1 | grep label >> gpurun_out/r04/arx2_t.txt
done
tail -${#@} gpurun_out/r04/arx2_t.txt
