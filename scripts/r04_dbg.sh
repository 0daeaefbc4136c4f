ZUKO_AMD_ARX2_QB=8 ZUKO_AMD_ARX2_FILL=2 ZUKO_AMD_CACHE_DIR=/root/repo/variants/8x2x1 ZUKO_AMD_JIT=0 python scripts/arx2_check.py --time-only --label x 2>&1 | tail -15
