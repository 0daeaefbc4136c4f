#!/bin/bash
# A/B of the standalone spline kernels: stream kernel (tile-dealing schemes) vs general uni_kernel
for c in 4 -1 0 2 6 8; do echo "== stream kernel, ZUKO_AMD_K1_CHUNK=$c"; ZUKO_AMD_K1_CHUNK=$c bash scripts/k1.sh 2>&1 | grep rqs_; done
echo "== general kernel"; ZUKO_AMD_NO_STREAM=1 bash scripts/k1.sh 2>&1 | grep rqs_
