#!/bin/bash
# usage: pmc_run.sh <outdir-tag> <kernel-grep> -- <command...> ; runs the listed PMC passes (env PMC_SETS="a:C1 C2;b:C3 ...")
TAG=$1; KGREP=$2; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
IFS=';' read -ra SETS <<< "$PMC_SETS"
for set in "${SETS[@]}"; do
  name=${set%%:*}; ctrs=${set#*:}
  (cd /tmp && rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pmc_$name -o p -- "$@" > $OUT/pmc_$name.log 2>&1)
  find $OUT/pmc_$name -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$name.csv \;
  python $ROOT/scripts/summarize_pmc.py $OUT/pmc_$name.csv | grep "$KGREP" | tee $OUT/pmc_$name.summary.txt
  rm -rf $OUT/pmc_$name
done
