#!/bin/bash
# round-3 GPU session 11: does the L2 warm-up turn the consumer's weight fetches into L2 hits?  FETCH_SIZE per launch of the harness chain
set -u
R=$(pwd); O=$R/gpurun_out/r03_s11; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for w in 0 1; do
  WARM=$w GEN=3 IVG_DG3_ALL=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'dg3_kernel|dgemm_kernel' -d /tmp/prof_w$w -o p --output-format csv -- $R/tools/ubench/bin/dgemm_phase small 64 > $O/run_w$w.log 2>&1
  F=$(find /tmp/prof_w$w -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/fetch_w$w.json > $O/fetch_w$w.txt 2>&1)
  echo "== WARM=$w"; head -8 $O/fetch_w$w.txt
done
echo done > $O/done.txt
