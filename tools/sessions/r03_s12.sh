#!/bin/bash
# round-3 GPU session 12: L2 warm-up with default-policy (not non-temporal) weight requests in the consumer: FETCH_SIZE and time
set -u
R=$(pwd); O=$R/gpurun_out/r03_s12; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for nt in 1 0; do
  IVG_DG3_NT=$nt WARM=1 GEN=3 IVG_DG3_ALL=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex 'dg3_kernel|dgemm_kernel' -d /tmp/prof_n$nt -o p --output-format csv -- $R/tools/ubench/bin/dgemm_phase small 64 > $O/run_n$nt.log 2>&1
  F=$(find /tmp/prof_n$nt -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/fetch_n$nt.json > $O/fetch_n$nt.txt 2>&1)
  echo "== IVG_DG3_NT=$nt WARM=1"; head -5 $O/fetch_n$nt.txt
done
cd $R
for e in "IVG_DG3_NT=1 WARM=1" "IVG_DG3_NT=0 WARM=1" "IVG_DG3_NT=0 WARM=0" "IVG_DG3_NT=1 WARM=0"; do
  echo "== $e"; env $e GEN=3 IVG_DG3_ALL=1 timeout 100 tools/ubench/bin/dgemm_phase small 64 | head -1
done 2>&1 | tee $O/time.txt
echo done > $O/done.txt
