#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the DEFAULT command (four batches in flight, 40 KiB decode-GEMM footprint while they run)
set -u
R=$(pwd); O=$R/gpurun_out/r04_pmc_default; mkdir -p $O; export TMPDIR=/tmp
CMD="--steps 2 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 700 rocprofv3 --pmc $C --kernel-include-regex 'decode_attn|conv3x3|igemm_kernel|gemm256|dgemm|dg3_kernel' -d /tmp/prof_$C -o p --output-format csv -- python $R/bench.py $CMD > $O/pmc_$C.log 2>&1
  F=$(find /tmp/prof_$C -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_$C.json > $O/pmc_$C.txt 2>&1)
  rm -rf /tmp/prof_$C
done
cd $R
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/pmc_traffic_default.json "python bench.py $CMD" > $O/pmc_traffic.txt 2>&1
cat $O/pmc_traffic.txt; grep -h "dgemm\|dg3" $O/pmc_FETCH_SIZE.txt | head -20 | cut -c1-200
