#!/bin/bash
# round-2 GPU session 5: full GPU suite + the profiles of the bench command (kernel trace / stats, PMC traffic passes)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_s5; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -12 $O/pytest_all.txt
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt --output-format csv -- $BENCH > $O/bench_under_trace.json 2> $O/trace.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && head -60 "$ST" > $O/bench_kernel_stats.csv
[ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 5 > $O/kernel_trace_summary.txt 2>&1
PM="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile"
for C in FETCH_SIZE WRITE_SIZE; do
  IVG_NO_GRAPH=1 timeout 900 rocprofv3 --pmc $C --kernel-include-regex 'decode_attn|conv3x3|igemm_kernel|gemm256|dgemm' -d /tmp/prof_$C -o p --output-format csv -- $PM > $O/pmc_$C.log 2>&1
  F=$(find /tmp/prof_$C -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_$C.json > $O/pmc_$C.txt 2>&1)
done
cd $R
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/pmc_traffic.json "IVG_NO_GRAPH=1 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile" > $O/pmc_traffic.txt 2>&1
cat $O/pmc_traffic.txt; tail -3 $O/kernel_trace_summary.txt
echo done > $O/done.txt
