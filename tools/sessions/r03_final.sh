#!/bin/bash
# round-3 final GPU session: full suite, smoke, the bench line, kernel trace / PMC traffic / MFMA + LDS counters of the bench command at HEAD
set -u
R=$(pwd)
O=$R/gpurun_out/r03_final; mkdir -p $O
export TMPDIR=/tmp
rm -f $R/gpurun_out/r03_parity_margins.jsonl $R/gpurun_out/r03_bf16_deviations.jsonl
timeout 1800 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -6 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
tail -c 1500 $O/bench_n1.json
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt --output-format csv -- $BENCH > $O/bench_under_trace.json 2> $O/trace.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && head -80 "$ST" > $O/bench_kernel_stats.csv
[ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 7 > $O/kernel_trace_summary.txt 2>&1
[ -n "$KT" ] && python $R/tools/trace_classes.py "$KT" 7 $O/kernel_trace_classes.json "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode" > $O/kernel_trace_classes.txt 2>&1
PM="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex 'decode_attn|conv3x3|igemm_kernel|gemm256|dgemm|dg3_kernel' -d /tmp/prof_$C -o p --output-format csv -- $PM > $O/pmc_$C.log 2>&1
  F=$(find /tmp/prof_$C -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_$C.json > $O/pmc_$C.txt 2>&1)
done
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-include-regex 'conv3x3|gemm256|igemm_kernel|xattn|flash_prefill' -d /tmp/prof_mfma -o p --output-format csv -- $PM > $O/pmc_mfma.log 2>&1
F=$(find /tmp/prof_mfma -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_mfma.json > $O/pmc_mfma.txt 2>&1)
cd $R
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/pmc_traffic.json "python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile" > $O/pmc_traffic.txt 2>&1
python tools/pmc_mfma_table.py $O/pmc_mfma.json > $O/pmc_mfma_table.txt 2>&1
cat $O/pmc_traffic.txt; head -30 $O/pmc_mfma_table.txt; tail -3 $O/kernel_trace_summary.txt; cat $O/kernel_trace_classes.txt
for c in 3 4 5; do
  timeout 600 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode > $O/bench_config$c.json 2> $O/bench_config$c.err
done
timeout 600 python tools/mbrl_bench.py 16 12 > $O/mbrl.txt 2>&1; tail -4 $O/mbrl.txt
echo done > $O/done.txt
