#!/bin/bash
# round-3 closing GPU session at HEAD (two batches in flight by default): full suite, smoke, the driver's bench command, kernel trace of the default command
set -u
R=$(pwd)
O=$R/gpurun_out/r03_final2; mkdir -p $O
export TMPDIR=/tmp
rm -f $R/gpurun_out/r03_parity_margins.jsonl $R/gpurun_out/r03_bf16_deviations.jsonl
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -6 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
tail -c 1200 $O/bench_n1.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt --output-format csv -- python $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile > $O/bench_under_trace.json 2> $O/trace.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && head -60 "$ST" > $O/lanes2_kernel_stats.csv
[ -n "$KT" ] && python $R/tools/sessions/overlap_report.py "$KT" > $O/lanes2_overlap.txt 2>&1
tail -5 $O/lanes2_overlap.txt
echo done > $O/done.txt
