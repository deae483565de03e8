#!/bin/bash
# round-5: kernel trace of the LANES-ONLY default command at HEAD (after the batches-in-flight profile)
set -u
R=$(pwd); O=$R/gpurun_out/r05_final4; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
DEF="python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile --only-lanes"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_l4 -o l4 --output-format csv -- $DEF > $O/bench_under_trace_lanes4.json 2> $O/trace4.err
KT=$(find /tmp/prof_l4 -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_l4 -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && head -60 "$ST" > $O/lanes4_kernel_stats.csv
[ -n "$KT" ] && python $R/tools/sessions/overlap_report.py "$KT" > $O/lanes4_overlap.txt 2>&1
tail -20 $O/lanes4_overlap.txt
echo done > $O/done.txt
