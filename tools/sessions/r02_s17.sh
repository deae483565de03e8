#!/bin/bash
# round-2 GPU session 17: kernel trace of one pass with the current defaults (where does the tokenizer time go now)
set -u
O=gpurun_out/r02_s17; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt --output-format csv -- python $R/tools/quick_bench.py --iters 3 > $R/$O/quick_under_trace.txt 2>&1
cd $R
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1)
ST=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
python tools/trace_summary.py $KT 4 > $O/trace_summary.txt 2>&1
cp $ST $O/kernel_stats.csv
head -70 $O/trace_summary.txt
echo done > $O/done.txt
