#!/bin/bash
# round 6, session 25: kernel trace of the 1e-3-compliant (x3) mode at HEAD (24-bit K / V cache, x3 tile GEMM)
set -u
R=$(pwd); O=$R/gpurun_out/r06_s25; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt --output-format csv -- python $R/tools/quick_bench.py --dec x3 --llm x3 --iters 2 > $O/q.txt 2> $O/trace.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 3 > $O/trace_x3.txt 2>&1
head -45 $O/trace_x3.txt | cut -c1-150
tail -1 $O/q.txt
echo done > $O/done.txt
