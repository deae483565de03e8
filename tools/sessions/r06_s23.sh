#!/bin/bash
# round 6, session 23: is the device ever idle inside the four-lane loop?
set -u
R=$(pwd); O=$R/gpurun_out/r06_s23; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt --output-format csv -- python $R/bench.py --only-lanes --steps 32 --warmup 4 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile > $O/bench.json 2> $O/trace.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python $R/tools/idle_report.py "$KT" 0.45 0.95 > $O/idle.txt 2>&1
cat $O/idle.txt
tail -c 600 $O/bench.json | head -c 300
echo done > $O/done.txt
