#!/bin/bash
# round 5, session 7: the conv3x3 kernels IN the decode stage (residuals, output statistics, a hot chip) -- kernel trace of the
# quick stage bench with the 256-pixel kernel only / the persistent kernel by policy / everywhere
set -u
R=$(pwd); O=$R/gpurun_out/r05_s7; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for W in 0 1 2; do
  IVG_CONV_WIDE=$W timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_kt$W -o kt --output-format csv -- python $R/tools/quick_bench.py --iters 3 > $O/quick_w$W.txt 2> $O/trace_w$W.err
  KT=$(find /tmp/prof_kt$W -name "*kernel_trace.csv" | head -1)
  [ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 4 > $O/trace_w$W.txt 2>&1
  tail -1 $O/quick_w$W.txt
  grep "conv3x3\|^kernel" $O/trace_w$W.txt | cut -c1-170
  rm -rf /tmp/prof_kt$W
done
echo done > $O/done.txt
