#!/bin/bash
# round-2 GPU session 26: the fused / separate GroupNorm-apply agreement test and the kernel trace of the bench command with the
# end-of-round defaults (GroupNorm apply inside the conv3x3 staging)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_s26; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 300 python -m pytest tests/test_gpu_models.py -q -m gpu --tb=short -p no:cacheprovider -x -k "fused_and_separate or bit_exact or bf16_mode" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt --output-format csv -- $BENCH > $O/bench_under_trace.json 2> $O/trace.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && head -80 "$ST" > $O/bench_kernel_stats.csv
[ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 7 > $O/kernel_trace_summary.txt 2>&1
head -30 $O/kernel_trace_summary.txt | cut -c1-130; tail -1 $O/kernel_trace_summary.txt
echo done > $O/done.txt
