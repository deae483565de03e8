#!/bin/bash
# round 5, session 10: gemm256l<128> for the 128-channel 1x1 shortcuts -- op tests, decode-stage trace, then the FULL GPU suite and smoke
set -u
R=$(pwd); O=$R/gpurun_out/r05_s10; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider --tb=short -k "gemm256 or conv" > $O/pytest_ops.txt 2>&1
tail -3 $O/pytest_ops.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt --output-format csv -- python $R/tools/quick_bench.py --iters 5 > $O/quick.txt 2> $O/trace.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 6 > $O/trace.txt 2>&1
tail -1 $O/quick.txt | cut -c1-130
grep "igemm\|gemm256\|gn_apply\|conv3x3_kernelIDF16bLi64\|^kernel" $O/trace.txt | cut -c1-150
rm -rf /tmp/prof_kt
cd $R
timeout 300 python tools/quick_bench.py --iters 5 --res 256 --batch 16 > $O/quick256.txt 2>&1; tail -1 $O/quick256.txt
timeout 1800 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -6 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
echo done > $O/done.txt
