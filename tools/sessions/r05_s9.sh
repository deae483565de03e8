#!/bin/bash
# round 5, session 9: decoder tail as one fused conv3x3 launch, 1x1 shortcuts with Cout % 256 == 0 on gemm256l -- model-level parity
# tests, then the decode stage with each switch on / off (kernel trace)
set -u
R=$(pwd); O=$R/gpurun_out/r05_s9; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_gpu_callers.py -q -x -p no:cacheprovider --tb=short > $O/pytest_models.txt 2>&1
tail -5 $O/pytest_models.txt
cd /tmp
for V in "1 1" "0 1" "1 0" "0 0"; do
  set -- $V
  IVG_TAIL_FUSE=$1 IVG_SHORTCUT_GEMM256=$2 timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt --output-format csv -- python $R/tools/quick_bench.py --iters 5 > $O/quick_t$1_s$2.txt 2> $O/trace.err
  KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1)
  [ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 6 > $O/trace_t$1_s$2.txt 2>&1
  echo "tail_fuse=$1 shortcut_gemm256=$2: $(tail -1 $O/quick_t$1_s$2.txt | cut -c1-120)"
  grep "igemm\|gemm256\|gn_apply\|conv3x3_kernelIDF16bLi64\|^kernel" $O/trace_t$1_s$2.txt | cut -c1-150
  rm -rf /tmp/prof_kt
done
IVG_TAIL_FUSE=1 IVG_SHORTCUT_GEMM256=1 timeout 300 python $R/tools/quick_bench.py --iters 5 --res 256 --batch 16 > $O/quick256_on.txt 2>&1; tail -1 $O/quick256_on.txt
IVG_TAIL_FUSE=0 IVG_SHORTCUT_GEMM256=0 timeout 300 python $R/tools/quick_bench.py --iters 5 --res 256 --batch 16 > $O/quick256_off.txt 2>&1; tail -1 $O/quick256_off.txt
echo done > $O/done.txt
