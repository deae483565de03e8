#!/bin/bash
# round 6, session 15: the decoders' tail on a 16-channel instance of the 3x3 kernel (was: 64 channels computed, 3 stored)
set -u
R=$(pwd); O=$R/gpurun_out/r06_s15; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider --tb=short -k "tail or conv3x3 or conv_out or subpixel" > $O/pytest_ops.txt 2>&1
tail -5 $O/pytest_ops.txt
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_gpu_bf16_deviation.py tests/test_gpu_callers.py -q -x -p no:cacheprovider --tb=short -k "not llama and not decode_path and not rollout" > $O/pytest_models.txt 2>&1
tail -4 $O/pytest_models.txt
for i in 1 2; do
echo "64x64 decode: $(timeout 300 python tools/quick_bench.py --decode-only --iters 6 2>&1 | tail -1)"
done
echo "256x256 decode: $(timeout 300 python tools/quick_bench.py --decode-only --iters 4 --res 256 --batch 16 2>&1 | tail -1)"
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt --output-format csv -- python $R/tools/quick_bench.py --decode-only --iters 4 > $O/q.txt 2> $O/trace.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 5 > $O/trace_decode.txt 2>&1
grep "Li16E\|Li64E\|^kernel" $O/trace_decode.txt | cut -c1-150
echo done > $O/done.txt
