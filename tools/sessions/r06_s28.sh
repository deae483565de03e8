#!/bin/bash
# round 6, session 28: x3 GroupNorm-fused convolutions with v_rcp_f32 in the SiLU (the operand is cut to 2^-17 right after)
set -u
R=$(pwd); O=$R/gpurun_out/r06_s28; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider --tb=short -k "x3" > $O/pytest_ops.txt 2>&1
tail -3 $O/pytest_ops.txt
timeout 1200 python -m pytest tests/test_gpu_x3.py -q -x -p no:cacheprovider --tb=short -k "detokenize or decode" > $O/pytest_x3.txt 2>&1
tail -3 $O/pytest_x3.txt
for i in 1 2; do
echo "64x64 x3 decode: $(timeout 300 python tools/quick_bench.py --decode-only --dec x3 --iters 5 2>&1 | tail -1 | cut -c1-110)"
done
echo "256x256 x3 decode: $(timeout 300 python tools/quick_bench.py --decode-only --dec x3 --iters 3 --res 256 --batch 16 2>&1 | tail -1 | cut -c1-110)"
echo done > $O/done.txt
