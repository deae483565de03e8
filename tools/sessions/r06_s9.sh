#!/bin/bash
# round 6, session 9: shared-prefix attention on the matrix cores (prefix_attn_kernel) -- op test vs fp64, model tests, VP2-shaped call
# with the kernel on / off; gemm256l row ranges beyond 2 GiB (test + the 256x256 decode stage)
set -u
R=$(pwd); O=$R/gpurun_out/r06_s9; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_shared.py -q -x -p no:cacheprovider --tb=short > $O/pytest_shared.txt 2>&1
tail -15 $O/pytest_shared.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider --tb=short -k "gemm256" > $O/pytest_gemm256.txt 2>&1
tail -5 $O/pytest_gemm256.txt
echo "vp2 mfma default: $(timeout 300 python tools/vp2_bench.py 200 4 2>&1 | tail -1)"
echo "vp2 mfma off:     $(IVG_DEV=1 IVG_SHARED_MFMA_MIN=0 timeout 300 python tools/vp2_bench.py 200 4 2>&1 | tail -1)"
echo "vp2 n=64 mfma on:  $(timeout 300 python tools/vp2_bench.py 64 4 2>&1 | tail -1)"
echo "vp2 n=64 mfma off: $(IVG_DEV=1 IVG_SHARED_MFMA_MIN=0 timeout 300 python tools/vp2_bench.py 64 4 2>&1 | tail -1)"
echo "vp2 n=16 mfma on:  $(timeout 300 python tools/vp2_bench.py 16 6 2>&1 | tail -1)"
echo "vp2 n=16 mfma off: $(IVG_DEV=1 IVG_SHARED_MFMA_MIN=0 timeout 300 python tools/vp2_bench.py 16 6 2>&1 | tail -1)"
echo "256x256 decode: $(timeout 300 python tools/quick_bench.py --decode-only --iters 4 --res 256 --batch 16 2>&1 | tail -1)"
echo done > $O/done.txt
