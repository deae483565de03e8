#!/bin/bash
# round 6, session 17: 256 x 256-tile split-bf16 ("x3") GEMM for the prompt pass / dense 1x1 layers of the 1e-3-compliant mode
set -u
R=$(pwd); O=$R/gpurun_out/r06_s17; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider --tb=short -k "gemm256 or tail or gemm_bias" > $O/pytest_ops.txt 2>&1
tail -5 $O/pytest_ops.txt
timeout 1200 python -m pytest tests/test_gpu_x3.py -q -x -p no:cacheprovider --tb=short > $O/pytest_x3.txt 2>&1
tail -4 $O/pytest_x3.txt
for arm in 1 0 1 0; do
echo "x3 mode, IVG_GEMM256X3=$arm: $(IVG_DEV=1 IVG_GEMM256X3=$arm timeout 300 python tools/quick_bench.py --dec x3 --llm x3 --iters 3 2>&1 | tail -1)"
done
echo done > $O/done.txt
