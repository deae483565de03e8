#!/bin/bash
# round 6, session 19: 24-bit K / V cache of the x3 rollout
set -u
R=$(pwd); O=$R/gpurun_out/r06_s19; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_x3.py -q -x -p no:cacheprovider --tb=short > $O/pytest_x3.txt 2>&1
tail -15 $O/pytest_x3.txt
for arm in 1 0 1 0; do
echo "x3 mode, IVG_KV24=$arm: $(IVG_DEV=1 IVG_KV24=$arm timeout 300 python tools/quick_bench.py --dec x3 --llm x3 --iters 3 2>&1 | tail -1)"
done
echo done > $O/done.txt
