#!/bin/bash
# round 6, session 2: the launch-profile parity tests (decode_lds_kb 0 / 40) + the retired-switch build on the GPU
set -u
R=$(pwd); O=$R/gpurun_out/r06_s2; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_x3.py tests/test_gpu_evaluate.py -q -x -p no:cacheprovider --tb=short > $O/pytest_models.txt 2>&1
tail -6 $O/pytest_models.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider --tb=short > $O/pytest_ops.txt 2>&1
tail -6 $O/pytest_ops.txt
echo done > $O/done.txt
