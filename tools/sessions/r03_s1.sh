#!/bin/bash
# round-3 GPU session 1: full-vocabulary parity tests + L2-warm micro-benchmark + decode GEMM phase stamps
set -u
O=gpurun_out/r03_s1; mkdir -p $O
export TMPDIR=/tmp
nproc > $O/nproc.txt
timeout 900 python -m pytest tests/test_gpu_vocab.py -q -m gpu --tb=short -p no:cacheprovider -s > $O/pytest_vocab.txt 2>&1
tail -5 $O/pytest_vocab.txt
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_callers.py -q -m gpu --tb=short -p no:cacheprovider -s -k "full_width or fp32_decode or config1 or kept or mbrl" > $O/pytest_changed.txt 2>&1
tail -5 $O/pytest_changed.txt
timeout 120 tools/ubench/bin/l2warm_ubench > $O/l2warm.txt 2>&1; echo "l2warm rc $?"
cat $O/l2warm.txt
timeout 120 tools/ubench/bin/dgemm_phase small 64 > $O/dgemm_phase_small.txt 2>&1; echo "phase rc $?"
cat $O/dgemm_phase_small.txt
timeout 120 tools/ubench/bin/dgemm_phase medium 64 > $O/dgemm_phase_medium.txt 2>&1
echo done > $O/done.txt
