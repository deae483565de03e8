#!/bin/bash
# round 6, session 4: shared-context rollouts / decodes (ivg_generate_shared, ivg_detokenize_shared) -- parity tests
set -u
R=$(pwd); O=$R/gpurun_out/r06_s4; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_shared.py -q -x -p no:cacheprovider --tb=short > $O/pytest_shared.txt 2>&1
tail -30 $O/pytest_shared.txt
timeout 900 python -m pytest tests/test_gpu_callers.py tests/test_gpu_edges.py -q -x -p no:cacheprovider --tb=short > $O/pytest_callers.txt 2>&1
tail -5 $O/pytest_callers.txt
echo done > $O/done.txt
