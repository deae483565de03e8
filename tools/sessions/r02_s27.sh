#!/bin/bash
# round-2 GPU session 27: last sanity of the committed state -- op-level suite and smoke() on the library built from HEAD
set -u
O=gpurun_out/r02_s27; mkdir -p $O
export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_ops.txt 2>&1; tail -2 $O/pytest_ops.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
echo done > $O/done.txt
