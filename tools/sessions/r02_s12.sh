#!/bin/bash
set -u
O=gpurun_out/r02_s12; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -k "conv3x3 or groupnorm" > $O/pytest_gn.txt 2>&1; tail -5 $O/pytest_gn.txt
for e in "IVG_GN_APPLY_FUSE=1" "IVG_GN_APPLY_FUSE=0" "IVG_GN_APPLY_FUSE=1" "IVG_GN_APPLY_FUSE=0"; do
  echo "== $e" >> $O/quick.txt; env $e timeout 300 python tools/quick_bench.py --iters 5 >> $O/quick.txt 2>&1
done
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --deselect tests/test_gpu_ops.py > $O/pytest_rest.txt 2>&1
tail -8 $O/pytest_rest.txt
echo done > $O/done.txt
