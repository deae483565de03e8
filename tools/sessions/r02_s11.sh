#!/bin/bash
set -u
O=gpurun_out/r02_s11; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -k "epilogue_groupnorm or conv3x3_halo" > $O/pytest_gn.txt 2>&1; tail -3 $O/pytest_gn.txt
for e in "IVG_GN_FUSE=1" "IVG_GN_FUSE=0" "IVG_GN_FUSE=1" "IVG_GN_FUSE=0"; do
  echo "== $e" >> $O/quick.txt; env $e timeout 300 python tools/quick_bench.py --iters 5 >> $O/quick.txt 2>&1
done
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
echo done > $O/done.txt
