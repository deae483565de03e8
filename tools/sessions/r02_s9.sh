#!/bin/bash
# round-2 GPU session 9: eager decode steps by default + GroupNorm statistics fused into the conv3x3 epilogue
set -u
O=gpurun_out/r02_s9; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -k "conv3x3 or groupnorm or conv_modes" > $O/pytest_ops.txt 2>&1
tail -6 $O/pytest_ops.txt
for e in "X=1" "IVG_GN_FUSE=0" "IVG_GRAPH=1"; do
  echo "== $e" >> $O/quick.txt; env $e timeout 300 python tools/quick_bench.py --iters 3 >> $O/quick.txt 2>&1
done
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --deselect tests/test_gpu_ops.py > $O/pytest_rest.txt 2>&1
tail -12 $O/pytest_rest.txt
echo done > $O/done.txt
