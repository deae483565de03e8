#!/bin/bash
set -u
O=gpurun_out/r02_s14; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -x -k "conv" > $O/pytest_conv.txt 2>&1; tail -8 $O/pytest_conv.txt
IVG_C3_PRE=0 timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=line -p no:cacheprovider -k "conv3x3 or conv_modes" > $O/pytest_conv_nopre.txt 2>&1; tail -3 $O/pytest_conv_nopre.txt
for shape in "64 128 128 0" "64 256 128 0" "32 256 256 0" "32 512 256 0" "16 512 512 0" "16 512 512 1" "32 256 256 1" "64 128 128 0 128 fp32" "32 256 256 0 128 fp32"; do
  for pre in 1 0; do
    IVG_C3_PRE=$pre timeout 120 python tools/conv_bench.py $shape 2>&1 | tail -1 | sed "s/^/pre=$pre /" >> $O/conv_bench.txt
  done
done
cat $O/conv_bench.txt
for e in "IVG_C3_PRE=1" "IVG_C3_PRE=0" "IVG_C3_PRE=1"; do
  echo "== $e" >> $O/quick.txt; env $e timeout 300 python tools/quick_bench.py --iters 5 >> $O/quick.txt 2>&1
done
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu --tb=short -p no:cacheprovider -x > $O/pytest_models.txt 2>&1; tail -5 $O/pytest_models.txt
echo done > $O/done.txt
