#!/bin/bash
# round-2 GPU session 19: conv3x3 with two steps per barrier (IVG_C3_TPB=2, default) vs one
set -u
O=gpurun_out/r02_s19; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -x -k "conv" > $O/pytest_conv.txt 2>&1; tail -4 $O/pytest_conv.txt
for shape in "64 128 128 0" "64 256 128 0" "32 256 256 0" "32 512 256 0" "16 512 512 0" "16 512 512 1" "32 256 256 1"; do
  for t in 2 1; do
    IVG_C3_TPB=$t timeout 120 python tools/conv_bench.py $shape 2>&1 | tail -1 | sed "s/^/tpb=$t /" >> $O/conv_bench.txt
  done
done
cat $O/conv_bench.txt
for e in "IVG_C3_TPB=2" "IVG_C3_TPB=1" "IVG_C3_TPB=2"; do
  echo "== $e" >> $O/quick.txt; env $e timeout 300 python tools/quick_bench.py --iters 5 >> $O/quick.txt 2>&1
done
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu --tb=short -p no:cacheprovider -x -k "tok or bf16 or 256 or 64" > $O/pytest_models.txt 2>&1; tail -4 $O/pytest_models.txt
echo done > $O/done.txt
