#!/bin/bash
# round-2 GPU session 20: gemm256 with two K steps per barrier (A/B), new conv3x3 test shapes
set -u
O=gpurun_out/r02_s20; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -x -k "gemm256 or conv3x3 or gemm_bias" > $O/pytest_ops.txt 2>&1; tail -4 $O/pytest_ops.txt
for e in "IVG_G256_PAIR=1" "IVG_G256_PAIR=0" "IVG_G256_PAIR=1" "IVG_G256_PAIR=0"; do
  echo "== $e" >> $O/quick.txt; env $e timeout 300 python tools/quick_bench.py --iters 5 >> $O/quick.txt 2>&1
done
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu --tb=short -p no:cacheprovider -x -k "llama or logits or eval or flash" > $O/pytest_models.txt 2>&1; tail -4 $O/pytest_models.txt
echo done > $O/done.txt
