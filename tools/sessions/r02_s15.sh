#!/bin/bash
set -u
O=gpurun_out/r02_s15; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -x -k "decode_gemm or skinny or sampler" > $O/pytest_ops.txt 2>&1; tail -5 $O/pytest_ops.txt
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu --tb=short -p no:cacheprovider -x -k "llama or generate or rollout or decode or medium or bf16 or logits or eval" > $O/pytest_models.txt 2>&1; tail -5 $O/pytest_models.txt
for e in "IVG_X=1" "IVG_X=2"; do
  echo "== $e" >> $O/quick.txt; env $e timeout 300 python tools/quick_bench.py --iters 5 >> $O/quick.txt 2>&1
done
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
echo done > $O/done.txt
