#!/bin/bash
set -u
O=gpurun_out/r02_s13; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -k "cross_attention or batched_attention or softmax" > $O/pytest_xattn.txt 2>&1; tail -5 $O/pytest_xattn.txt
timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu --tb=short -p no:cacheprovider -k "bf16 or tokenizer or tokenize" > $O/pytest_tok.txt 2>&1; tail -5 $O/pytest_tok.txt
for e in "IVG_FLASH_XATT=1" "IVG_FLASH_XATT=0" "IVG_FLASH_XATT=1" "IVG_FLASH_XATT=0"; do
  echo "== $e" >> $O/quick.txt; env $e timeout 300 python tools/quick_bench.py --iters 5 >> $O/quick.txt 2>&1
done
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
echo done > $O/done.txt
