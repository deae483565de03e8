#!/bin/bash
# round-2 GPU session 2: second-generation decode GEMM -- parity, tile sweep, rollout A/B
set -u
O=gpurun_out/r02_s2; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "decode_gemm or skinny" > $O/pytest_dg.txt 2>&1
tail -3 $O/pytest_dg.txt
timeout 900 python tools/dgemm_sweep.py > $O/sweep_small.txt 2>&1
tail -2 $O/sweep_small.txt
for cfg in "IVG_DG=0" "IVG_DG=1" "IVG_DG=1 IVG_ATTN_NT=1" "IVG_DG=1 IVG_ATTN_NT=1 IVG_CHAINS=2"; do
  echo "== $cfg" >> $O/quick.txt
  env $cfg timeout 300 python tools/quick_bench.py --iters 3 >> $O/quick.txt 2>&1
done
cat $O/quick.txt | grep -E "==|pred_frames"
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -5 $O/pytest_all.txt
echo done > $O/done.txt
