#!/bin/bash
# round-2 GPU session 18: (a) does the half-line shape of the conv3x3 weight requests bound the kernel?  (probe: same bytes as whole lines)
# (b) decode attention after restoring predicated loads
set -u
O=gpurun_out/r02_s18; mkdir -p $O
export TMPDIR=/tmp
for shape in "64 128 128 0" "32 256 256 0" "16 512 512 0" "32 256 256 1"; do
  for wl in 0 1; do
    IVG_C3_WLINE=$wl timeout 120 python tools/conv_bench.py $shape 2>&1 | tail -1 | sed "s/^/wline=$wl /" >> $O/conv_bench.txt
  done
done
cat $O/conv_bench.txt
timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu --tb=short -p no:cacheprovider -x -k "llama or generate or rollout or decode or medium or logits" > $O/pytest_models.txt 2>&1; tail -3 $O/pytest_models.txt
for e in "IVG_X=1" "IVG_X=2"; do
  echo "== $e" >> $O/quick.txt; env $e timeout 300 python tools/quick_bench.py --iters 5 >> $O/quick.txt 2>&1
done
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
echo done > $O/done.txt
