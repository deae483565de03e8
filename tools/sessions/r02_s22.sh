#!/bin/bash
# round-2 GPU session 22: decode attention prologue (append at the end, first key round unpredicated, requests before conversions)
# and kernarg preload (-mllvm -amdgpu-kernarg-preload-count=16): tests with the new library, then A/B of three builds
set -u
O=gpurun_out/r02_s22; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_ops.txt 2>&1; tail -3 $O/pytest_ops.txt
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_callers.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_models.txt 2>&1; tail -3 $O/pytest_models.txt
cp ivideogpt_amd/lib/libivg.so /tmp/libivg_new.so
for v in new nopl base new nopl base; do
  case $v in new) cp /tmp/libivg_new.so ivideogpt_amd/lib/libivg.so;; nopl) cp ivideogpt_amd/lib/alt/libivg_nopl.so ivideogpt_amd/lib/libivg.so;; base) cp ivideogpt_amd/lib/alt/libivg_base.so ivideogpt_amd/lib/libivg.so;; esac
  echo "== $v" >> $O/quick.txt; timeout 300 python tools/quick_bench.py --iters 5 >> $O/quick.txt 2>&1
done
cp /tmp/libivg_new.so ivideogpt_amd/lib/libivg.so
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
echo done > $O/done.txt
