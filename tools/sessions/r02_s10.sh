#!/bin/bash
# round-2 GPU session 10: kernel-level A/B of the GroupNorm-statistics fusion (rocprofv3 kernel trace of one pass each)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_s10; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for v in 1 0; do
  IVG_GN_FUSE=$v timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_gn$v -o t --output-format csv -- python $R/tools/quick_bench.py --iters 2 > $O/quick_fuse$v.txt 2>&1
  KT=$(find /tmp/prof_gn$v -name "*kernel_trace.csv" | head -1)
  [ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 3 | grep -E "kernel |conv3x3|gn_|kernel time" > $O/trace_fuse$v.txt 2>&1
  grep pred_frames $O/quick_fuse$v.txt | cut -c1-150
done
cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -k "epilogue_groupnorm" > $O/pytest_gn.txt 2>&1; tail -3 $O/pytest_gn.txt
echo done > $O/done.txt
