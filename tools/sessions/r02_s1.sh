#!/bin/bash
# round-2 GPU session 1: decode-step micro-benchmarks + chain / nt-load A/B runs of the rollout (development aid)
set -u
O=gpurun_out/r02_s1; mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/ubench/bin/decode_ubench > $O/ubench.txt 2>&1; echo "ubench rc $?" >> $O/ubench.txt
python -c "import torch; print(torch.__version__)" > $O/torch.txt 2>&1
for cfg in "IVG_CHAINS=1" "IVG_CHAINS=2" "IVG_CHAINS=4" "IVG_ATTN_NT=1" "IVG_ATTN_NT=1 IVG_CHAINS=2"; do
  echo "== $cfg" >> $O/quick.txt
  env $cfg timeout 300 python tools/quick_bench.py --iters 3 >> $O/quick.txt 2>&1
done
# kernel trace of the two-chain run: do the branches overlap?
cd /tmp && IVG_CHAINS=2 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -o c2 --output-format csv -- python $GRAFT_REPO_ROOT/tools/quick_bench.py --iters 1 > $GRAFT_REPO_ROOT/$O/prof_c2.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_c2 -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python tools/sessions/overlap_report.py "$f" > $O/overlap_c2.txt 2>&1; fi
echo done > $O/done.txt
