#!/bin/bash
# round-2 GPU session 8: graph replay vs per-node dispatch vs eager launches (the rollout runs faster under rocprofv3 --kernel-trace)
set -u
O=gpurun_out/r02_s8; mkdir -p $O
export TMPDIR=/tmp
for e in "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "IVG_NO_GRAPH=1" "IVG_NO_GRAPH=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "AMD_DIRECT_DISPATCH=0" "IVG_NO_GRAPH=1 HIP_FORCE_DEV_KERNARG=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 IVG_GRAPH_STEPS=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 IVG_GRAPH_STEPS=32"; do
  echo "== $e" >> $O/quick.txt; env $e timeout 300 python tools/quick_bench.py --iters 3 >> $O/quick.txt 2>&1
done
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
echo done > $O/done.txt
