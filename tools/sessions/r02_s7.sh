#!/bin/bash
# round-2 GPU session 7: why does the rollout run faster under rocprofv3?  Runtime settings A/B on the kernel boundary and the rollout
set -u
O=gpurun_out/r02_s7; mkdir -p $O
export TMPDIR=/tmp
U=tools/ubench/bin/decode_ubench
echo "== plain" >> $O/boundary.txt; $U b >> $O/boundary.txt 2>&1
for e in "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_INTERRUPT=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "GPU_MAX_HW_QUEUES=1" "DEBUG_HIP_GRAPH_DOT_PRINT=0 AMD_DIRECT_DISPATCH=0" "HSA_NO_SCRATCH_RECLAIM=1" "ROC_SIGNAL_POOL_SIZE=4096" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "HIP_LAUNCH_BLOCKING=0 ROC_USE_FGS_KERNARG=0" "HSA_XNACK=0"; do
  echo "== $e" >> $O/boundary.txt; env $e $U b >> $O/boundary.txt 2>&1
done
echo "== under rocprofv3 --kernel-trace" >> $O/boundary.txt
(cd /tmp && rocprofv3 --kernel-trace -d /tmp/prof_b -o b -- $GRAFT_REPO_ROOT/$U b) >> $O/boundary.txt 2>&1
grep -E "==|T1|T2" $O/boundary.txt
for e in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_INTERRUPT=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "ROC_ACTIVE_WAIT_TIMEOUT=1000"; do
  echo "== $e" >> $O/quick.txt; env $e timeout 300 python tools/quick_bench.py --iters 3 >> $O/quick.txt 2>&1
done
echo "== under rocprofv3 --kernel-trace" >> $O/quick.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_q -o q -- python $GRAFT_REPO_ROOT/tools/quick_bench.py --iters 3) >> $O/quick.txt 2>&1
echo "== under rocprofv3 --hip-runtime-trace only" >> $O/quick.txt
(cd /tmp && timeout 300 rocprofv3 --hip-runtime-trace -d /tmp/prof_q2 -o q -- python $GRAFT_REPO_ROOT/tools/quick_bench.py --iters 3) >> $O/quick.txt 2>&1
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
env | grep -iE "^(HSA|HIP|ROC|GPU|AMD)" > $O/env.txt
echo done > $O/done.txt
