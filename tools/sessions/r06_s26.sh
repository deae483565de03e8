#!/bin/bash
# round 6, session 26: warm-up policy of the decode GEMMs in the x3 mode (fp32 weights: twice the bytes the policy was tuned on)
set -u
R=$(pwd); O=$R/gpurun_out/r06_s26; mkdir -p $O; export TMPDIR=/tmp
for arm in "" "IVG_DG3_WARM=0" "IVG_WARM_GATE_UP=1" "" "IVG_DG3_WARM=0"; do
echo "x3 mode [$arm]: $(env IVG_DEV=1 $arm timeout 300 python tools/quick_bench.py --dec x3 --llm x3 --iters 3 2>&1 | tail -1 | cut -c1-140)"
done
echo done > $O/done.txt
