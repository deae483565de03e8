#!/bin/bash
# two batches in flight: do the single-chain latency measures (weight warm-up, deeper value prefetch, third-generation GEMM) still pay?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r03_s25.txt; : > $O
B="python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-profile --lanes 2"
run() { echo "== $1" >> $O; shift; env "$@" 2>>gpurun_out/r03_s25.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step; single', round(d.get('single_lane',{}).get('value',0),1))" >> $O; }
run "default"            X=1 $B
run "IVG_DG3_WARM=0"     IVG_DG3_WARM=0 $B
run "IVG_ATTN_PRE2=1"    IVG_ATTN_PRE2=1 $B
run "IVG_DG3=0"          IVG_DG3=0 $B
run "default"            X=1 $B
cat $O
