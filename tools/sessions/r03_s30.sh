#!/bin/bash
# lanes over ONE copy of the weights in HBM (replica()) vs a packed copy per lane
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r03_s30.txt; : > $O
B="python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile --lanes 2"
run() { echo "== $1" >> $O; shift; env "$@" 2>>gpurun_out/r03_s30.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step; single', round(d.get('single_lane',{}).get('value',0),1))" >> $O; }
run "shared weights (replica)"   X=1 $B
run "a copy per lane"            IVG_LANE_SHARE=0 $B
cat $O; grep -i "error\|Traceback" -A5 gpurun_out/r03_s30.err | head -20
