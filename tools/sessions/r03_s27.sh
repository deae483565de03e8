#!/bin/bash
# lanes started together every round (lock-step): does it make three lanes reproducible?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r03_s27.txt; : > $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-mode --no-profile"
run() { echo "== $1" >> $O; shift; env "$@" 2>>gpurun_out/r03_s27.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step; single', round(d.get('single_lane',{}).get('value',0),1))" >> $O; }
run "lanes 3 lockstep"   IVG_LANE_LOCKSTEP=1 $B --lanes 3
run "lanes 3 free"       X=1 $B --lanes 3
run "lanes 3 lockstep"   IVG_LANE_LOCKSTEP=1 $B --lanes 3
run "lanes 2 lockstep"   IVG_LANE_LOCKSTEP=1 $B --lanes 2
run "lanes 4 lockstep"   IVG_LANE_LOCKSTEP=1 $B --lanes 4
cat $O
