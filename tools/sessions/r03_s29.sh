#!/bin/bash
# two lanes: is the host's launch path (two threads launching eagerly) what holds the pair at 355 ms?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r03_s29.txt; : > $O
B="python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile --lanes 2"
run() { echo "== $1" >> $O; shift; env "$@" 2>>gpurun_out/r03_s29.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step; single', round(d.get('single_lane',{}).get('value',0),1))" >> $O; }
run "IVG_GRAPH=1"              IVG_GRAPH=1 $B
run "AMD_DIRECT_DISPATCH=0"    AMD_DIRECT_DISPATCH=0 $B
run "default"                  X=1 $B
cat $O
