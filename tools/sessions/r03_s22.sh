#!/bin/bash
# lanes: hardware-queue count sweep, one stream per lane
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r03_s22.txt; : > $O
B="python bench.py --steps 16 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile"
run() { echo "== $1" >> $O; shift; env "$@" 2>>gpurun_out/r03_s22.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step; single', round(d.get('single_lane',{}).get('value',0),1))" >> $O; }



run "lanes 2 direct"          IVG_LANE_STREAMS=direct $B --lanes 2
run "lanes 3 direct"          IVG_LANE_STREAMS=direct $B --lanes 3
run "lanes 2 direct hwq 2" GPU_MAX_HW_QUEUES=2 IVG_LANE_STREAMS=direct $B --lanes 2
run "lanes 4 direct"          IVG_LANE_STREAMS=direct $B --lanes 4
cat $O
