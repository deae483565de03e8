#!/bin/bash
# lanes: stream priority of the rollout engines, number of hardware queues
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r03_s20.txt; : > $O
B="python bench.py --steps 16 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile"
run() { echo "== $1" >> $O; shift; env "$@" 2>>gpurun_out/r03_s20.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step; single', round(d.get('single_lane',{}).get('value',0),1))" >> $O; }
run "lanes 2 base"            X=1 $B --lanes 2
run "lanes 2 llm prio -1"     IVG_LLM_STREAM_PRIORITY=-1 $B --lanes 2
run "lanes 2 hwq 8"           GPU_MAX_HW_QUEUES=8 $B --lanes 2
run "lanes 2 hwq 8 + prio"    GPU_MAX_HW_QUEUES=8 IVG_LLM_STREAM_PRIORITY=-1 $B --lanes 2
run "lanes 3 hwq 12 + prio"   GPU_MAX_HW_QUEUES=12 IVG_LLM_STREAM_PRIORITY=-1 $B --lanes 3
run "lanes 2 hwq 2"           GPU_MAX_HW_QUEUES=2 $B --lanes 2
cat $O
