#!/bin/bash
# engine on the caller's stream: tests, lanes 2 / 3 at the driver's K / W, config 5 lanes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r03_s23.txt; : > $O
timeout 600 python -m pytest tests/test_gpu_evaluate.py tests/test_gpu_edges.py -q -m gpu -x 2>&1 | tail -3 >> $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-mode --no-profile"
run() { echo "== $1" >> $O; shift; env "$@" 2>>gpurun_out/r03_s23.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step; single', round(d.get('single_lane',{}).get('value',0),1))" >> $O; }
run "lanes 3"           X=1 $B --lanes 3
run "lanes 2"           X=1 $B --lanes 2
run "config 5 lanes 3"  X=1 $B --config 5 --lanes 3
run "config 3 lanes 3"  X=1 $B --config 3 --lanes 3
cat $O
