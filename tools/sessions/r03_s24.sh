#!/bin/bash
# conv3x3: GroupNorm transform interleaved with the MFMAs (GNA == 2) -- correctness, A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r03_s24.txt; : > $O
timeout 500 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "groupnorm or conv3x3" 2>&1 | tail -3 >> $O
timeout 300 python -m pytest tests/test_gpu_models.py -q -m gpu -x -k "fuse or golden or GroupNorm" 2>&1 | tail -3 >> $O
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-mode"
run() { echo "== $1" >> $O; shift; env "$@" 2>>gpurun_out/r03_s24.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c=[r for r in d['roofline_other'] if 'conv3x3' in r['kernel']][0]
        print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step; single', round(d.get('single_lane',{}).get('value',0),1), 'stage', {k:round(v,1) for k,v in d['stage_ms'].items() if k.endswith('_ms')}, 'conv3x3 ms/step', round(c['kernel_ms_per_step'],2), 'frac', round(c['frac'],3))" >> $O; }
run "IL=1 lanes 1"  IVG_GNA_IL=1 $B --lanes 1
run "IL=0 lanes 1"  IVG_GNA_IL=0 $B --lanes 1
run "IL=1 lanes 1"  IVG_GNA_IL=1 $B --lanes 1
run "IL=0 lanes 1"  IVG_GNA_IL=0 $B --lanes 1
run "IL=1 lanes 2"  IVG_GNA_IL=1 $B --lanes 2 --steps 16 --no-profile
run "IL=0 lanes 2"  IVG_GNA_IL=0 $B --lanes 2 --steps 16 --no-profile
cat $O
