#!/bin/bash
# round 4, session 8: MFMA-bound phases on co-resident (small) kernels while several batches are in flight
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_s8; mkdir -p $O; export TMPDIR=/tmp
B="python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile --no-other-configs"
R=$O/lanes.txt; : > $R
run() { echo "== $1" >> $R; shift; timeout 300 env "$@" 2>>$O/lanes.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stage_ms']; sl = d.get('single_lane', {})
        print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step | single', round(sl.get('value',0),1), 'median ms', round(sl.get('ms_per_step_median',0),2), '| stages', round(s['encode_ms'],1), round(s['rollout_ms'],1), round(s['decode_ms'],1))" >> $R; }
run "4 lanes, prompt GEMMs on the 128x128 implicit GEMM (IVG_GEMM256=0)"   IVG_GEMM256=0 $B
run "4 lanes, 3x3 convolutions on the implicit GEMM (IVG_CONV3X3=0)"        IVG_CONV3X3=0 $B
run "4 lanes, separate GroupNorm apply pass (IVG_GN_APPLY_FUSE=0)"          IVG_GN_APPLY_FUSE=0 $B
run "4 lanes, one-pass prompt attention off (IVG_FLASH_PREFILL=0)"          IVG_FLASH_PREFILL=0 $B
cat $R
grep -i "error\|Traceback" -A8 $O/lanes.err | head -20
