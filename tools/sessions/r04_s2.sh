#!/bin/bash
# round 4, session 2: lanes A/B again with the lane streams on distinct hardware queues; x3 conv op test
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_s2; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q --tb=short -p no:cacheprovider -k "x3 or gemm256 or sampler" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
B="python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile"
R=$O/lanes.txt; : > $R
run() { echo "== $1" >> $R; shift; timeout 300 env "$@" 2>>$O/lanes.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stage_ms']; sl = d.get('single_lane', {})
        print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step | single', round(sl.get('value',0),1), 'median ms', round(sl.get('ms_per_step_median',0),2), '| stages', round(s['encode_ms'],1), round(s['rollout_ms'],1), round(s['decode_ms'],1))" >> $R; }
run "lanes2 baseline"                 X=1 $B --lanes 2
run "lanes2 gate"                     X=1 $B --lanes 2 --conv-gate 1
run "lanes3 nocap nogate"             X=1 $B --lanes 3
run "lanes3 nocap gate"               X=1 $B --lanes 3 --conv-gate 1
run "lanes2 cap lds76 gate"           IVG_CONV_CAP=1 IVG_DECODE_LDS_KB=76 $B --lanes 2 --conv-gate 1
run "lanes3 cap lds76 gate"           IVG_CONV_CAP=1 IVG_DECODE_LDS_KB=76 $B --lanes 3 --conv-gate 1
run "lanes3 cap lds160 gate"          IVG_CONV_CAP=1 $B --lanes 3 --conv-gate 1
run "lanes3 cap lds76 nogate"         IVG_CONV_CAP=1 IVG_DECODE_LDS_KB=76 $B --lanes 3
run "lanes4 cap lds76 gate"           IVG_CONV_CAP=1 IVG_DECODE_LDS_KB=76 $B --lanes 4 --conv-gate 1
run "lanes3 cap lds76 gate nowarm"    IVG_CONV_CAP=1 IVG_DECODE_LDS_KB=76 IVG_DG3_WARM=0 $B --lanes 3 --conv-gate 1
run "lanes2 nowarm"                   IVG_DG3_WARM=0 $B --lanes 2
cat $R
grep -i "error\|Traceback" -A5 $O/lanes.err | head -20
