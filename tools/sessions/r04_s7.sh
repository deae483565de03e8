#!/bin/bash
# round 4, session 7: graph replay with 4-6 batches in flight; compliant (x3) mode with two batches in flight; new tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_s7; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_evaluate.py -q --tb=short -p no:cacheprovider -k "in_flight or lanes" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
B="python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile --no-other-configs"
R=$O/lanes.txt; : > $R
run() { echo "== $1" >> $R; shift; timeout 300 env "$@" 2>>$O/lanes.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stage_ms']; sl = d.get('single_lane', {})
        print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step | single', round(sl.get('value',0),1), 'median ms', round(sl.get('ms_per_step_median',0),2), '| stages', round(s['encode_ms'],1), round(s['rollout_ms'],1), round(s['decode_ms'],1))" >> $R; }
run "4 lanes graph replay"     IVG_GRAPH=1 $B --lanes 4
run "5 lanes graph replay"     IVG_GRAPH=1 $B --lanes 5
run "6 lanes graph replay"     IVG_GRAPH=1 $B --lanes 6
run "5 lanes hwq 16"           GPU_MAX_HW_QUEUES=16 $B --lanes 5
run "x3 mode, 2 lanes, no lane switches"   X=1 $B --lanes 2 --decode-dtype x3 --llm-dtype x3 --lane-switches none --steps 6
run "x3 mode, 2 lanes, in-flight switches" X=1 $B --lanes 2 --decode-dtype x3 --llm-dtype x3 --steps 6
run "x3 mode, 3 lanes, in-flight switches" X=1 $B --lanes 3 --decode-dtype x3 --llm-dtype x3 --steps 6
cat $R
grep -i "error\|Traceback" -A8 $O/lanes.err | head -30
