#!/bin/bash
# round 4, session 3: x3 parity tests; more lanes / hardware queues; CU-split experiment; compliant-mode bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_s3; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q --tb=short -p no:cacheprovider > $O/pytest_x3.txt 2>&1
tail -8 $O/pytest_x3.txt
B="python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile"
R=$O/lanes.txt; : > $R
run() { echo "== $1" >> $R; shift; timeout 300 env "$@" 2>>$O/lanes.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stage_ms']; sl = d.get('single_lane', {})
        print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step | single', round(sl.get('value',0),1), 'median ms', round(sl.get('ms_per_step_median',0),2), '| stages', round(s['encode_ms'],1), round(s['rollout_ms'],1), round(s['decode_ms'],1))" >> $R; }
run "lanes3"                          X=1 $B --lanes 3
run "lanes4"                          X=1 $B --lanes 4
run "lanes4 hwq8"                     GPU_MAX_HW_QUEUES=8 $B --lanes 4
run "lanes5 hwq8"                     GPU_MAX_HW_QUEUES=8 $B --lanes 5
run "lanes6 hwq8"                     GPU_MAX_HW_QUEUES=8 $B --lanes 6
run "lanes3 hwq8"                     GPU_MAX_HW_QUEUES=8 $B --lanes 3
run "lanes3 cusplit128 block hwq8"    GPU_MAX_HW_QUEUES=8 $B --lanes 3 --cu-split 128
run "lanes3 cusplit128 interleave hwq8" GPU_MAX_HW_QUEUES=8 IVG_CU_SPLIT_MODE=interleave $B --lanes 3 --cu-split 128
run "lanes4 cusplit128 interleave hwq8" GPU_MAX_HW_QUEUES=8 IVG_CU_SPLIT_MODE=interleave $B --lanes 4 --cu-split 128
run "lanes3 cusplit160 interleave hwq8" GPU_MAX_HW_QUEUES=8 IVG_CU_SPLIT_MODE=interleave $B --lanes 3 --cu-split 160
run "lanes3 cusplit96 interleave hwq8"  GPU_MAX_HW_QUEUES=8 IVG_CU_SPLIT_MODE=interleave $B --lanes 3 --cu-split 96
cat $R
timeout 400 python bench.py --steps 6 --warmup 1 --lanes 2 --no-cpu-baseline --no-profile > $O/bench_modes.json 2> $O/bench_modes.err
python -c "
import json
d=[json.loads(l) for l in open('$O/bench_modes.json') if l.startswith('{')][0]
print('headline', round(d['value'],1)); print('fp32_mode', d.get('fp32_mode')); print('compliant_mode', d.get('compliant_mode'))"
timeout 200 python bench.py --steps 3 --warmup 1 --lanes 1 --decode-dtype x3 --llm-dtype x3 --no-cpu-baseline --no-fp32-mode --no-profile > $O/x3_stage.json 2>> $O/bench_modes.err
python -c "
import json
d=[json.loads(l) for l in open('$O/x3_stage.json') if l.startswith('{')][0]; print('x3 mode', round(d['value'],1), d['stage_ms'])"
grep -i "error\|Traceback" -A8 $O/lanes.err $O/bench_modes.err | head -40
