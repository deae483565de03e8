#!/bin/bash
# round 4, session 1: tests of the refactor; occupancy-capped conv + phase gate A/B; fp32-mode breakdown
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_s1; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_evaluate.py tests/test_gpu_edges.py -x -q --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
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
run "lanes2 cap lds76 gate"           IVG_CONV_CAP=1 IVG_DECODE_LDS_KB=76 $B --lanes 2 --conv-gate 1
run "lanes3 cap lds76 gate"           IVG_CONV_CAP=1 IVG_DECODE_LDS_KB=76 $B --lanes 3 --conv-gate 1
run "lanes3 cap lds160 gate"          IVG_CONV_CAP=1 $B --lanes 3 --conv-gate 1
run "lanes3 nocap gate"               X=1 $B --lanes 3 --conv-gate 1
run "lanes3 cap lds76 nogate"         IVG_CONV_CAP=1 IVG_DECODE_LDS_KB=76 $B --lanes 3
run "lanes4 cap lds76 gate"           IVG_CONV_CAP=1 IVG_DECODE_LDS_KB=76 $B --lanes 4 --conv-gate 1
run "lanes1 nocap lds76 (rollout cost of the LDS budget)"  IVG_DECODE_LDS_KB=76 $B --lanes 1
cat $R
# fp32 mode: stage split + kernel stats
timeout 300 python bench.py --steps 3 --warmup 1 --lanes 1 --decode-dtype fp32 --llm-dtype fp32 --no-cpu-baseline --no-fp32-mode --no-profile > $O/fp32_mode.json 2> $O/fp32_mode.err
python -c "
import json
d=[json.loads(l) for l in open('$O/fp32_mode.json') if l.startswith('{')][0]; print('fp32 mode', round(d['value'],1), d['stage_ms'])"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_f32 -o f32 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --lanes 1 --decode-dtype fp32 --llm-dtype fp32 --no-cpu-baseline --no-fp32-mode --no-profile > $GRAFT_REPO_ROOT/$O/fp32_trace_run.json 2> $GRAFT_REPO_ROOT/$O/fp32_trace.err
ST=$(find /tmp/prof_f32 -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && head -50 "$ST" > $GRAFT_REPO_ROOT/$O/fp32_kernel_stats.csv
cd $GRAFT_REPO_ROOT; head -30 $O/fp32_kernel_stats.csv | cut -c1-200
grep -i "error\|Traceback" -A5 $O/lanes.err | head -20
