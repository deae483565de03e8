#!/bin/bash
# round 4, session 4: new tests; LDS budget of the decode GEMMs with several batches in flight; compliant mode with dg3 X3; kernel trace of 3 lanes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_s4; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_x3.py tests/test_gpu_edges.py tests/test_gpu_callers.py -x -q --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
B="python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile"
R=$O/lanes.txt; : > $R
run() { echo "== $1" >> $R; shift; timeout 300 env "$@" 2>>$O/lanes.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stage_ms']; sl = d.get('single_lane', {})
        print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step | single', round(sl.get('value',0),1), 'median ms', round(sl.get('ms_per_step_median',0),2), '| stages', round(s['encode_ms'],1), round(s['rollout_ms'],1), round(s['decode_ms'],1))" >> $R; }
run "lanes3 lds160"     X=1 $B --lanes 3
run "lanes3 lds100"     IVG_DECODE_LDS_KB=100 $B --lanes 3
run "lanes3 lds76"      IVG_DECODE_LDS_KB=76 $B --lanes 3
run "lanes3 lds52"      IVG_DECODE_LDS_KB=52 $B --lanes 3
run "lanes4 lds160"     X=1 $B --lanes 4
run "lanes4 lds76"      IVG_DECODE_LDS_KB=76 $B --lanes 4
run "lanes4 lds52"      IVG_DECODE_LDS_KB=52 $B --lanes 4
run "lanes3 dg3off (gen2 decode GEMMs)"  IVG_DG3=0 $B --lanes 3
cat $R
timeout 500 python bench.py --steps 6 --warmup 1 --lanes 3 --no-cpu-baseline --no-profile > $O/bench_modes.json 2> $O/bench_modes.err
python -c "
import json
d=[json.loads(l) for l in open('$O/bench_modes.json') if l.startswith('{')][0]
print('headline', round(d['value'],1)); print('fp32_mode', d.get('fp32_mode',{}).get('value')); print('compliant_mode', d.get('compliant_mode'))"
timeout 200 python bench.py --steps 3 --warmup 1 --lanes 1 --decode-dtype x3 --llm-dtype x3 --no-cpu-baseline --no-fp32-mode --no-profile > $O/x3_stage.json 2>> $O/bench_modes.err
python -c "
import json
d=[json.loads(l) for l in open('$O/x3_stage.json') if l.startswith('{')][0]; print('x3 mode', round(d['value'],1), d['stage_ms'])"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_l3 -o l3 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --lanes 3 --no-cpu-baseline --no-fp32-mode --no-profile > $GRAFT_REPO_ROOT/$O/trace_run.json 2> $GRAFT_REPO_ROOT/$O/trace.err
KT=$(find /tmp/prof_l3 -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_l3 -name "*kernel_stats.csv" | head -1)
cd $GRAFT_REPO_ROOT
[ -n "$ST" ] && head -40 "$ST" > $O/lanes3_kernel_stats.csv
[ -n "$KT" ] && python tools/sessions/overlap_report.py "$KT" > $O/lanes3_overlap.txt 2>&1
tail -25 $O/lanes3_overlap.txt
grep -i "error\|Traceback" -A8 $O/lanes.err $O/bench_modes.err | head -30
