#!/bin/bash
# round 4, session 6: full GPU suite; capped conv beside small-footprint decode GEMMs; the driver's bench command
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_s6; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/r03_parity_margins.jsonl gpurun_out/r03_bf16_deviations.jsonl
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -12 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
B="python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile --no-other-configs"
R=$O/lanes.txt; : > $R
run() { echo "== $1" >> $R; shift; timeout 300 env "$@" 2>>$O/lanes.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stage_ms']; sl = d.get('single_lane', {})
        print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step | single', round(sl.get('value',0),1), 'median ms', round(sl.get('ms_per_step_median',0),2), '| stages', round(s['encode_ms'],1), round(s['rollout_ms'],1), round(s['decode_ms'],1))" >> $R; }
run "default (4 lanes, in-flight switches)"     X=1 $B
run "4 lanes, conv cap, gate"                   IVG_CONV_CAP=1 $B --conv-gate 1
run "4 lanes, conv cap, no gate"                IVG_CONV_CAP=1 $B
run "4 lanes, no lane switches"                 X=1 $B --lane-switches none
cat $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python -c "
import json
d=[json.loads(l) for l in open('$O/bench_n1.json') if l.startswith('{')][0]
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],2)); print('single', d['single_lane']); print('stages', d['stage_ms'])
print('fp32', d.get('fp32_mode',{}).get('value')); print('compliant', d.get('compliant_mode')); print('other', d.get('other_configs')); print('cpu', d.get('cpu_baseline'))
print('roofline', {k: d['roofline'][k] for k in ('kernel','frac','achieved','kernel_ms_per_step')})"
grep -i "error\|Traceback" -A8 $O/lanes.err $O/bench_n1.err | head -30
