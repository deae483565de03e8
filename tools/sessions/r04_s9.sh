#!/bin/bash
# round 4, session 9: collectives beside four batches in flight (1-rank RCCL group); the x3 two-lane figure inside the full command; new tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_s9; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_evaluate.py -q --tb=short -p no:cacheprovider -k "x3 or in_flight or rccl or lanes" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
B="python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile --no-other-configs"
R=$O/lanes.txt; : > $R
run() { echo "== $1" >> $R; shift; timeout 300 env "$@" 2>>$O/lanes.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stage_ms']; sl = d.get('single_lane', {})
        print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step | single', round(sl.get('value',0),1), 'median ms', round(sl.get('ms_per_step_median',0),2), '| stages', round(s['encode_ms'],1), round(s['rollout_ms'],1), round(s['decode_ms'],1))" >> $R; }
run "4 lanes, no process group"                                              X=1 $B
run "4 lanes, 1-rank RCCL group, gathers by the gatherer thread"             IVG_FORCE_COLLECTIVE=1 $B
run "4 lanes, 1-rank RCCL group, gathers by the lanes (Turnstile)"           IVG_FORCE_COLLECTIVE=1 $B --gather-mode lanes
run "4 lanes, 1-rank RCCL group, gatherer thread, 16 hardware queues"        IVG_FORCE_COLLECTIVE=1 GPU_MAX_HW_QUEUES=16 $B
run "4 lanes, 1-rank RCCL group, lanes, 16 hardware queues"                  IVG_FORCE_COLLECTIVE=1 GPU_MAX_HW_QUEUES=16 $B --gather-mode lanes
cat $R
timeout 500 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-profile --no-other-configs > $O/bench_modes.json 2> $O/bench_modes.err
python -c "
import json
d=[json.loads(l) for l in open('$O/bench_modes.json') if l.startswith('{')][0]
print('headline', round(d['value'],1)); print('fp32_mode', d.get('fp32_mode',{}).get('value')); print('compliant_mode', d.get('compliant_mode'))"
grep -i "error\|Traceback" -A8 $O/lanes.err $O/bench_modes.err | head -30
