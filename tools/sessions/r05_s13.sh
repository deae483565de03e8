#!/bin/bash
# round 5, session 13: the same 256 trajectories resident as 2 x 128 instead of 4 x 64 (fewer, larger launches per token)?
set -u
R=$(pwd); O=$R/gpurun_out/r05_s13; mkdir -p $O
for V in "128 1" "128 2" "128 3" "96 3" "64 4"; do
  set -- $V
  timeout 400 python bench.py --batch $1 --lanes $2 --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile --only-lanes > $O/bench_b$1_l$2.json 2> $O/bench_b$1_l$2.err
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open('$O/bench_b$1_l$2.json') if l.startswith('{')][0]
    print('batch $1 x lanes $2: value', round(d['value'],1), 'f/s, ms per step (of $1 trajectories)', round(d['ms_per_step'],1), 'stages', {k: round(v,1) for k,v in d['stage_ms'].items() if k.endswith('_ms')})
except Exception as e:
    print('batch $1 lanes $2 failed', e); print(open('$O/bench_b$1_l$2.err').read()[-600:])
PY
done
echo done > $O/done.txt
