#!/bin/bash
# round 5, session 11: is the four-lane rollout phase bound by the host threads' launch rate?  (host time to enqueue a rollout vs its
# device interval, 1 / 2 / 4 lanes)
set -u
R=$(pwd); O=$R/gpurun_out/r05_s11; mkdir -p $O; export TMPDIR=/tmp
for L in 2 4; do
  timeout 400 python bench.py --lanes $L --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs --only-lanes > $O/bench_l$L.json 2> $O/bench_l$L.err
  python - <<PY
import json
d=[json.loads(l) for l in open('$O/bench_l$L.json') if l.startswith('{')][0]
r=d['roofline_in_flight']
print('lanes $L: value', round(d['value'],1), 'in-flight TB/s', round(r['achieved']/1e3,2), 'phase ms', round(r['rollout_phase_ms'],1))
for p in r['per_lane']: print('   device interval', [round(x,1) for x in p['rollout_interval_ms']], '=', round(p['rollout_interval_ms'][1]-p['rollout_interval_ms'][0],1), 'ms | host enqueue', round(p['rollout_host_enqueue_ms'],1), 'ms | attn us', round(p['decode_attn_mean_launch_us'],1), 'gemm us', round(p['decode_gemm_mean_launch_us'],1))
PY
done
echo done > $O/done.txt
