#!/bin/bash
# round 5, session 14: four lanes share one copy of the weights -- default cache policy for the decode GEMMs' weight requests instead of
# non-temporal ones while batches are in flight?
set -u
R=$(pwd); O=$R/gpurun_out/r05_s14; mkdir -p $O
for V in 0 1 0 1; do
  IVG_DECODE_W_SHARED=$V timeout 400 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs --only-lanes > $O/bench_w$V.json 2> $O/bench_w$V.err
  python - <<PY
import json
d=[json.loads(l) for l in open('$O/bench_w$V.json') if l.startswith('{')][0]
r=d['roofline_in_flight']
print('w_shared=$V: value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), '| in flight', round(r['achieved']/1e3,2), 'TB/s, phase', round(r['rollout_phase_ms'],1), 'ms, gemm us', [round(p['decode_gemm_mean_launch_us'],1) for p in r['per_lane']], 'attn us', [round(p['decode_attn_mean_launch_us'],1) for p in r['per_lane']])
PY
done
echo done > $O/done.txt
