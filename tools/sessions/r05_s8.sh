#!/bin/bash
# round 5, session 8: per-engine decode-GEMM LDS budget + in-flight roofline pass + other_configs with lanes -- tests and the driver's command
set -u
R=$(pwd); O=$R/gpurun_out/r05_s8; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_evaluate.py -q -x -p no:cacheprovider --tb=short -k "lds_budget or in_flight or two_lanes or two_engines" > $O/pytest_eval.txt 2>&1
tail -8 $O/pytest_eval.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err ) 2> $O/bench_time.txt; tail -3 $O/bench_time.txt; tail -3 $O/bench_n1.err
python - <<'PY' "$O"
import json, sys
O = sys.argv[1]
d = [json.loads(l) for l in open(O + '/bench_n1.json') if l.startswith('{')][0]
print('value', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), '| single', round(d['single_lane']['value'], 1), '| fp32', round(d['fp32_mode']['value'], 1),
      '| x3', round(d['compliant_mode']['value'], 1), d['compliant_mode'].get('lanes_in_flight'))
for k, v in d['other_configs'].items():
    print(k, round(v['value'], 1) if v.get('value') else v, v.get('lanes_in_flight'))
print('stages', d['stage_ms'])
r = d['roofline_in_flight']
print('in flight', {k: r[k] for k in ('achieved', 'frac', 'rollout_phase_ms', 'decode_attn_GB', 'decode_gemm_weight_GB', 'decode_attn_mean_launch_us_in_flight')}, r['decode_attn_only'])
print('roofline', {k: d['roofline'][k] for k in ('kernel', 'achieved', 'frac', 'kernel_ms_per_step')})
for r in d['roofline_other']:
    print('  other', {k: r.get(k) for k in ('kernel', 'achieved', 'frac', 'kernel_ms_per_step')})
PY
echo done > $O/done.txt
