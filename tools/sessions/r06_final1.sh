#!/bin/bash
# round 6, closing session 1 at HEAD: full GPU suite, smoke, the driver's bench command, the 1-rank RCCL line, MBRL step path
set -u
R=$(pwd); O=$R/gpurun_out/r06_final1; mkdir -p $O; export TMPDIR=/tmp
rm -f $R/gpurun_out/r03_parity_margins.jsonl $R/gpurun_out/r03_bf16_deviations.jsonl
timeout 2400 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -6 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python - <<PY
import json
d=[json.loads(l) for l in open('$O/bench_n1.json') if l.startswith('{')][0]
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), '| single', round(d['single_lane']['value'],1), '| fp32', round(d['fp32_mode']['value'],1), '| x3', round(d['compliant_mode']['value'],1), d['compliant_mode'].get('lanes_in_flight',{}).get('value'))
for r in [d['roofline']] + d['roofline_other']: print(r['kernel'][:40], 'frac', round(r['frac'],3), 'profiler', round(r.get('frac_profiler',0),3), 'sustained', round(r.get('frac_of_sustained',0),3), 'ms/step', round(r['kernel_ms_per_step'],1), 'traffic', r.get('traffic'), r.get('mean_launch_us_by_kind'))
r=d['roofline_in_flight']; print('in flight', round(r['achieved'],0), 'GB/s', round(r['frac'],3), 'phase ms', round(r['rollout_phase_ms'],1), r['per_lane'][0].get('decode_gemm_mean_launch_us_by_kind'))
for k,v in d['other_configs'].items(): print(k, round(v['value'],1), v.get('lanes_in_flight',{}).get('value'), v.get('stage_ms'), (v.get('roofline') or {}).get('frac'))
for k,v in d['shared_context'].items(): print(k, v.get('shared_context'), v.get('plain'), v.get('speedup'))
print('stages', d['stage_ms'], 'cpu', d['cpu_baseline']['value'])
PY
timeout 300 env IVG_FORCE_COLLECTIVE=1 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile > $O/bench_rccl_1rank.json 2> $O/bench_rccl.err
python -c "
import json
d=[json.loads(l) for l in open('$O/bench_rccl_1rank.json') if l.startswith('{')][0]; print('4 lanes with the per-step all-gather through RCCL (1 rank, gatherer thread):', round(d['value'],1), 'f/s')"
timeout 300 python tools/mbrl_bench.py 16 12 > $O/mbrl_rollout.txt 2>&1; tail -2 $O/mbrl_rollout.txt
grep -i "error\|Traceback" -A6 $O/bench_n1.err $O/bench_rccl.err | head -20
echo done > $O/done.txt
