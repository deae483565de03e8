#!/bin/bash
# round-4 closing session at HEAD: full GPU suite, smoke, the driver's bench command, the same under a 1-rank RCCL group, configs 3-5 with lanes
set -u
R=$(pwd); O=$R/gpurun_out/r04_final2; mkdir -p $O; export TMPDIR=/tmp
rm -f $R/gpurun_out/r03_parity_margins.jsonl $R/gpurun_out/r03_bf16_deviations.jsonl
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -8 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python -c "
import json
d=[json.loads(l) for l in open('$O/bench_n1.json') if l.startswith('{')][0]
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), '| single', round(d['single_lane']['value'],1), '| fp32', round(d['fp32_mode']['value'],1), '| x3', round(d['compliant_mode']['value'],1), d['compliant_mode'].get('lanes_in_flight'))
for r in [d['roofline']] + d['roofline_other']: print(r['kernel'][:40], 'frac', round(r['frac'],3), 'stamps', round(r.get('frac_stamps',0),3), 'ms/step', round(r['kernel_ms_per_step'],1), 'traffic', r.get('traffic'))
print({k: round(v['value'],1) for k, v in d['other_configs'].items()}, 'cpu', d['cpu_baseline']['value'])"
timeout 300 env IVG_FORCE_COLLECTIVE=1 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile > $O/bench_rccl_1rank.json 2> $O/bench_rccl.err
python -c "
import json
d=[json.loads(l) for l in open('$O/bench_rccl_1rank.json') if l.startswith('{')][0]; print('4 lanes with the per-step all-gather through RCCL (1 rank, gatherer thread):', round(d['value'],1), 'f/s')"
for c in 3 4 5; do
  timeout 400 python bench.py --config $c --steps 8 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile > $O/bench_config$c.json 2> $O/bench_config$c.err
  python -c "
import json
d=[json.loads(l) for l in open('$O/bench_config$c.json') if l.startswith('{')][0]; print('config $c: 4 lanes', round(d['value'],1), 'f/s; one lane', round(d['single_lane']['value'],1))"
done
grep -i "error\|Traceback" -A6 $O/bench_n1.err $O/bench_rccl.err | head -20
echo done > $O/done.txt
