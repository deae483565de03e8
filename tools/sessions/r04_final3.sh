#!/bin/bash
# round-4 closing session at HEAD (after the last prune: old gemm256 kernel, dead launchers): full GPU suite, smoke, the driver's bench command
set -u
R=$(pwd); O=$R/gpurun_out/r04_final3; mkdir -p $O; export TMPDIR=/tmp
rm -f $R/gpurun_out/r03_parity_margins.jsonl $R/gpurun_out/r03_bf16_deviations.jsonl
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -8 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python -c "
import json
d=[json.loads(l) for l in open('$O/bench_n1.json') if l.startswith('{')][0]
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), '| single', round(d['single_lane']['value'],1), '| fp32', round(d['fp32_mode']['value'],1), '| x3', round(d['compliant_mode']['value'],1), d['compliant_mode'].get('lanes_in_flight'))
for r in [d['roofline']] + d['roofline_other']: print(r['kernel'][:40], 'frac', round(r['frac'],3), 'stamps', round(r.get('frac_stamps',0),3), 'ms/step', round(r['kernel_ms_per_step'],1))
print({k: round(v['value'],1) for k, v in d['other_configs'].items()}, 'cpu', d['cpu_baseline']['value'], 'stages', d['stage_ms']['encode_ms'], d['stage_ms']['rollout_ms'], d['stage_ms']['decode_ms'])"
grep -i "error\|Traceback" -A6 $O/bench_n1.err | head -20
echo done > $O/done.txt
