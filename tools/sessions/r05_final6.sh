#!/bin/bash
# round-5 closing session at HEAD (after the snake MFMA order in conv3x3 / conv3x3w / gemm256l): A/B of the order on the decode stage is
# not possible after the fact (no switch: same results by construction) -- full GPU suite, smoke, the driver's bench command
set -u
R=$(pwd); O=$R/gpurun_out/r05_final6; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python tools/conv_ab.py 896 64 policy > $O/conv_ab_896.txt 2>&1; grep "narrow" $O/conv_ab_896.txt | cut -c1-80
rm -f $R/gpurun_out/r03_parity_margins.jsonl $R/gpurun_out/r03_bf16_deviations.jsonl
timeout 1800 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -5 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python - <<PY
import json
d=[json.loads(l) for l in open('$O/bench_n1.json') if l.startswith('{')][0]
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), '| single', round(d['single_lane']['value'],1), '| fp32', round(d['fp32_mode']['value'],1), '| x3', round(d['compliant_mode']['value'],1), d['compliant_mode'].get('lanes_in_flight',{}).get('value'))
for r in [d['roofline']] + d['roofline_other']: print(r['kernel'][:40], 'frac', round(r['frac'],3), 'sustained', round(r.get('frac_of_sustained',0),3), 'ms/step', round(r['kernel_ms_per_step'],1))
for k,v in d['other_configs'].items(): print(k, round(v['value'],1), round(v['lanes_in_flight']['value'],1))
print('stages', {k: round(v,1) for k,v in d['stage_ms'].items() if k.endswith('_ms')}, 'in flight', round(d['roofline_in_flight']['frac'],3), 'cpu', d['cpu_baseline']['value'])
PY
echo done > $O/done.txt
