#!/bin/bash
# round-5: the driver's bench command once more at HEAD (another box: the spread of a power-capped workload)
set -u
R=$(pwd); O=$R/gpurun_out/r05_final5; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python - <<PY
import json
d=[json.loads(l) for l in open('$O/bench_n1.json') if l.startswith('{')][0]
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), '| single', round(d['single_lane']['value'],1), '| fp32', round(d['fp32_mode']['value'],1), '| x3', round(d['compliant_mode']['value'],1), d['compliant_mode'].get('lanes_in_flight',{}).get('value'))
for k,v in d['other_configs'].items(): print(k, round(v['value'],1), round(v['lanes_in_flight']['value'],1))
print('stages', {k: round(v,1) for k,v in d['stage_ms'].items() if k.endswith('_ms')}, 'in flight', round(d['roofline_in_flight']['frac'],3))
PY
echo done > $O/done.txt
