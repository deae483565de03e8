#!/bin/bash
# HEAD sanity: the evaluate / lanes tests, smoke, the default bench command
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r03_s28.txt; : > $O
timeout 400 python -m pytest tests/test_gpu_evaluate.py -q -m gpu 2>&1 | tail -2 >> $O
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> $O
timeout 400 python bench.py > gpurun_out/r03_s28_bench.json 2> gpurun_out/r03_s28.err; echo "bench rc=$?" >> $O
python - >> $O <<'PY'
import json
for l in open('gpurun_out/r03_s28_bench.json'):
    if l.startswith('{'):
        d = json.loads(l); print(d['steps'], d['warmup'], round(d['value'], 1), 'f/s', round(d['ms_per_step'], 2), 'ms/step; single', round(d['single_lane']['value'], 1), 'fp32', round(d['fp32_mode']['value'],1), 'cpu', round(d['cpu_baseline']['value'],2), 'roofline frac', round(d['roofline']['frac'],3))
PY
cat $O
