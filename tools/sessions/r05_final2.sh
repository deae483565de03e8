#!/bin/bash
# round-5 closing session at HEAD: one-batch kernel trace (class means for bench.py), full GPU suite, smoke, the driver's bench command
set -u
R=$(pwd); O=$R/gpurun_out/r05_final2; mkdir -p $O; export TMPDIR=/tmp
L1C="python bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt --output-format csv -- python $R/bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs > $O/bench_under_trace_lanes1.json 2> $O/trace1.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && head -80 "$ST" > $O/bench_kernel_stats.csv
[ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 10 > $O/kernel_trace_summary.txt 2>&1
[ -n "$KT" ] && python $R/tools/trace_classes.py "$KT" 10 $O/kernel_trace_classes.json "$L1C" > $O/kernel_trace_classes.txt 2>&1
python -c "
import json; d=json.load(open('$O/kernel_trace_classes.json'))['classes']; print({k:(round(v['launches_per_step'],1), round(v['mean_us'],2), round(v['ms_per_step'],2)) for k,v in d.items()})"
rm -rf /tmp/prof_kt
cd $R
cp $O/kernel_trace_classes.json profiles/r05_kernel_trace_classes.json
rm -f $R/gpurun_out/r03_parity_margins.jsonl $R/gpurun_out/r03_bf16_deviations.jsonl
timeout 1800 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -6 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python - <<PY
import json
d=[json.loads(l) for l in open('$O/bench_n1.json') if l.startswith('{')][0]
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), '| single', round(d['single_lane']['value'],1), '| fp32', round(d['fp32_mode']['value'],1), '| x3', round(d['compliant_mode']['value'],1), d['compliant_mode'].get('lanes_in_flight'))
for r in [d['roofline']] + d['roofline_other']: print(r['kernel'][:40], 'frac', round(r['frac'],3), 'stamps', round(r.get('frac_stamps',0),3), 'sustained', round(r.get('frac_of_sustained',0),3), 'ms/step', round(r['kernel_ms_per_step'],1), 'traffic', r.get('traffic'))
r=d['roofline_in_flight']; print('in flight', round(r['achieved'],0), 'GB/s', round(r['frac'],3), 'phase ms', round(r['rollout_phase_ms'],1))
for k,v in d['other_configs'].items(): print(k, round(v['value'],1), v.get('lanes_in_flight'))
print('stages', d['stage_ms'], 'cpu', d['cpu_baseline']['value'])
PY
echo done > $O/done.txt
