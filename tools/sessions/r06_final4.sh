#!/bin/bash
# round 6, closing session 4 at HEAD (16-channel tail, 24-bit K / V cache + 256-tile GEMM of the x3 mode): config-2 rocprofv3 evidence again (kernel trace classes,
# FETCH / WRITE, MFMA / LDS), full GPU suite, smoke, the driver's bench command, 1-rank RCCL line, MBRL
set -u
R=$(pwd); O=$R/gpurun_out/r06_final4; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
L1C="python bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt --output-format csv -- python $R/bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs > $O/bench_under_trace.json 2> $O/trace.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && head -80 "$ST" > $O/bench_kernel_stats.csv
[ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 10 > $O/kernel_trace_summary.txt 2>&1
[ -n "$KT" ] && python $R/tools/trace_classes.py "$KT" 10 $O/kernel_trace_classes.json "$L1C" > $O/kernel_trace_classes.txt 2>&1
python -c "
import json; d=json.load(open('$O/kernel_trace_classes.json'))['classes']; print({k:(round(v['launches_per_step'],1), round(v['mean_us'],2), round(v['ms_per_step'],2)) for k,v in d.items()})"
rm -rf /tmp/prof_kt
PMC_CMD="python bench.py --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile"
PM="python $R/bench.py --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex 'decode_attn|conv3x3|igemm_kernel|gemm256|dgemm|dg3_kernel' -d /tmp/prof_$C -o p --output-format csv -- $PM > $O/pmc_$C.log 2>&1
  F=$(find /tmp/prof_$C -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_$C.json > $O/pmc_$C.txt 2>&1)
  rm -rf /tmp/prof_$C
done
(cd $R && python tools/pmc_traffic.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/pmc_traffic.json "$PMC_CMD" > $O/pmc_traffic.txt 2>&1)
cat $O/pmc_traffic.txt
cd $R
cp $O/kernel_trace_classes.json profiles/r06_kernel_trace_classes.json; cp $O/pmc_traffic.json profiles/r06_pmc_traffic.json
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
echo done > $O/done.txt
