#!/bin/bash
# round-5 closing GPU session: kernel traces (one batch / four batches in flight, lanes only), PMC traffic + MFMA / LDS counters, the
# driver's bench command, the RCCL path with lanes, the MBRL step path
set -u
R=$(pwd); O=$R/gpurun_out/r05_final; mkdir -p $O; export TMPDIR=/tmp
L1C="python bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs"
L1="python $R/bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt --output-format csv -- $L1 > $O/bench_under_trace_lanes1.json 2> $O/trace1.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && head -80 "$ST" > $O/bench_kernel_stats.csv
[ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 10 > $O/kernel_trace_summary.txt 2>&1
[ -n "$KT" ] && python $R/tools/trace_classes.py "$KT" 10 $O/kernel_trace_classes.json "$L1C" > $O/kernel_trace_classes.txt 2>&1
cat $O/kernel_trace_classes.txt
rm -rf /tmp/prof_kt
DEF="python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile --only-lanes"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_l4 -o l4 --output-format csv -- $DEF > $O/bench_under_trace_lanes4.json 2> $O/trace4.err
KT=$(find /tmp/prof_l4 -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_l4 -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && head -60 "$ST" > $O/lanes4_kernel_stats.csv
[ -n "$KT" ] && python $R/tools/sessions/overlap_report.py "$KT" > $O/lanes4_overlap.txt 2>&1
tail -20 $O/lanes4_overlap.txt
rm -rf /tmp/prof_l4
PMC_CMD="python bench.py --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile"
PM="python $R/bench.py --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex 'decode_attn|conv3x3|igemm_kernel|gemm256|dgemm|dg3_kernel' -d /tmp/prof_$C -o p --output-format csv -- $PM > $O/pmc_$C.log 2>&1
  F=$(find /tmp/prof_$C -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_$C.json > $O/pmc_$C.txt 2>&1)
  rm -rf /tmp/prof_$C
done
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-include-regex 'conv3x3|gemm256|igemm_kernel|xattn|flash_prefill' -d /tmp/prof_mfma -o p --output-format csv -- $PM > $O/pmc_mfma.log 2>&1
F=$(find /tmp/prof_mfma -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_mfma.json > $O/pmc_mfma.txt 2>&1)
cd $R
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/pmc_traffic.json "$PMC_CMD" > $O/pmc_traffic.txt 2>&1
python tools/pmc_mfma_table.py $O/pmc_mfma.json > $O/pmc_mfma_table.txt 2>&1
cat $O/pmc_traffic.txt; head -30 $O/pmc_mfma_table.txt | cut -c1-130
# the profiler-clock files have to be where bench.py looks for them
cp $O/kernel_trace_classes.json profiles/r05_kernel_trace_classes.json; cp $O/pmc_traffic.json profiles/r05_pmc_traffic.json
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
timeout 300 env IVG_FORCE_COLLECTIVE=1 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile > $O/bench_rccl_1rank.json 2> $O/bench_rccl.err
python -c "
import json
d=[json.loads(l) for l in open('$O/bench_rccl_1rank.json') if l.startswith('{')][0]; print('4 lanes with the per-step all-gather through RCCL (1 rank, gatherer thread):', round(d['value'],1), 'f/s')"
timeout 300 python tools/mbrl_bench.py 16 12 > $O/mbrl_rollout.txt 2>&1; tail -2 $O/mbrl_rollout.txt
grep -i "error\|Traceback" -A6 $O/bench_n1.err $O/bench_rccl.err | head -20
echo done > $O/done.txt
