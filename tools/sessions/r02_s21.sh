#!/bin/bash
# round-2 GPU session 21: full GPU suite, smoke, the bench line, the profiles of the bench command (kernel trace / stats, PMC traffic,
# MFMA-busy of the MFMA kernels), the other BASELINE configs and the MBRL step path -- with the defaults of the end of the round
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_s21; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -6 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
tail -c 2500 $O/bench_n1.json
BENCH="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt --output-format csv -- $BENCH > $O/bench_under_trace.json 2> $O/trace.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && head -80 "$ST" > $O/bench_kernel_stats.csv
[ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 5 > $O/kernel_trace_summary.txt 2>&1
PM="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex 'decode_attn|conv3x3|igemm_kernel|gemm256|dgemm' -d /tmp/prof_$C -o p --output-format csv -- $PM > $O/pmc_$C.log 2>&1
  F=$(find /tmp/prof_$C -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_$C.json > $O/pmc_$C.txt 2>&1)
done
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-include-regex 'conv3x3|gemm256|igemm_kernel|xattn|flash_prefill' -d /tmp/prof_mfma -o p --output-format csv -- $PM > $O/pmc_mfma.log 2>&1
F=$(find /tmp/prof_mfma -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_mfma.json > $O/pmc_mfma.txt 2>&1)
cd $R
python tools/pmc_traffic.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/pmc_traffic.json "python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile" > $O/pmc_traffic.txt 2>&1
cat $O/pmc_traffic.txt; tail -3 $O/kernel_trace_summary.txt
for c in 3 4 5; do
  timeout 600 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode > $O/bench_config$c.json 2> $O/bench_config$c.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_config$c.json") if l.startswith("{")][-1])
    print("config $c", round(d["value"],1), "frames/s", round(d["ms_per_step"],1), "ms/step", d.get("stage_ms"), d["config"]["workload"][:80])
except Exception as e: print("config $c ERR", e)
PY
done
timeout 600 python tools/mbrl_bench.py 16 12 > $O/mbrl.txt 2>&1; tail -4 $O/mbrl.txt
echo done > $O/done.txt
