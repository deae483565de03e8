#!/bin/bash
# key-permuted V^T tiles in the one-pass attention kernels (one ds_read_b128 per fragment): op / model tests, per-kernel time and LDS
# conflict counters, then the full GPU suite, smoke and the driver's bench command
set -u
R=$(pwd); O=$R/gpurun_out/r04_vt; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -q -x -p no:cacheprovider -k "one_pass_cross_attention or flash_prefill or batched_attention or self_attention" > $O/pytest_attn.txt 2>&1
tail -3 $O/pytest_attn.txt
grep -q " passed" $O/pytest_attn.txt && ! grep -q "failed" $O/pytest_attn.txt || { echo "ATTENTION TESTS FAILED"; tail -30 $O/pytest_attn.txt; exit 1; }
L1="python $R/bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt --output-format csv -- $L1 > $O/bench_under_trace.json 2> $O/trace.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 10 > $O/kernel_trace_summary.txt 2>&1
grep "xattn\|flash_prefill\|^kernel" $O/kernel_trace_summary.txt | cut -c1-150
rm -rf /tmp/prof_kt
PM="python $R/bench.py --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile"
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-include-regex 'xattn|flash_prefill' -d /tmp/prof_mfma -o p --output-format csv -- $PM > $O/pmc_mfma.log 2>&1
F=$(find /tmp/prof_mfma -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_mfma.json > $O/pmc_mfma.txt 2>&1)
cd $R
python tools/pmc_mfma_table.py $O/pmc_mfma.json > $O/pmc_mfma_table.txt 2>&1; cat $O/pmc_mfma_table.txt | cut -c1-130
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -6 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python -c "
import json
d=[json.loads(l) for l in open('$O/bench_n1.json') if l.startswith('{')][0]
print('value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), '| single', round(d['single_lane']['value'],1), '| fp32', round(d['fp32_mode']['value'],1), '| x3', round(d['compliant_mode']['value'],1), d['compliant_mode'].get('lanes_in_flight'))
print({k: round(v['value'],1) for k, v in d['other_configs'].items()}, 'stages', d['stage_ms']['encode_ms'], d['stage_ms']['rollout_ms'], d['stage_ms']['decode_ms'], 'tokenize_full', d['stage_ms'].get('tokenize_full_ms'))"
echo done > $O/done.txt
