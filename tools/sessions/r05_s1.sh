#!/bin/bash
# round 5, session 1: the persistent two-tile conv3x3 (conv3x3w.hip) -- correctness, per-shape A/B against the 256-pixel kernel,
# exact per-kernel durations (kernel trace), MFMA-busy / LDS counters, and the decode stage with the switch on / off
set -u
R=$(pwd); O=$R/gpurun_out/r05_s1; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_wide.py -q -x -p no:cacheprovider --tb=short > $O/pytest_wide.txt 2>&1
tail -15 $O/pytest_wide.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider -k "conv" --tb=short > $O/pytest_conv_ops.txt 2>&1
tail -3 $O/pytest_conv_ops.txt
timeout 300 python tools/conv_ab.py 896 64 > $O/conv_ab_896.txt 2>&1; cat $O/conv_ab_896.txt
timeout 300 python tools/conv_ab.py 128 64 > $O/conv_ab_128.txt 2>&1; cat $O/conv_ab_128.txt
timeout 400 python tools/conv_ab.py 224 256 > $O/conv_ab_256res.txt 2>&1; cat $O/conv_ab_256res.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_kt -o kt --output-format csv -- python $R/tools/conv_ab.py 896 64 > $O/trace_run.txt 2> $O/trace.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 1 > $O/kernel_trace_summary.txt 2>&1
grep "conv3x3\|^kernel" $O/kernel_trace_summary.txt | cut -c1-170
rm -rf /tmp/prof_kt
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-include-regex 'conv3x3' -d /tmp/prof_mfma -o p --output-format csv -- python $R/tools/conv_ab.py 896 64 > $O/pmc_mfma.log 2>&1
F=$(find /tmp/prof_mfma -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_mfma.json > $O/pmc_mfma.txt 2>&1)
cd $R
python tools/pmc_mfma_table.py $O/pmc_mfma.json > $O/pmc_mfma_table.txt 2>&1; cut -c1-140 $O/pmc_mfma_table.txt
IVG_CONV_WIDE=0 timeout 300 python tools/quick_bench.py --iters 3 > $O/quick_narrow.txt 2>&1; tail -1 $O/quick_narrow.txt
IVG_CONV_WIDE=1 timeout 300 python tools/quick_bench.py --iters 3 > $O/quick_wide.txt 2>&1; tail -1 $O/quick_wide.txt
echo done > $O/done.txt
