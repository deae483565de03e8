#!/bin/bash
# round 5, session 2: conv3x3w with the next step's halo fragments read under the current step's MFMAs (PF) -- correctness, A/B per
# shape against PF=0 and the 256-pixel kernel, timing probes (no epilogue / no input normalisation), counters
set -u
R=$(pwd); O=$R/gpurun_out/r05_s2; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_wide.py -q -x -p no:cacheprovider --tb=short > $O/pytest_wide.txt 2>&1
tail -5 $O/pytest_wide.txt
IVG_CONV_WIDE_PF=0 timeout 900 python -m pytest tests/test_gpu_conv_wide.py -q -x -p no:cacheprovider --tb=short -k "against_fp64 or fused_input" > $O/pytest_wide_pf0.txt 2>&1
tail -2 $O/pytest_wide_pf0.txt
timeout 300 python tools/conv_ab.py 896 64 pf > $O/conv_ab_896_pf.txt 2>&1; cat $O/conv_ab_896_pf.txt
timeout 300 python tools/conv_ab.py 896 64 probe > $O/conv_ab_896_probe.txt 2>&1; cat $O/conv_ab_896_probe.txt
timeout 300 python tools/conv_ab.py 128 64 pf > $O/conv_ab_128_pf.txt 2>&1; cat $O/conv_ab_128_pf.txt
cd /tmp
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-include-regex 'conv3x3' -d /tmp/prof_mfma -o p --output-format csv -- python $R/tools/conv_ab.py 896 64 pf > $O/pmc_mfma.log 2>&1
F=$(find /tmp/prof_mfma -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_mfma.json > $O/pmc_mfma.txt 2>&1)
cd $R
python tools/pmc_mfma_table.py $O/pmc_mfma.json > $O/pmc_mfma_table.txt 2>&1; cut -c1-140 $O/pmc_mfma_table.txt
IVG_CONV_WIDE=1 timeout 300 python tools/quick_bench.py --iters 3 > $O/quick_wide.txt 2>&1; tail -1 $O/quick_wide.txt
echo done > $O/done.txt
