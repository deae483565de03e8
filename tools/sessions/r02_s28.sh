#!/bin/bash
# round-2 GPU session 28: MFMA-busy / LDS counters of the conv3x3 instances with the end-of-round defaults (fused input GroupNorm)
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_s28; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
PM="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile"
timeout 110 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-include-regex 'conv3x3' -d /tmp/prof_mfma -o p --output-format csv -- $PM > $O/pmc_mfma.log 2>&1
F=$(find /tmp/prof_mfma -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_mfma.json > $O/pmc_mfma.txt 2>&1)
head -5 $O/pmc_mfma.txt | cut -c1-120
echo done > $O/done.txt
