#!/bin/bash
# round 5, session 4: what the bf16 matrix pipe sustains (zero / random operands, with LDS fragment traffic) -- the ceiling the
# convolution kernels are priced against
set -u
R=$(pwd); O=$R/gpurun_out/r05_s4; mkdir -p $O
timeout 120 tools/ubench/bin/mfma_power 20 > $O/mfma_power_20ms.txt 2>&1; cat $O/mfma_power_20ms.txt
timeout 200 tools/ubench/bin/mfma_power 300 > $O/mfma_power_300ms.txt 2>&1; cat $O/mfma_power_300ms.txt
(rocm-smi --showpower --showclocks --showmaxpower 2>&1 | head -40) > $O/rocm_smi.txt; cat $O/rocm_smi.txt
echo done > $O/done.txt
