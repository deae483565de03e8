#!/bin/bash
# round-2 GPU session 24: where do the waves of the rebuilt conv3x3 loop spend their cycles (PMC split), two shapes
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_s24; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for shape in "64 128 128 0" "16 512 512 0" "32 256 256 1"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-include-regex 'conv3x3' -d /tmp/prof_c$i -o p --output-format csv -- python $R/tools/conv_bench.py $shape > $O/pmc_c$i.log 2>&1
  F=$(find /tmp/prof_c$i -name "*counter_collection.csv" | head -1)
  echo "== shape $shape" >> $O/pmc_split.txt
  [ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_c$i.json >> $O/pmc_split.txt 2>&1)
done
cat $O/pmc_split.txt | cut -c1-150
echo done > $O/done.txt
