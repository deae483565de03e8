#!/bin/bash
# round 6, session 6: the committed rocprofv3 evidence at HEAD -- config 2 (kernel trace one lane / four lanes, FETCH / WRITE, MFMA / LDS)
# and, new, BASELINE config 4 (256 x 256, B = 16: kernel trace, FETCH / WRITE, MFMA / LDS)
set -u
R=$(pwd); O=$R/gpurun_out/r06_s6; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
prof_set () {   # $1 = tag, $2 = extra bench flags
  TAG=$1; FL=$2
  L1C="python bench.py $FL --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs"
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt --output-format csv -- python $R/bench.py $FL --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs > $O/${TAG}bench_under_trace.json 2> $O/${TAG}trace.err
  KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_kt -name "*kernel_stats.csv" | head -1)
  [ -n "$ST" ] && head -80 "$ST" > $O/${TAG}bench_kernel_stats.csv
  [ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 10 > $O/${TAG}kernel_trace_summary.txt 2>&1
  [ -n "$KT" ] && python $R/tools/trace_classes.py "$KT" 10 $O/${TAG}kernel_trace_classes.json "$L1C" > $O/${TAG}kernel_trace_classes.txt 2>&1
  cat $O/${TAG}kernel_trace_classes.txt
  rm -rf /tmp/prof_kt
  PMC_CMD="python bench.py $FL --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile"
  PM="python $R/bench.py $FL --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile"
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --kernel-include-regex 'decode_attn|conv3x3|igemm_kernel|gemm256|dgemm|dg3_kernel' -d /tmp/prof_$C -o p --output-format csv -- $PM > $O/${TAG}pmc_$C.log 2>&1
    F=$(find /tmp/prof_$C -name "*counter_collection.csv" | head -1)
    [ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/${TAG}pmc_$C.json > $O/${TAG}pmc_$C.txt 2>&1)
    rm -rf /tmp/prof_$C
  done
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-include-regex 'conv3x3|gemm256|igemm_kernel|xattn|flash_prefill' -d /tmp/prof_mfma -o p --output-format csv -- $PM > $O/${TAG}pmc_mfma.log 2>&1
  F=$(find /tmp/prof_mfma -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/${TAG}pmc_mfma.json > $O/${TAG}pmc_mfma.txt 2>&1)
  rm -rf /tmp/prof_mfma
  (cd $R && python tools/pmc_traffic.py $O/${TAG}pmc_FETCH_SIZE.json $O/${TAG}pmc_WRITE_SIZE.json $O/${TAG}pmc_traffic.json "$PMC_CMD" > $O/${TAG}pmc_traffic.txt 2>&1)
  (cd $R && python tools/pmc_mfma_table.py $O/${TAG}pmc_mfma.json > $O/${TAG}pmc_mfma_table.txt 2>&1)
  cat $O/${TAG}pmc_traffic.txt; head -24 $O/${TAG}pmc_mfma_table.txt | cut -c1-140
}
prof_set "" ""
prof_set "c4_" "--config 4"
DEF="python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile --only-lanes"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_l4 -o l4 --output-format csv -- $DEF > $O/bench_under_trace_lanes4.json 2> $O/trace4.err
KT=$(find /tmp/prof_l4 -name "*kernel_trace.csv" | head -1); ST=$(find /tmp/prof_l4 -name "*kernel_stats.csv" | head -1)
[ -n "$ST" ] && head -60 "$ST" > $O/lanes4_kernel_stats.csv
[ -n "$KT" ] && python $R/tools/sessions/overlap_report.py "$KT" > $O/lanes4_overlap.txt 2>&1
tail -25 $O/lanes4_overlap.txt
rm -rf /tmp/prof_l4
echo done > $O/done.txt
