#!/bin/bash
# x3 (compliant) mode with three / four batches in flight and the default decode-GEMM footprint (r04_lanes.txt has 2 lanes: 2,301; 3 lanes with the 40 KiB footprint: 2,229)
set -u
R=$(pwd); O=$R/gpurun_out/r04_x3_lanes; mkdir -p $O
C="--decode-dtype x3 --llm-dtype x3 --lane-switches none --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-profile --no-other-configs"
for l in 3 4; do
  echo "== x3 mode, $l lanes, no lane switches" >> $O/x3.txt
  timeout 400 python bench.py $C --lanes $l 2>$O/err_$l.txt | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(round(d['value'],1),'f/s',round(d['ms_per_step'],2),'ms/step | single',round(d['single_lane']['value'],1))" >> $O/x3.txt
done
cat $O/x3.txt; tail -3 $O/err_4.txt
