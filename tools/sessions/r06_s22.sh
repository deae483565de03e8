#!/bin/bash
# round 6, session 22: compliant (x3) mode with 2 / 3 / 4 batches in flight now that its K / V cache is 24-bit
set -u
R=$(pwd); O=$R/gpurun_out/r06_s22; mkdir -p $O; export TMPDIR=/tmp
for L in 2 3 4; do
  timeout 600 python bench.py --steps 8 --warmup 2 --x3-lanes $L --no-cpu-baseline --no-other-configs --no-profile > $O/x3_l$L.json 2> $O/x3_l$L.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/x3_l$L.json").read().strip().splitlines()[-1]); c=d["compliant_mode"]; print("x3 lanes $L:", round(c["value"],1), "one lane;", round(c["lanes_in_flight"]["value"],1), "with", c["lanes_in_flight"]["lanes"], "in flight; fp32", round(d["fp32_mode"]["value"],1), "| headline", round(d["value"],1), "single", round(d["single_lane"]["value"],1))
except Exception as e:
    print("failed", e); print(open("$O/x3_l$L.err").read()[-800:])
PY
done
echo done > $O/done.txt
