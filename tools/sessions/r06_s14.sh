#!/bin/bash
# round 6, session 14: BASELINE config 4 (256x256, B = 16: light rollouts, heavy convolutions) and config 3 with more batches in flight
set -u
R=$(pwd); O=$R/gpurun_out/r06_s14; mkdir -p $O; export TMPDIR=/tmp
for C in 4 3; do
for L in 4 6 8; do
  timeout 600 python bench.py --config $C --lanes $L --only-lanes --steps $((3*L)) --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile > $O/c${C}_l$L.json 2> $O/c${C}_l$L.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/c${C}_l$L.json").read().strip().splitlines()[-1]); print("config $C lanes $L:", round(d["value"],1), "f/s", round(d["ms_per_step"],1), "ms/step")
except Exception as e:
    print("config $C lanes $L failed", e); print(open("$O/c${C}_l$L.err").read()[-500:])
PY
done
done
echo done > $O/done.txt
