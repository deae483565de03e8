#!/bin/bash
# round-3 GPU session 19: two lanes, phase offsets, more steps
set -u
O=gpurun_out/r03_s19; mkdir -p $O
export TMPDIR=/tmp
for off in 0 70 110 150; do
  echo "== offset $off" >> $O/bench.txt; IVG_LANE_OFFSET_MS=$off timeout 600 python bench.py --lanes 2 --steps 16 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-profile >> $O/bench.txt 2>&1
done
python - <<'PY'
import json
off=0
for l in open("gpurun_out/r03_s19/bench.txt"):
    if l.startswith("=="): print(l.strip()); off=float(l.split()[-1])
    elif l.startswith("{"):
        d=json.loads(l); tot=d["ms_per_step"]*d["steps"]; print("   ", round(d["value"],1), "frames/s", round(d["ms_per_step"],1), "ms/step; without the start offset:", round((tot-off)/d["steps"],1), "ms/step =", round(64*14*d["steps"]/((tot-off)*1e-3),1), "frames/s")
PY
echo done > $O/done.txt
