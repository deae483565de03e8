#!/bin/bash
# round-3 GPU session 18: several batches in flight on one GPU (bench.py --lanes): 1, 2, 3 lanes
set -u
O=gpurun_out/r03_s18; mkdir -p $O
export TMPDIR=/tmp
for l in 1 2 3 2; do
  echo "== lanes $l" >> $O/bench.txt; timeout 600 python bench.py --lanes $l --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-profile >> $O/bench.txt 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r03_s18/bench.txt"):
    if l.startswith("=="): print(l.strip())
    elif l.startswith("{"):
        d=json.loads(l); print("   ", round(d["value"],1), "frames/s", round(d["ms_per_step"],1), "ms/step", d.get("single_lane"))
    elif "Error" in l or "error" in l: print(l.strip()[:200])
PY
tail -5 $O/bench.txt | cut -c1-300
echo done > $O/done.txt
