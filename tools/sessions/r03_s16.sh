#!/bin/bash
# round-3 GPU session 16: fused GroupNorm apply only up to n N tiles (A/B)
set -u
O=gpurun_out/r03_s16; mkdir -p $O
export TMPDIR=/tmp
for e in "IVG_GN_APPLY_FUSE_MAXN=99" "IVG_GN_APPLY_FUSE_MAXN=1" "IVG_GN_APPLY_FUSE_MAXN=2" "IVG_GN_APPLY_FUSE_MAXN=99" "IVG_GN_APPLY_FUSE_MAXN=1" "IVG_GN_APPLY_FUSE_MAXN=2"; do
  echo "== $e" >> $O/bench.txt; env $e timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-profile >> $O/bench.txt 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r03_s16/bench.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"],1), round(d["ms_per_step"],2), {k:round(v,1) for k,v in d["stage_ms"].items() if k.endswith("_ms")})
PY
echo done > $O/done.txt
