#!/bin/bash
# round-3 GPU session 8: medium-transformer decode GEMM generations per GEMM (harness), config 5 A/B
set -u
O=gpurun_out/r03_s8; mkdir -p $O
export TMPDIR=/tmp
P=tools/ubench/bin/dgemm_phase
( for m in 2222 3333 3323 3223 3322 2323; do GEN=3 WARM=1 GENMASK=$m timeout 60 $P medium 64 | head -1; done
  for f in "4,2,0,8" "2,2,0,8" "4,2,1,8" "1,2,0,16"; do GEN=3 WARM=1 IVG_DG3_FORCE=$f timeout 60 $P medium 64 | head -5; done
  for m in 2222 3333 3323 3233; do GEN=3 WARM=1 GENMASK=$m timeout 60 $P small 64 | head -1; done
) > $O/medium.txt 2>&1
cat $O/medium.txt
for e in "IVG_DG3=0" "IVG_DG3=1"; do
  echo "== $e" >> $O/config5.txt; env $e timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile >> $O/config5.txt 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r03_s8/config5.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], {k:round(v,1) for k,v in d["stage_ms"].items() if k.endswith("_ms")})
PY
echo done > $O/done.txt
