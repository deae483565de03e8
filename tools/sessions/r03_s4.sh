#!/bin/bash
# round-3 GPU session 4: full GPU test suite at HEAD + A/B of the decode-attention value prefetch
set -u
O=gpurun_out/r03_s4; mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/r03_parity_margins.jsonl
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -x > $O/pytest_all.txt 2>&1
tail -12 $O/pytest_all.txt
for e in "IVG_ATTN_PRE2=0" "IVG_ATTN_PRE2=1"; do
  echo "== $e" >> $O/bench.txt; env $e timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-mode >> $O/bench.txt 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r03_s4/bench.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["stage_ms"], [ (r["kernel"][:22], round(r["kernel_ms_per_step"],1), round(r["frac"],3)) for r in [d["roofline"]]+d["roofline_other"]])
PY
echo done > $O/done.txt
