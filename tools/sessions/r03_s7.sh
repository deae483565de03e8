#!/bin/bash
# round-3 GPU session 7: bf16 deviation records; per-GPU shapes of configs 3, 4, 5
set -u
O=gpurun_out/r03_s7; mkdir -p $O
export TMPDIR=/tmp
rm -f gpurun_out/r03_bf16_deviations.jsonl
timeout 900 python -m pytest tests/test_gpu_bf16_deviation.py -q -m gpu --tb=short -p no:cacheprovider -s > $O/pytest_dev.txt 2>&1
tail -5 $O/pytest_dev.txt
for c in 3 4 5; do
  echo "== config $c" >> $O/configs.txt; timeout 600 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode >> $O/configs.txt 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r03_s7/configs.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], {k:round(v,1) for k,v in d["stage_ms"].items() if k.endswith("_ms")}, [ (r["kernel"][:22], round(r["kernel_ms_per_step"],1), round(r["frac"],3)) for r in [d["roofline"]]+d["roofline_other"]])
PY
echo done > $O/done.txt
