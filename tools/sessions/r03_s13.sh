#!/bin/bash
# round-3 GPU session 13: staggered in-place GroupNorm transform of the conv3x3 (A/B + correctness)
set -u
O=gpurun_out/r03_s13; mkdir -p $O
export TMPDIR=/tmp
IVG_C3_STAGGER=1 timeout 600 python -m pytest tests/test_gpu_models.py tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -k "groupnorm or bf16 or detokenize or gn_conv or conv_gn" > $O/pytest_stg.txt 2>&1
tail -3 $O/pytest_stg.txt
for e in "IVG_C3_STAGGER=0" "IVG_C3_STAGGER=1" "IVG_C3_STAGGER=0" "IVG_C3_STAGGER=1"; do
  echo "== $e" >> $O/bench.txt; env $e timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-mode >> $O/bench.txt 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r03_s13/bench.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"],1), round(d["ms_per_step"],2), {k:round(v,1) for k,v in d["stage_ms"].items() if k.endswith("_ms")}, [ (r["kernel"][:22], round(r["kernel_ms_per_step"],1), round(r["frac"],3)) for r in [d["roofline"]]+d["roofline_other"] if "conv3x3" in r["kernel"]])
PY
echo done > $O/done.txt
