#!/bin/bash
# round-3 GPU session 6: blocked tile order of the prompt-pass GEMM (A/B), kernel trace of the step
set -u
O=gpurun_out/r03_s6; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -k "gemm256" > $O/pytest_ops.txt 2>&1
tail -3 $O/pytest_ops.txt
timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu --tb=short -p no:cacheprovider -k "llama_small or flash or full_width_llama" > $O/pytest_models.txt 2>&1
tail -3 $O/pytest_models.txt
for e in "IVG_G256_GROUP=0" "IVG_G256_GROUP=1" "IVG_G256_GROUP=0" "IVG_G256_GROUP=1"; do
  echo "== $e" >> $O/bench.txt; env $e timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-mode >> $O/bench.txt 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r03_s6/bench.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], {k:round(v,1) for k,v in d["stage_ms"].items() if k.endswith("_ms")}, [ (r["kernel"][:22], round(r["kernel_ms_per_step"],1), round(r["frac"],3)) for r in [d["roofline"]]+d["roofline_other"]])
PY
echo done > $O/done.txt
