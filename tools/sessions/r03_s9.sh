#!/bin/bash
# round-3 GPU session 9: per-GEMM generation picks + L2 warm-up from second-generation launches; decode tests; configs 2 and 5
set -u
O=gpurun_out/r03_s9; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -k "decode_gemm or skinny" > $O/pytest_ops.txt 2>&1
tail -3 $O/pytest_ops.txt
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py -q -m gpu --tb=short -p no:cacheprovider -k "llama or rollout or generate or decode or fp32_decode or config" > $O/pytest_models.txt 2>&1
tail -3 $O/pytest_models.txt
P=tools/ubench/bin/dgemm_phase
( for s in small medium; do
    GEN=3 WARM=1 timeout 60 $P $s 64 | head -1
    GEN=3 WARM=0 timeout 60 $P $s 64 | head -1
    GEN=3 WARM=1 IVG_DG3_ALL=1 timeout 60 $P $s 64 | head -1
    GEN=2 WARM=1 timeout 60 $P $s 64 | head -1
    GEN=2 WARM=0 timeout 60 $P $s 64 | head -1
  done ) > $O/phase.txt 2>&1
cat $O/phase.txt
for c in 2 5; do
  echo "== config $c" >> $O/bench.txt; timeout 600 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode >> $O/bench.txt 2>&1
done
echo "== config 2 IVG_ATTN_PRE2=1" >> $O/bench.txt; IVG_ATTN_PRE2=1 timeout 600 python bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode >> $O/bench.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r03_s9/bench.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], {k:round(v,1) for k,v in d["stage_ms"].items() if k.endswith("_ms")}, [ (r["kernel"][:22], round(r["kernel_ms_per_step"],1), round(r["frac"],3)) for r in [d["roofline"]]+d["roofline_other"]])
PY
echo done > $O/done.txt
