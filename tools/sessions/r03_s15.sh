#!/bin/bash
# round-3 GPU session 15: partial W tiles (wr rows per workgroup) of the third-generation decode GEMM: tests, harness, bench A/B
set -u
O=gpurun_out/r03_s15; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -k "decode_gemm or skinny" > $O/pytest_ops.txt 2>&1
tail -3 $O/pytest_ops.txt
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py -q -m gpu --tb=short -p no:cacheprovider -k "llama or rollout or generate or decode or fp32_decode or config" > $O/pytest_models.txt 2>&1
tail -3 $O/pytest_models.txt
P=tools/ubench/bin/dgemm_phase
( for s in small medium; do
    for w in 0 1 0 1; do IVG_DG3_WR=$w GEN=3 WARM=1 timeout 60 $P $s 64 | head -5; done
  done ) > $O/phase.txt 2>&1
grep "layer chain" $O/phase.txt
for e in "IVG_DG3_WR=0" "IVG_DG3_WR=1" "IVG_DG3_WR=0" "IVG_DG3_WR=1"; do
  echo "== $e" >> $O/bench.txt; env $e timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-mode >> $O/bench.txt 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r03_s15/bench.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print(round(d["value"],1), round(d["ms_per_step"],2), {k:round(v,1) for k,v in d["stage_ms"].items() if k.endswith("_ms")}, [ (r["kernel"][:22], round(r["kernel_ms_per_step"],1), round(r["frac"],3)) for r in [d["roofline"]]+d["roofline_other"] if "dgemm" in r["kernel"]])
PY
echo done > $O/done.txt
