#!/bin/bash
# round-3 GPU session 2: third-generation decode GEMM: correctness at every model shape, phase stamps, variants, first step numbers
set -u
O=gpurun_out/r03_s2; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -k "decode_gemm or skinny" > $O/pytest_dg.txt 2>&1
tail -8 $O/pytest_dg.txt
P=tools/ubench/bin/dgemm_phase
( GEN=2 timeout 60 $P small 64 | head -6
  GEN=3 WARM=0 timeout 60 $P small 64
  GEN=3 WARM=1 timeout 60 $P small 64 | head -6
  GEN=3 WARM=1 IVG_DG3_WARM_CAP=4 timeout 60 $P small 64 | head -6
  GEN=3 WARM=1 IVG_DG3_WARM_CAP=16 timeout 60 $P small 64 | head -6
  for f in "0,0,1,0" "0,0,0,8" "0,0,0,4" "2,2,0,0" "4,2,0,0"; do GEN=3 WARM=0 IVG_DG3_FORCE=$f timeout 60 $P small 64 | head -6; done
  GEN=2 timeout 60 $P medium 64 | head -6
  GEN=3 WARM=0 timeout 60 $P medium 64 | head -6
  GEN=3 WARM=1 timeout 60 $P medium 64 | head -6
) > $O/phase.txt 2>&1
cat $O/phase.txt
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu --tb=short -p no:cacheprovider -k "llama or rollout or generate or decode or fp32_decode" > $O/pytest_models.txt 2>&1
tail -5 $O/pytest_models.txt
for e in "IVG_DG3=0" "IVG_DG3_WARM=0" "X=1"; do
  echo "== $e" >> $O/bench.txt; env $e timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-mode >> $O/bench.txt 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r03_s2/bench.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["stage_ms"], [ (r["kernel"][:22], round(r["kernel_ms_per_step"],1), round(r["frac"],3)) for r in [d["roofline"]]+d["roofline_other"]])
PY
echo done > $O/done.txt
