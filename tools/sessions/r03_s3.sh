#!/bin/bash
# round-3 GPU session 3: dgemm3 after the residual-order fix; CU-masked two-chain experiment; one-pass self-attention op test
set -u
O=gpurun_out/r03_s3; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -k "decode_gemm or skinny or one_pass" > $O/pytest_ops.txt 2>&1
tail -4 $O/pytest_ops.txt
P=tools/ubench/bin/dgemm_phase
( for lay in 0 1 2; do GEN=3 WARM=0 timeout 120 $P small 64 chains2 $lay; done
  GEN=2 timeout 120 $P small 64 chains2 0
  GEN=3 WARM=1 timeout 60 $P small 64 | head -5
) > $O/chains2.txt 2>&1
cat $O/chains2.txt
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu --tb=short -p no:cacheprovider -k "llama or rollout or generate or decode or fp32_decode" > $O/pytest_models.txt 2>&1
tail -5 $O/pytest_models.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-mode > $O/bench.txt 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r03_s3/bench.txt"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["stage_ms"], [ (r["kernel"][:22], round(r["kernel_ms_per_step"],1), round(r["frac"],3)) for r in [d["roofline"]]+d["roofline_other"]])
PY
echo done > $O/done.txt
