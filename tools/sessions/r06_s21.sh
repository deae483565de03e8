#!/bin/bash
# round 6, session 21: decode attention with 16 / 24 loads per lane in flight for small batches (bit-identical results)
set -u
R=$(pwd); O=$R/gpurun_out/r06_s21; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_edges.py tests/test_gpu_callers.py -q -x -p no:cacheprovider --tb=short -k "llama or rollout or decode_path or greedy or sampled or invariance or mbrl or vp2 or generate or continue" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
for arm in 1 0 1 0; do
echo "MBRL IVG_ATTN_DEEP=$arm: $(IVG_DEV=1 IVG_ATTN_DEEP=$arm timeout 300 python tools/mbrl_bench.py 16 12 2>&1 | tail -1)"
done
for C in 3 4; do
for arm in 1 0; do
  IVG_DEV=1 IVG_ATTN_DEEP=$arm timeout 600 python bench.py --config $C --lanes 1 --only-lanes --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-profile > $O/c${C}_a$arm.json 2> $O/c${C}_a$arm.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/c${C}_a$arm.json").read().strip().splitlines()[-1]); print("config $C one lane IVG_ATTN_DEEP=$arm:", round(d["value"],1), "f/s", round(d["ms_per_step"],1), "ms/step")
except Exception as e:
    print("config $C failed", e); print(open("$O/c${C}_a$arm.err").read()[-500:])
PY
done
done
echo done > $O/done.txt
