#!/bin/bash
# round 6, session 30: split-bf16 arithmetic in the second-generation decode GEMMs of x3 engines (lm_head; every GEMM with batches in flight)
set -u
R=$(pwd); O=$R/gpurun_out/r06_s30; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider --tb=short -k "decode_gemm_model_shapes or skinny" > $O/pytest_ops.txt 2>&1
tail -4 $O/pytest_ops.txt
timeout 1500 python -m pytest tests/test_gpu_x3.py -q -x -p no:cacheprovider --tb=short > $O/pytest_x3.txt 2>&1
tail -4 $O/pytest_x3.txt
for arm in 1 0 1 0; do
echo "x3 one lane, IVG_DG2_X3=$arm: $(IVG_DEV=1 IVG_DG2_X3=$arm timeout 300 python tools/quick_bench.py --dec x3 --llm x3 --iters 3 2>&1 | tail -1 | cut -c1-150)"
done
for arm in 1 0; do
  IVG_DEV=1 IVG_DG2_X3=$arm timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --no-profile > $O/x3_a$arm.json 2> $O/x3_a$arm.err
  python - <<PY
import json
d=json.loads(open("$O/x3_a$arm.json").read().strip().splitlines()[-1]); c=d["compliant_mode"]; print("IVG_DG2_X3=$arm: x3", round(c["value"],1), "one lane;", round(c["lanes_in_flight"]["value"],1), "with", c["lanes_in_flight"]["lanes"], "in flight | headline", round(d["value"],1))
PY
done
echo done > $O/done.txt
