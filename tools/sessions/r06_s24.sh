#!/bin/bash
# round 6, session 24: the prompt pass's RMSNorms folded into the q/k/v and gate/up GEMMs (gemm256l ROWNORM)
set -u
R=$(pwd); O=$R/gpurun_out/r06_s24; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider --tb=short -k "gemm256 or rms" > $O/pytest_ops.txt 2>&1
tail -4 $O/pytest_ops.txt
timeout 2400 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_bf16_deviation.py tests/test_gpu_shared.py tests/test_gpu_evaluate.py -q -x -p no:cacheprovider --tb=short > $O/pytest_models.txt 2>&1
tail -4 $O/pytest_models.txt
for arm in 1 0 1 0; do
echo "IVG_PROMPT_ROWNORM=$arm: $(IVG_DEV=1 IVG_PROMPT_ROWNORM=$arm timeout 300 python tools/quick_bench.py --iters 5 2>&1 | tail -1)"
done
for arm in 1 0 1 0; do
  IVG_DEV=1 IVG_PROMPT_ROWNORM=$arm timeout 600 python bench.py --only-lanes --steps 16 --warmup 4 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile > $O/l4_a$arm.json 2> $O/l4_a$arm.err
  python - <<PY
import json
d=json.loads(open("$O/l4_a$arm.json").read().strip().splitlines()[-1]); print("four lanes IVG_PROMPT_ROWNORM=$arm:", round(d["value"],1), "f/s", round(d["ms_per_step"],2), "ms/step")
PY
done
echo done > $O/done.txt
