#!/bin/bash
# round 6, session 10: tests of the last commit (shared attention op test, gemm256 row ranges); the latency-bound callers with replayed
# step graphs (IVG_GRAPH=1) -- MBRL step-wise rollout, BASELINE config 3 one lane, predict.py x5
set -u
R=$(pwd); O=$R/gpurun_out/r06_s10; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_shared.py -q -x -p no:cacheprovider --tb=short > $O/pytest_shared.txt 2>&1
tail -4 $O/pytest_shared.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider --tb=short -k "gemm256" > $O/pytest_gemm256.txt 2>&1
tail -3 $O/pytest_gemm256.txt
for G in 0 1; do
  echo "mbrl IVG_GRAPH=$G: $(IVG_GRAPH=$G timeout 300 python tools/mbrl_bench.py 16 12 2>&1 | grep reuse | tr '\n' '|')"
  echo "config 3 one lane IVG_GRAPH=$G: $(IVG_GRAPH=$G timeout 300 python bench.py --config 3 --lanes 1 --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['stage_ms'])")"
  echo "config 2 one lane IVG_GRAPH=$G: $(IVG_GRAPH=$G timeout 300 python bench.py --lanes 1 --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-profile --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['stage_ms'])")"
done
echo done > $O/done.txt
