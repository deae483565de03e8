#!/bin/bash
# round-2 GPU session 3: full GPU suite with the new boundary / eval / metrics / ingest tests, medium-model GEMM sweep
set -u
O=gpurun_out/r02_s3; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -40 $O/pytest_all.txt
echo "== default" >> $O/quick.txt; timeout 300 python tools/quick_bench.py --iters 3 >> $O/quick.txt 2>&1
echo "== IVG_CHAINS=2" >> $O/quick.txt; IVG_CHAINS=2 timeout 300 python tools/quick_bench.py --iters 3 >> $O/quick.txt 2>&1
grep -E "==|pred_frames" $O/quick.txt
timeout 900 python tools/dgemm_sweep.py 1024 4096 64 > $O/sweep_medium.txt 2>&1
tail -1 $O/sweep_medium.txt
IVG_CHAINS=2 timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_chains2.txt 2>&1
tail -5 $O/pytest_chains2.txt
echo done > $O/done.txt
