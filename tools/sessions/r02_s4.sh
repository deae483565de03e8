#!/bin/bash
# round-2 GPU session 4: all-DMA pipelined decode GEMM -- parity, sweep, bench line with the new bench.py
set -u
O=gpurun_out/r02_s4; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -k "decode_gemm or skinny or ingest or metrics" > $O/pytest_ops.txt 2>&1
tail -5 $O/pytest_ops.txt
echo "== default" >> $O/quick.txt; timeout 300 python tools/quick_bench.py --iters 3 >> $O/quick.txt 2>&1
grep -E "==|pred_frames" $O/quick.txt | cut -c1-200
SWEEP_QUICK=0 timeout 900 python tools/dgemm_sweep.py > $O/sweep_small.txt 2>&1
tail -1 $O/sweep_small.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json
timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider --deselect tests/test_gpu_ops.py > $O/pytest_rest.txt 2>&1
tail -15 $O/pytest_rest.txt
echo done > $O/done.txt
