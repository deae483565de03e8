#!/bin/bash
# round 5, session 6: conv3x3w with its coverage policy (default switches) -- tests, A/B at the three decoder batch shapes, socket
# power / shader clock under the convolution kernels, decode stage
set -u
R=$(pwd); O=$R/gpurun_out/r05_s6; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_wide.py -q -x -p no:cacheprovider --tb=short > $O/pytest_wide.txt 2>&1
tail -3 $O/pytest_wide.txt
timeout 300 python tools/conv_ab.py 896 64 policy > $O/conv_ab_896_policy.txt 2>&1; cat $O/conv_ab_896_policy.txt
timeout 300 python tools/conv_ab.py 128 64 policy > $O/conv_ab_128_policy.txt 2>&1; cat $O/conv_ab_128_policy.txt
timeout 300 python tools/conv_ab.py 224 256 policy > $O/conv_ab_256res_policy.txt 2>&1; cat $O/conv_ab_256res_policy.txt
timeout 300 python tools/conv_power.py 2.5 > $O/conv_power.txt 2>&1; cat $O/conv_power.txt
IVG_CONV_WIDE=0 timeout 300 python tools/quick_bench.py --iters 5 > $O/quick_narrow.txt 2>&1; tail -1 $O/quick_narrow.txt
IVG_CONV_WIDE=1 timeout 300 python tools/quick_bench.py --iters 5 > $O/quick_wide.txt 2>&1; tail -1 $O/quick_wide.txt
IVG_CONV_WIDE=0 timeout 300 python tools/quick_bench.py --iters 3 --res 256 --batch 16 > $O/quick256_narrow.txt 2>&1; tail -1 $O/quick256_narrow.txt
IVG_CONV_WIDE=1 timeout 300 python tools/quick_bench.py --iters 3 --res 256 --batch 16 > $O/quick256_wide.txt 2>&1; tail -1 $O/quick256_wide.txt
echo done > $O/done.txt
