#!/bin/bash
# round 5, session 3: conv3x3w -- second wave of a SIMD normalises after its MFMAs (LATE), workgroups started in phases (STAGGER)
set -u
R=$(pwd); O=$R/gpurun_out/r05_s3; mkdir -p $O; export TMPDIR=/tmp
IVG_CONV_WIDE_LATE=1 IVG_CONV_WIDE_STAGGER=2 timeout 900 python -m pytest tests/test_gpu_conv_wide.py -q -x -p no:cacheprovider --tb=short > $O/pytest_wide_late.txt 2>&1
tail -3 $O/pytest_wide_late.txt
timeout 300 python tools/conv_ab.py 896 64 stagger > $O/conv_ab_896_stagger.txt 2>&1; cat $O/conv_ab_896_stagger.txt
timeout 300 python tools/conv_ab.py 128 64 stagger > $O/conv_ab_128_stagger.txt 2>&1; cat $O/conv_ab_128_stagger.txt
timeout 300 python tools/conv_ab.py 224 256 stagger > $O/conv_ab_256res_stagger.txt 2>&1; cat $O/conv_ab_256res_stagger.txt
echo done > $O/done.txt
