#!/bin/bash
# round 5, session 19: op-level test of the fused decoder tail
set -u
R=$(pwd); O=$R/gpurun_out/r05_s19; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -q -p no:cacheprovider --tb=short -k "fused_decoder_tail or conv_out_planar" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
