#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_s11; mkdir -p $O
timeout 300 python tools/conv_rollout_pair.py 2.5 > $O/pair.txt 2> $O/pair.err; cat $O/pair.txt; tail -3 $O/pair.err
timeout 200 python tools/mbrl_bench.py 16 12 > $O/mbrl.txt 2>&1; tail -3 $O/mbrl.txt
