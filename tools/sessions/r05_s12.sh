#!/bin/bash
# round 5, session 12: host enqueue time of a rollout vs device time, 1 / 2 / 4 threads, at B = 64 and at B = 2 (tiny device work)
set -u
R=$(pwd); O=$R/gpurun_out/r05_s12; mkdir -p $O
timeout 300 python tools/host_enqueue.py 64 > $O/host_enqueue_b64.txt 2>&1; tail -3 $O/host_enqueue_b64.txt
timeout 300 python tools/host_enqueue.py 2 > $O/host_enqueue_b2.txt 2>&1; tail -3 $O/host_enqueue_b2.txt
echo done > $O/done.txt
