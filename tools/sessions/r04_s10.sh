#!/bin/bash
# round 4, session 10: kernel trace of the capped-convolution mode (per-launch durations of the decode attention beside a capped conv grid)
R=$(pwd); O=$R/gpurun_out/r04_s10; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
B="python $R/bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile"
timeout 600 env IVG_CONV_CAP=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_cap -o cap --output-format csv -- $B --conv-gate 1 > $O/bench_cap.json 2> $O/cap.err
KT=$(find /tmp/prof_cap -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python $R/tools/sessions/overlap_report.py "$KT" > $O/cap_overlap.txt 2>&1
cat $O/cap_overlap.txt | cut -c1-130
python -c "
import json
d=[json.loads(l) for l in open('$O/bench_cap.json') if l.startswith('{')][0]; print('under trace: capped conv + gate, 4 lanes:', round(d['value'],1), 'f/s')"
