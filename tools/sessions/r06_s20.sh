#!/bin/bash
# round 6, session 20: where does a step-wise MBRL rollout (B = 16, horizon 12) spend its time?
set -u
R=$(pwd); O=$R/gpurun_out/r06_s20; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python tools/mbrl_bench.py 16 12 > $O/mbrl.txt 2>&1; tail -2 $O/mbrl.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --hip-trace -d /tmp/prof_kt -o kt --output-format csv -- python $R/tools/mbrl_bench.py 16 12 > $O/q.txt 2> $O/trace.err
KT=$(find /tmp/prof_kt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python $R/tools/trace_summary.py "$KT" 10 > $O/trace_mbrl.txt 2>&1
head -40 $O/trace_mbrl.txt | cut -c1-150
HT=$(find /tmp/prof_kt -name "*hip_api_trace.csv" | head -1)
[ -n "$HT" ] && python - "$HT" > $O/hip_api.txt 2>&1 <<'PY'
import csv, sys, collections
d = collections.defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Function"]; d[k][0] += 1; d[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, t) in sorted(d.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:45s} {n:9d} calls {t / 1e6:10.1f} ms  {t / max(n, 1) / 1e3:8.2f} us/call")
PY
head -25 $O/hip_api.txt
echo done > $O/done.txt
