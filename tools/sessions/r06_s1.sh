#!/bin/bash
# round 6, session 1: baseline of HEAD (round-5 closing state) on this round's box: the driver's command, then the lanes-only
# line with the lanes' LDS budget at 40 (default) / 64 / 96 KiB (where does the in-flight GEMM stand before any change)
set -u
R=$(pwd); O=$R/gpurun_out/r06_s1; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
for KB in 40 64 96; do
  timeout 300 python bench.py --only-lanes --steps 8 --warmup 2 --lane-lds-kb $KB --no-cpu-baseline --no-fp32-mode --no-other-configs > $O/lanes_kb$KB.json 2> $O/lanes_kb$KB.err
  python - <<PY
import json
d=json.loads(open("$O/lanes_kb$KB.json").read().strip().splitlines()[-1])
print("lds_kb $KB:", d["value"], d.get("ms_per_step"), json.dumps(d.get("roofline_in_flight",{}))[:400])
PY
done
echo done > $O/done.txt
