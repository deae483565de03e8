#!/bin/bash
# round-2 GPU session 6: graph-steps sweep, the other BASELINE configs, the MBRL step path, the full bench line
set -u
O=gpurun_out/r02_s6; mkdir -p $O
export TMPDIR=/tmp
for n in 1 8 16 32 64; do
  echo "== IVG_GRAPH_STEPS=$n" >> $O/quick.txt; IVG_GRAPH_STEPS=$n timeout 300 python tools/quick_bench.py --iters 3 >> $O/quick.txt 2>&1
done
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
for c in 3 4 5; do
  timeout 600 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-mode > $O/bench_config$c.json 2> $O/bench_config$c.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/bench_config$c.json") if l.startswith("{")][-1])
    print("config $c", round(d["value"],1), "frames/s", round(d["ms_per_step"],1), "ms/step", d["stage_ms"], d["config"]["workload"][:80])
except Exception as e: print("config $c ERR", e)
PY
done
timeout 600 python tools/mbrl_bench.py 16 12 > $O/mbrl.txt 2>&1; tail -4 $O/mbrl.txt
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
tail -c 1500 $O/bench_n1.json
echo done > $O/done.txt
