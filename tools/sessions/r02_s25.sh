#!/bin/bash
# round-2 GPU session 25: GroupNorm + SiLU applied inside the conv3x3 halo staging (IVG_GN_APPLY_FUSE=1) with the rebuilt loop
set -u
O=gpurun_out/r02_s25; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -x -k "fused_input_groupnorm or conv3x3" > $O/pytest_ops.txt 2>&1; tail -3 $O/pytest_ops.txt
for e in "IVG_GN_APPLY_FUSE=1" "IVG_GN_APPLY_FUSE=0" "IVG_GN_APPLY_FUSE=1" "IVG_GN_APPLY_FUSE=0"; do
  echo "== $e" >> $O/quick.txt; env $e timeout 300 python tools/quick_bench.py --iters 5 >> $O/quick.txt 2>&1
done
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
WIN=$(python - <<PY
import json
t={"1":[],"0":[]}; cur=None
for l in open("$O/quick.txt"):
    if l.startswith("== "): cur=l.strip()[-1]
    elif l.startswith("{") and cur: t[cur].append(json.loads(l)["total_ms"])
a=sum(t["1"])/max(1,len(t["1"])); b=sum(t["0"])/max(1,len(t["0"]))
print("fused" if (t["1"] and t["0"] and a < b - 1.0) else "separate", round(a,2), round(b,2))
PY
)
echo "winner: $WIN" | tee $O/winner.txt
if [[ "$WIN" == fused* ]] && ! grep -q "failed" $O/pytest_ops.txt; then
  export IVG_GN_APPLY_FUSE=1
  timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_callers.py tests/test_gpu_fullsize.py -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_rest.txt 2>&1; tail -4 $O/pytest_rest.txt
  timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
  python - <<PY
import json
d=json.loads([l for l in open("$O/bench_n1.json") if l.startswith("{")][-1])
print("bench", round(d["value"],1), d["ms_per_step"], d["stage_ms"], d["roofline"]["frac"], [round(o["frac"],3) for o in d["roofline_other"]])
PY
fi
echo done > $O/done.txt
