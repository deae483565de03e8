#!/bin/bash
# round-2 GPU session 23: decode attention prologue, second form (requests before conversions, first key round unpredicated, cache
# append by unordered stores right after the RoPE) against the committed kernel; if it wins: full suite + bench line with it
set -u
O=gpurun_out/r02_s23; mkdir -p $O
export TMPDIR=/tmp
cp ivideogpt_amd/lib/libivg.so /tmp/libivg_new.so
timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu --tb=short -p no:cacheprovider -x -k "llama or generate or rollout or medium or bf16 or logits or sampled" > $O/pytest_models.txt 2>&1; tail -3 $O/pytest_models.txt
for v in new base new base; do
  case $v in new) cp /tmp/libivg_new.so ivideogpt_amd/lib/libivg.so;; base) cp ivideogpt_amd/lib/alt/libivg_base.so ivideogpt_amd/lib/libivg.so;; esac
  echo "== $v" >> $O/quick.txt; timeout 300 python tools/quick_bench.py --iters 5 >> $O/quick.txt 2>&1
done
cp /tmp/libivg_new.so ivideogpt_amd/lib/libivg.so
grep -E "==|pred_frames" $O/quick.txt | cut -c1-150
WIN=$(python - <<PY
import json,re
t={"new":[],"base":[]}; cur=None
for l in open("$O/quick.txt"):
    if l.startswith("== "): cur=l.split()[1]
    elif l.startswith("{") and cur: t[cur].append(json.loads(l)["generate_ms"])
a=sum(t["new"])/max(1,len(t["new"])); b=sum(t["base"])/max(1,len(t["base"]))
print("new" if (t["new"] and t["base"] and a < b - 0.5) else "base", round(a,2), round(b,2))
PY
)
echo "winner: $WIN" | tee $O/winner.txt
if [[ "$WIN" == new* ]] && grep -q "passed" $O/pytest_models.txt && ! grep -q "failed" $O/pytest_models.txt; then
  timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1; tail -4 $O/pytest_all.txt
  timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
  python - <<PY
import json
d=json.loads([l for l in open("$O/bench_n1.json") if l.startswith("{")][-1])
print("bench", round(d["value"],1), d["ms_per_step"], d["stage_ms"], d["roofline"]["frac"], [round(o["frac"],3) for o in d["roofline_other"]])
PY
fi
echo done > $O/done.txt
