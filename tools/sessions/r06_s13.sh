#!/bin/bash
# round 6, session 13: decode attention requests its first round of key rows before it reads the step counter -- tests, then one lane /
# four lanes against the previous commit's numbers of the same box (s12: 4,391 one lane)
set -u
R=$(pwd); O=$R/gpurun_out/r06_s13; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_shared.py tests/test_gpu_models.py tests/test_gpu_callers.py -q -x -p no:cacheprovider --tb=short > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
for i in 1 2; do
timeout 400 python bench.py --lanes 1 --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs > $O/l1_$i.json 2> $O/l1_$i.err
python - <<PY
import json
d=json.loads(open("$O/l1_$i.json").read().strip().splitlines()[-1])
a=d["roofline"]
print("lanes 1:", round(d["value"],1), "f/s | rollout", round(d["stage_ms"]["rollout_ms"],1), "| attn", round(a["avg_launch_ms"]*1e3,2), "us frac", round(a["frac"],3), a.get("fit"))
PY
done
timeout 400 python bench.py --only-lanes --steps 12 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs > $O/l4.json 2> $O/l4.err
python - <<PY
import json
d=json.loads(open("$O/l4.json").read().strip().splitlines()[-1]); r=d["roofline_in_flight"]
print("lanes 4:", round(d["value"],1), "f/s | phase", round(r["rollout_phase_ms"],1), "attn us", round(r["decode_attn_mean_launch_us_in_flight"],1), r["per_lane"][0]["decode_gemm_mean_launch_us_by_kind"])
PY
echo "config3: $(timeout 300 python bench.py --config 3 --lanes 1 --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['stage_ms']['rollout_ms'],1))")"
echo "mbrl: $(timeout 300 python tools/mbrl_bench.py 16 12 2>&1 | grep "reuse_cache=True")"
echo done > $O/done.txt
