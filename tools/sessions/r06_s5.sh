#!/bin/bash
# round 6, session 5: the driver's command with the sub-pixel upsamplers and the shared_context entries; caller tests
set -u
R=$(pwd); O=$R/gpurun_out/r06_s5; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_callers.py tests/test_gpu_evaluate.py tests/test_gpu_shared.py -q -x -p no:cacheprovider --tb=short > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "single", d["single_lane"]["value"], "stage", {k:round(v,1) for k,v in d["stage_ms"].items() if k.endswith("_ms")})
print("compliant", d["compliant_mode"]["value"], d["compliant_mode"].get("lanes_in_flight",{}).get("value"), "fp32", d["fp32_mode"]["value"])
for k,v in d["other_configs"].items(): print(k, v.get("value"), v.get("lanes_in_flight",{}).get("value"))
print(json.dumps(d.get("shared_context"), indent=1)[:3000])
PY
echo done > $O/done.txt
