#!/bin/bash
# round-3 GPU session 14: full suite at HEAD (generate_without_action added), smoke, bench line
set -u
O=gpurun_out/r03_s14; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider > $O/pytest_all.txt 2>&1
tail -5 $O/pytest_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r03_s14/bench_n1.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["stage_ms"])
for r in [d["roofline"]]+d["roofline_other"]:
    print(r["kernel"][:30], round(r["kernel_ms_per_step"],1), round(r["frac"],3), r.get("frac_rocprof"), r.get("traffic"))
print(d.get("fp32_mode",{}).get("value"), d.get("cpu_baseline",{}).get("value"))
PY
echo done > $O/done.txt
