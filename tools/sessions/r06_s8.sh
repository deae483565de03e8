#!/bin/bash
# round 6, session 8: per-kind LDS budgets of the in-flight decode GEMMs -- which GEMMs gain from keeping their whole K range in flight
# (one memory round trip instead of three to six) although their workgroups get bigger
set -u
R=$(pwd); O=$R/gpurun_out/r06_s8; mkdir -p $O; export TMPDIR=/tmp
run () {
  TAG=$1; shift
  env "$@" timeout 400 python bench.py --only-lanes --steps 12 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs > $O/$TAG.json 2> $O/$TAG.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$TAG.json").read().strip().splitlines()[-1])
    r=d["roofline_in_flight"]; p=r["per_lane"][0]
    print("$TAG:", round(d["value"],1), "f/s", round(d["ms_per_step"],2), "ms | phase", round(r["rollout_phase_ms"],1), "| attn us", round(r["decode_attn_mean_launch_us_in_flight"],1), "gemm us", round(p["decode_gemm_mean_launch_us"],2), p.get("decode_gemm_mean_launch_us_by_kind"))
except Exception as e:
    print("$TAG failed", e); print(open("$O/$TAG.err").read()[-600:])
PY
}
run base1 IVG_DEV=0
run o48 IVG_DEV=1 IVG_INFLIGHT_KB=0,48,0,0,0
run q72 IVG_DEV=1 IVG_INFLIGHT_KB=72,0,0,0,0
run g72 IVG_DEV=1 IVG_INFLIGHT_KB=0,0,72,0,0
run d64 IVG_DEV=1 IVG_INFLIGHT_KB=0,0,0,64,0
run d128 IVG_DEV=1 IVG_INFLIGHT_KB=0,0,0,128,0
run l48 IVG_DEV=1 IVG_INFLIGHT_KB=0,0,0,0,48
run l96 IVG_DEV=1 IVG_INFLIGHT_KB=0,0,0,0,96
run oq IVG_DEV=1 IVG_INFLIGHT_KB=72,48,0,0,0
run all IVG_DEV=1 IVG_INFLIGHT_KB=72,48,72,64,48
run base2 IVG_DEV=0
echo done > $O/done.txt
