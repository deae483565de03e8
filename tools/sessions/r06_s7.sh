#!/bin/bash
# round 6, session 7: what the four-lane mode pays per decode-GEMM kind (new per-kind stamps) and two A/Bs of the batches-in-flight
# profile: prompt-pass GEMMs without the whole-CU gemm256l workgroups, second-generation decode GEMMs capped to one / two row tiles
set -u
R=$(pwd); O=$R/gpurun_out/r06_s7; mkdir -p $O; export TMPDIR=/tmp
run () {  # $1 tag, rest: env
  TAG=$1; shift
  env "$@" timeout 400 python bench.py --only-lanes --steps 12 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs > $O/$TAG.json 2> $O/$TAG.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$TAG.json").read().strip().splitlines()[-1])
    r=d["roofline_in_flight"]; p=r["per_lane"][0]
    print("$TAG:", round(d["value"],1), "f/s", round(d["ms_per_step"],2), "ms | in flight", round(r["frac"],3), "phase", round(r["rollout_phase_ms"],1), "| attn us", round(r["decode_attn_mean_launch_us_in_flight"],1), "gemm us", round(p["decode_gemm_mean_launch_us"],2), p.get("decode_gemm_mean_launch_us_by_kind"))
except Exception as e:
    print("$TAG failed", e); print(open("$O/$TAG.err").read()[-600:])
PY
}
run base1 IVG_DEV=0
run nog256 IVG_DEV=1 IVG_INFLIGHT_GEMM256=0
run mf1 IVG_DEV=1 IVG_DG2_MF_CAP=1
run mf2 IVG_DEV=1 IVG_DG2_MF_CAP=2
run both IVG_DEV=1 IVG_INFLIGHT_GEMM256=0 IVG_DG2_MF_CAP=1
run base2 IVG_DEV=0
echo done > $O/done.txt
