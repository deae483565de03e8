#!/bin/bash
# round 6, session 11: is the warm-up of the next decode GEMM's weights (2.4x counted traffic: Infinity-Cache re-reads) still worth it
# for one batch alone?  --lanes 1, warm-up on / off, ABAB
set -u
R=$(pwd); O=$R/gpurun_out/r06_s11; mkdir -p $O; export TMPDIR=/tmp
run () {
  TAG=$1; shift
  env "$@" timeout 400 python bench.py --lanes 1 --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs > $O/$TAG.json 2> $O/$TAG.err
  python - <<PY
import json
d=json.loads(open("$O/$TAG.json").read().strip().splitlines()[-1])
g=[r for r in [d["roofline"]]+d["roofline_other"] if "dgemm" in r["kernel"]][0]
print("$TAG:", round(d["value"],1), "f/s | stages", {k:round(v,1) for k,v in d["stage_ms"].items() if k in ("encode_ms","rollout_ms","decode_ms")}, "| gemm ms/step (stamps)", round(g["kernel_ms_per_step"],1), g.get("mean_launch_us_by_kind"))
PY
}
run warm1 IVG_DEV=0
run cold1 IVG_DEV=1 IVG_DG3_WARM=0
run warm2 IVG_DEV=0
run cold2 IVG_DEV=1 IVG_DG3_WARM=0
echo "config3 warm: $(timeout 300 python bench.py --config 3 --lanes 1 --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['stage_ms']['rollout_ms'],1))")"
echo "config3 cold: $(IVG_DEV=1 IVG_DG3_WARM=0 timeout 300 python bench.py --config 3 --lanes 1 --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['stage_ms']['rollout_ms'],1))")"
echo done > $O/done.txt
