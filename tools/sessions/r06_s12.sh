#!/bin/bash
# round 6, session 12: warm-up of the next decode GEMM's weights made conditional per shape (gate/up not warmed) -- ABAB on one lane,
# config 3, four lanes; then the FETCH_SIZE / WRITE_SIZE passes for the decode-GEMM class
set -u
R=$(pwd); O=$R/gpurun_out/r06_s12; mkdir -p $O; export TMPDIR=/tmp
run () {
  TAG=$1; shift
  env "$@" timeout 400 python bench.py --lanes 1 --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs > $O/$TAG.json 2> $O/$TAG.err
  python - <<PY
import json
d=json.loads(open("$O/$TAG.json").read().strip().splitlines()[-1])
g=[r for r in [d["roofline"]]+d["roofline_other"] if "dgemm" in r["kernel"]][0]
print("$TAG:", round(d["value"],1), "f/s | rollout", round(d["stage_ms"]["rollout_ms"],1), "| gemm ms/step (stamps)", round(g["kernel_ms_per_step"],1), g.get("mean_launch_us_by_kind"))
PY
}
run new1 IVG_DEV=0
run old1 IVG_DEV=1 IVG_WARM_GATE_UP=1
run new2 IVG_DEV=0
run old2 IVG_DEV=1 IVG_WARM_GATE_UP=1
for V in "IVG_DEV=0" "IVG_DEV=1 IVG_WARM_GATE_UP=1"; do
  echo "config3 [$V]: $(env $V timeout 300 python bench.py --config 3 --lanes 1 --steps 8 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['stage_ms']['rollout_ms'],1))")"
  echo "mbrl [$V]: $(env $V timeout 300 python tools/mbrl_bench.py 16 12 2>&1 | grep "reuse_cache=True")"
done
cd /tmp
PM="python $R/bench.py --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex 'decode_attn|conv3x3|igemm_kernel|gemm256|dgemm|dg3_kernel' -d /tmp/prof_$C -o p --output-format csv -- $PM > $O/pmc_$C.log 2>&1
  F=$(find /tmp/prof_$C -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && (cd $R/tools && python pmc_summary.py "$F" $O/pmc_$C.json > $O/pmc_$C.txt 2>&1)
  rm -rf /tmp/prof_$C
done
(cd $R && python tools/pmc_traffic.py $O/pmc_FETCH_SIZE.json $O/pmc_WRITE_SIZE.json $O/pmc_traffic.json "python bench.py --lanes 1 --steps 1 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-other-configs --no-profile" > $O/pmc_traffic.txt 2>&1)
cat $O/pmc_traffic.txt
echo done > $O/done.txt
