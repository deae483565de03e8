#!/bin/bash
# round 5, session 15: with default-policy weight requests in the lanes mode -- is the warm-up of the next launch's weights still worth
# it, and which LDS budget?
set -u
R=$(pwd); O=$R/gpurun_out/r05_s15; mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout 400 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs --only-lanes $EXTRA > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open('$O/bench_$tag.json') if l.startswith('{')][0]
    r=d['roofline_in_flight']
    print('$tag: value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), '| gemm us', round(sum(p['decode_gemm_mean_launch_us'] for p in r['per_lane'])/len(r['per_lane']),1), 'attn us', round(sum(p['decode_attn_mean_launch_us'] for p in r['per_lane'])/len(r['per_lane']),1))
except Exception as e:
    print('$tag failed', e)
PY
}
EXTRA=""
run base X=1
run nowarm IVG_DG3_WARM=0
EXTRA="--lane-lds-kb 32"; run kb32 X=1
EXTRA="--lane-lds-kb 52"; run kb52 X=1
EXTRA="--lane-lds-kb 76"; run kb76 X=1
EXTRA=""; run base2 X=1
run nowarm2 IVG_DG3_WARM=0
EXTRA="--lanes 3"; run lanes3 X=1
EXTRA="--lanes 5"; run lanes5 X=1
echo done > $O/done.txt
