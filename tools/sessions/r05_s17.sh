#!/bin/bash
# round 5, session 17: the decoders' convolutions on PART of the chip (the persistent kernel with a grid of 96-192 workgroups, one per
# CU, conv phases of the lanes serialised by the phase gate) while the other lanes' rollouts keep the remaining CUs -- half a chip of
# MFMAs is not power-capped, and a conv grid that leaves CUs free no longer stops the rollouts
set -u
R=$(pwd); O=$R/gpurun_out/r05_s17; mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout 400 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs --only-lanes --no-profile $EXTRA > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
try:
    d=[json.loads(l) for l in open('$O/bench_$tag.json') if l.startswith('{')][0]
    print('$tag: value', round(d['value'],1), 'ms', round(d['ms_per_step'],2), '| decode alone', round(d['stage_ms']['decode_ms'],1))
except Exception as e:
    print('$tag failed', e); print(open('$O/bench_$tag.err').read()[-400:])
PY
}
EXTRA=""; run base X=1
EXTRA="--conv-gate 1"; run wide128_gate IVG_CONV_WIDE=2 IVG_CONV_WIDE_GRID=128
EXTRA=""; run wide128_nogate IVG_CONV_WIDE=2 IVG_CONV_WIDE_GRID=128
EXTRA="--conv-gate 1"; run wide160_gate IVG_CONV_WIDE=2 IVG_CONV_WIDE_GRID=160
EXTRA="--conv-gate 1"; run wide192_gate IVG_CONV_WIDE=2 IVG_CONV_WIDE_GRID=192
EXTRA="--conv-gate 1"; run wide96_gate IVG_CONV_WIDE=2 IVG_CONV_WIDE_GRID=96
EXTRA="--conv-gate 1"; run wide256_gate IVG_CONV_WIDE=2
EXTRA=""; run base2 X=1
echo done > $O/done.txt
