#!/bin/bash
# round 4, session 12: wave-uniform GroupNorm map of conv3x3 -- parity, per-shape timing (the timed call includes the two tiny statistics kernels), bench A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_s12; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -k "fused_input_groupnorm or x3_split" > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
R=$O/shapes.txt; : > $R
for U in 0 1; do
  for S in "64 128 128" "32 256 256" "16 512 512" "32 512 256" "64 256 128"; do
    IVG_GNA_UNIFORM=$U timeout 100 python tools/conv_bench.py $S 0 896 bf16 1 >> $R 2>>$O/shapes.err
  done
  IVG_GNA_UNIFORM=$U timeout 100 python tools/conv_bench.py 64 128 128 0 128 fp32 1 >> $R 2>>$O/shapes.err
done
cat $R
B="python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile --no-other-configs"
for U in 0 1; do
  IVG_GNA_UNIFORM=$U timeout 300 $B 2>>$O/bench.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stage_ms']; sl = d['single_lane']
        print('IVG_GNA_UNIFORM=$U:', round(d['value'],1), 'f/s | single', round(sl['value'],1), '| stages', round(s['encode_ms'],2), round(s['rollout_ms'],1), round(s['decode_ms'],2))"
done
grep -i "error\|Traceback" -A5 $O/shapes.err $O/bench.err | head -20
