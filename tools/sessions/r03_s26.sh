#!/bin/bash
# two lanes with the gathers going through RCCL (1-rank group, IVG_FORCE_COLLECTIVE=1): collectives issued from two host threads in ticket order
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out/r03_s26.txt; : > $O
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
IVG_FORCE_COLLECTIVE=1 timeout 300 python bench.py --gpus 1 --steps 12 --warmup 2 --lanes 2 --no-cpu-baseline --no-fp32-mode --no-profile > gpurun_out/r03_s26_forced.json 2> gpurun_out/r03_s26.err; echo "rc=$?" >> $O
python - >> $O <<'PY'
import json
for l in open('gpurun_out/r03_s26_forced.json'):
    if l.startswith('{'):
        d = json.loads(l); print('forced RCCL gathers, lanes 2:', round(d['value'], 1), 'f/s', round(d['ms_per_step'], 2), 'ms/step; single', round(d['single_lane']['value'], 1))
PY
unset MASTER_ADDR MASTER_PORT RANK LOCAL_RANK WORLD_SIZE
timeout 300 python -m pytest tests/test_gpu_evaluate.py -q -m gpu -k "lanes or rccl or flight" 2>&1 | tail -2 >> $O
cat $O; grep -i "error\|nccl" gpurun_out/r03_s26.err | head -5
