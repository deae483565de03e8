#!/bin/bash
# round-3 GPU session 17: two bench processes side by side on one GPU (with and without CU masks): does the MFMA-bound decode of one batch
# overlap the latency / HBM-bound rollout of another?
set -u
O=gpurun_out/r03_s17; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-profile"
echo "== single" > $O/out.txt; timeout 300 $B >> $O/out.txt 2>&1
echo "== two processes, no masks" >> $O/out.txt
( timeout 400 $B > $O/a0.txt 2>&1 & timeout 400 $B > $O/b0.txt 2>&1 & wait )
cat $O/a0.txt $O/b0.txt >> $O/out.txt
echo "== two processes, HSA_CU_MASK halves" >> $O/out.txt
( HSA_CU_MASK=0:0-127 timeout 400 $B > $O/a1.txt 2>&1 & HSA_CU_MASK=0:128-255 timeout 400 $B > $O/b1.txt 2>&1 & wait )
cat $O/a1.txt $O/b1.txt >> $O/out.txt
echo "== two processes, HSA_CU_MASK interleaved (even / odd CUs)" >> $O/out.txt
EV=$(python -c "print(','.join(str(i) for i in range(0,256,2)))"); OD=$(python -c "print(','.join(str(i) for i in range(1,256,2)))")
( HSA_CU_MASK=0:$EV timeout 400 $B > $O/a2.txt 2>&1 & HSA_CU_MASK=0:$OD timeout 400 $B > $O/b2.txt 2>&1 & wait )
cat $O/a2.txt $O/b2.txt >> $O/out.txt
python - <<'PY'
import json
for l in open("gpurun_out/r03_s17/out.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print("   ", round(d["value"],1), "frames/s", round(d["ms_per_step"],1), "ms/step", {k:round(v,1) for k,v in d["stage_ms"].items() if k.endswith("_ms")})
PY
echo done > $O/done.txt
