#!/bin/bash
# round 4, session 5: decode-GEMM footprint sweep with 4-6 batches in flight; x3 mode kernel trace
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_s5; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_x3.py tests/test_gpu_edges.py tests/test_gpu_callers.py -q --tb=short -p no:cacheprovider > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
B="python bench.py --steps 12 --warmup 1 --no-cpu-baseline --no-fp32-mode --no-profile --no-other-configs"
R=$O/lanes.txt; : > $R
run() { echo "== $1" >> $R; shift; timeout 300 env "$@" 2>>$O/lanes.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); s = d['stage_ms']; sl = d.get('single_lane', {})
        print(round(d['value'],1), 'f/s', round(d['ms_per_step'],2), 'ms/step | single', round(sl.get('value',0),1), 'median ms', round(sl.get('ms_per_step_median',0),2), '| stages', round(s['encode_ms'],1), round(s['rollout_ms'],1), round(s['decode_ms'],1))" >> $R; }
run "lanes4 lds52"      IVG_DECODE_LDS_KB=52 $B --lanes 4
run "lanes4 lds40"      IVG_DECODE_LDS_KB=40 $B --lanes 4
run "lanes4 lds32"      IVG_DECODE_LDS_KB=32 $B --lanes 4
run "lanes4 lds24"      IVG_DECODE_LDS_KB=24 $B --lanes 4
run "lanes5 lds52"      IVG_DECODE_LDS_KB=52 $B --lanes 5
run "lanes5 lds32"      IVG_DECODE_LDS_KB=32 $B --lanes 5
run "lanes6 lds32"      IVG_DECODE_LDS_KB=32 $B --lanes 6
run "lanes4 dg3off"     IVG_DG3=0 $B --lanes 4
run "lanes4 dg3off lds52" IVG_DG3=0 IVG_DECODE_LDS_KB=52 $B --lanes 4
run "lanes4 dg3off lds32" IVG_DG3=0 IVG_DECODE_LDS_KB=32 $B --lanes 4
run "lanes4 lds52 nowarm" IVG_DECODE_LDS_KB=52 IVG_DG3_WARM=0 $B --lanes 4
cat $R
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_x3 -o x3 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --lanes 1 --decode-dtype x3 --llm-dtype x3 --no-cpu-baseline --no-fp32-mode --no-profile --no-other-configs > $GRAFT_REPO_ROOT/$O/x3_trace_run.json 2> $GRAFT_REPO_ROOT/$O/x3_trace.err
ST=$(find /tmp/prof_x3 -name "*kernel_stats.csv" | head -1)
cd $GRAFT_REPO_ROOT
[ -n "$ST" ] && head -40 "$ST" > $O/x3_kernel_stats.csv
cut -c1-150 $O/x3_kernel_stats.csv | head -32
# host CPU baseline: how long does the oracle take on all cores vs 32 / 64 threads (sample of 4 trajectories)
for th in 32 64 256; do timeout 200 env OMP_NUM_THREADS=$th MKL_NUM_THREADS=$th HIP_VISIBLE_DEVICES= python bench.py --cpu-baseline-worker --res 64 --frames 16 --cpu-sample 4 --cpu-threads $th 2>/dev/null | cut -c1-200; done
grep -i "error\|Traceback" -A8 $O/lanes.err | head -30
