#!/bin/bash
# round 5, session 18: socket power and shader clock over the four-lane loop (and over a one-lane loop): is the WHOLE step energy-bound?
set -u
R=$(pwd); O=$R/gpurun_out/r05_s18; mkdir -p $O
sample() { while true; do echo "t=$(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Power \(W\)' | sed 's/GPU\[0\]\s*: //' | tr '\n' ' ')"; sleep 0.2; done; }
for L in 4 1; do
  sample > $O/smi_l$L.txt 2>&1 &
  SMI=$!
  date +%s.%N > $O/t0_l$L.txt
  timeout 400 python bench.py --lanes $L --steps 60 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs --only-lanes --no-profile > $O/bench_l$L.json 2> $O/bench_l$L.err
  date +%s.%N > $O/t1_l$L.txt
  kill $SMI
done
python - <<PY
import json, re
for L in (4, 1):
    d=[json.loads(l) for l in open('$O/bench_l%d.json' % L) if l.startswith('{')][0]
    t1=float(open('$O/t1_l%d.txt' % L).read()); dur=d['ms_per_step']*d['steps']/1e3
    rows=[]
    for l in open('$O/smi_l%d.txt' % L):
        m=re.search(r't=([\d.]+)', l); p=re.search(r'Power \(W\): ([\d.]+)', l); c=re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', l)
        if m and p: rows.append((float(m.group(1)), float(p.group(1)), int(c.group(1)) if c else -1))
    # the timed loop is the last dur seconds before the process printed its line and exited (~1 s of teardown)
    sel=[(p,c) for t,p,c in rows if t1-1.5-dur < t < t1-1.5]
    pw=[p for p,_ in sel]; ck=[c for _,c in sel]
    hist={}
    for p in pw: hist[int(p//100)*100]=hist.get(int(p//100)*100,0)+1
    print('lanes %d: value %.1f f/s (%.1f ms/step, %d steps) | %d samples over the timed loop: power W mean %.0f max %.0f min %.0f | sclk MHz mean %.0f | histogram (W: samples) %s' % (L, d['value'], d['ms_per_step'], d['steps'], len(sel), sum(pw)/max(1,len(pw)), max(pw or [0]), min(pw or [0]), sum(ck)/max(1,len(ck)), dict(sorted(hist.items()))))
PY
echo done > $O/done.txt
