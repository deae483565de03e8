#!/bin/bash
# round 5, session 5: sustained bf16 MFMA rate vs operand data and LDS fragment traffic, with socket power / shader clock sampled
# beside it (rocm-smi every 0.25 s)
set -u
R=$(pwd); O=$R/gpurun_out/r05_s5; mkdir -p $O
timeout 120 tools/ubench/bin/mfma_power 20 > $O/mfma_power_20ms.txt 2>&1; cat $O/mfma_power_20ms.txt
( while true; do echo "t=$(date +%s.%N) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Power \(W\)' | sed 's/GPU\[0\]\s*: //' | tr '\n' ' ')"; sleep 0.25; done ) > $O/smi_samples.txt 2>&1 &
SMI=$!
timeout 200 tools/ubench/bin/mfma_power 1500 > $O/mfma_power_1500ms.txt 2>&1
kill $SMI
cat $O/mfma_power_1500ms.txt
python3 - <<'PY' "$O"
import re, sys
O = sys.argv[1]
arms = []
for l in open(O + "/mfma_power_1500ms.txt"):
    m = re.search(r"^(.{50}).*wall ([\d.]+) \.\. ([\d.]+) s", l)
    if m: arms.append((m.group(1).strip(), float(m.group(2)), float(m.group(3))))
samples = []
for l in open(O + "/smi_samples.txt"):
    m = re.search(r"t=([\d.]+)", l)
    p = re.search(r"Power \(W\): ([\d.]+)", l)
    c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l)
    if m and p: samples.append((float(m.group(1)), float(p.group(1)), int(c.group(1)) if c else -1))
with open(O + "/power_by_arm.txt", "w") as f:
    for name, t0, t1 in arms:
        ss = [(p, c) for t, p, c in samples if t0 + 0.5 < t < t1]
        if ss:
            line = f"{name:50s} samples {len(ss):3d}  power W mean {sum(p for p, _ in ss) / len(ss):7.1f} max {max(p for p, _ in ss):7.1f}  sclk MHz mean {sum(c for _, c in ss) / len(ss):6.0f} min {min(c for _, c in ss)}"
            print(line); f.write(line + "\n")
PY
echo done > $O/done.txt
