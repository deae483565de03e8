"""From a rocprofv3 kernel trace: how much of the decode attention's time runs concurrently with other kernels
(two-chain experiment), and the per-kernel mean durations of the decode step."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows)
dur = defaultdict(list)
for s, e, n in ev:
    dur[n].append((e - s) / 1e3)
print("kernel means (us):")
for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"  {n:60s} n={len(d):6d} mean={sum(d)/len(d):8.2f} total_ms={sum(d)/1e3:8.2f}")
# overlap: sweep
pts = []
for s, e, n in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy1 = busy2 = 0
level = defaultdict(int)
cur = 0; last = pts[0][0]
for t, d in pts:
    level[min(cur, 5)] += t - last
    if cur == 1: busy1 += t - last
    elif cur >= 2: busy2 += t - last
    cur += d; last = t
span = pts[-1][0] - pts[0][0]
print(f"time with exactly one kernel running {busy1/1e6:.2f} ms, with two or more {busy2/1e6:.2f} ms, span {span/1e6:.2f} ms")
print("kernels on the device at once (share of the span): " + "  ".join(f"{k}{'+' if k == 5 else ''}: {100 * v / span:.1f} %" for k, v in sorted(level.items())))
