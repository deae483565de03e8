"""From a rocprofv3 kernel trace of the lanes loop: share of the steady-state window (default: 35 % .. 95 % of the span) with 0 / 1 / 2 ...
kernels on the device, and the longest idle gaps with the kernels that end / start them.
Usage: python tools/idle_report.py <kernel_trace.csv> [lo_frac hi_frac]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
lo_f = float(sys.argv[2]) if len(sys.argv) > 2 else 0.35
hi_f = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50], r.get("Queue_Id", "?")) for r in rows)
t0, t1 = ev[0][0], max(e for _, e, _, _ in ev)
lo, hi = t0 + lo_f * (t1 - t0), t0 + hi_f * (t1 - t0)
pts = []
for s, e, n, q in ev:
    if e < lo or s > hi:
        continue
    pts.append((max(s, lo), 1, n)); pts.append((min(e, hi), -1, n))
pts.sort(key=lambda p: (p[0], p[1]))
level = defaultdict(float)
cur, last, last_end_name = 0, lo, "-"
gaps = []
for t, d, n in pts:
    level[min(cur, 5)] += t - last
    if cur == 0 and d == 1 and t - last > 0:
        gaps.append((t - last, last_end_name, n))
    cur += d
    if d == -1:
        last_end_name = n
    last = t
level[min(cur, 5)] += hi - last
span = hi - lo
print(f"window {span / 1e6:.1f} ms of a {(t1 - t0) / 1e6:.1f} ms trace; kernels on the device at once: " +
      "  ".join(f"{k}{'+' if k == 5 else ''}: {100 * v / span:.1f} %" for k, v in sorted(level.items())))
gaps.sort(reverse=True)
tot = sum(g[0] for g in gaps)
print(f"idle gaps: {len(gaps)} totalling {tot / 1e6:.1f} ms; > 20 us: {sum(1 for g in gaps if g[0] > 2e4)} totalling {sum(g[0] for g in gaps if g[0] > 2e4) / 1e6:.1f} ms")
by = defaultdict(lambda: [0, 0.0])
for g, a, b in gaps:
    k = (a[:40], b[:40]); by[k][0] += 1; by[k][1] += g
print("idle time by (kernel that ended, kernel that started):")
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"  {t / 1e6:8.2f} ms in {n:6d} gaps (mean {t / n / 1e3:7.1f} us)  after {k[0]:40s} before {k[1]}")
