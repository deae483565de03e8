"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid, workgroup) -- the names alone do not separate the
shapes one templated kernel is launched with.  Usage: python tools/trace_summary.py <kernel_trace.csv> [passes]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN3ivg(\d+)([a-z_0-9]+)", name)
    if m:
        return m.group(2)[:int(m.group(1))] + re.sub(r".*?kernel", "", name)[:24]
    return name[:60]


def main():
    path = sys.argv[1]
    passes = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    d = defaultdict(list)
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]),
               int(r["Workgroup_Size_X"]))
        d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in d.values())
    print(f"{'kernel':58s} {'blocks':>8s} {'gy':>5s} {'wg':>5s} {'calls/pass':>10s} {'med us':>9s} {'mean us':>9s} {'ms/pass':>9s} {'%':>6s}")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        if sum(v) / tot < 0.002:
            continue
        print(f"{k[0]:58s} {k[1]:8d} {k[2]:5d} {k[3]:5d} {len(v) / passes:10.1f} {v[len(v) // 2] / 1e3:9.2f} {sum(v) / len(v) / 1e3:9.2f} "
              f"{sum(v) / passes / 1e6:9.2f} {100 * sum(v) / tot:6.2f}")
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    print(f"kernel time {tot / 1e6:.1f} ms over a span of {span / 1e6:.1f} ms ({len(rows)} launches)")


if __name__ == "__main__":
    main()
