"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per launch for every (kernel, grid) pair.
Usage: python tools/pmc_summary.py <counter_collection.csv> <out.json>

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of
16 B/lane streaming reads at 64 bytes (MI355X_MICROARCH.md, HBM section) -- the x2 correction is applied by the consumer
of this summary (profiles/r01_pmc_traffic.json states it), not here."""
import csv
import json
import sys
from collections import defaultdict

from trace_summary import short


def main():
    path, out = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "0"
            wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or "1"
            key = (short(r["Kernel_Name"]), int(grid) // max(1, int(wg)), r["Counter_Name"])
            a = acc[key]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    rows = [{"kernel": k[0], "blocks": k[1], "counter": k[2], "launches": n, "mean": s / n, "total": s}
            for k, (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1])]
    with open(out, "w") as f:
        json.dump(rows, f, indent=1)
    for r in rows[:25]:
        print(f"{r['kernel']:58s} {r['blocks']:8d} {r['counter']:12s} n={r['launches']:7d} mean={r['mean']:14.1f} total={r['total']:16.1f}")


if __name__ == "__main__":
    main()
