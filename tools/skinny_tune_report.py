"""Match the kernel trace of tools/skinny_tune.py to its launch plan and print the median kernel time per configuration."""
import csv
import sys

trace, plan = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(trace)) if "skinny_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
i = 0
best = {}
for line in open(plan):
    name, mf, fn, w, rep = line.split()
    rep = int(rep)
    d = sorted(durs[i + 2:i + rep])   # drop the first two (cold) launches
    i += rep
    med = d[len(d) // 2]
    tag = "model" if mf == "0" else f"{mf},{fn},{w}"
    print(f"{name:7s} {tag:8s} {med:7.2f} us")
    if name not in best or med < best[name][0]:
        best[name] = (med, tag)
print("best:", best, "| launches matched", i, "of", len(durs))
