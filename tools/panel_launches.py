"""Average duration of each chol_panel2 launch of a step (Jo = 0 ..), from a rocprofv3 --kernel-trace csv.
    python3 tools/panel_launches.py <kernel_trace.csv> <launches per step>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
per = int(sys.argv[2])
names = collections.OrderedDict()
pan = [r for r in rows if "chol_panel2" in r["Kernel_Name"]]
pan.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in pan]
n = len(d) // per
d = d[len(d) - n * per:]
skip = n // 3
for j in range(per):
    v = [d[s * per + j] for s in range(skip, n)]
    print("launch %d: %.1f us (min %.1f)  grid %s" % (j, sum(v) / len(v), min(v), pan[len(pan) - per + j].get("Grid_Size", "?")))
print("sum %.1f us" % sum(sum(d[s * per + j] for j in range(per)) / 1 for s in range(skip, n) ) if False else "")
tot = [sum(d[s * per:(s + 1) * per]) for s in range(skip, n)]
print("panel per step: %.1f us" % (sum(tot) / len(tot)))
for k in ("strip2", "lift_partial", "gather", "chol_diag2"):
    v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if k in r["Kernel_Name"]]
    if v: print("%s: %.1f us avg over %d" % (k, sum(v[len(v)//3:]) / len(v[len(v)//3:]), len(v)))
