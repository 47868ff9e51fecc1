"""GPU occupancy of a two-lane run from a rocprofv3 --kernel-trace csv: fraction of the wall time with 0 / 1 / >= 2 kernels
in flight over a window of the trace (fractions of the kernel count, default the middle third), and the longest gaps.
    python3 tools/trace_overlap.py <kernel_trace.csv> [lo hi]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))
        if "lsspa" in r["Kernel_Name"] and "gram" not in r["Kernel_Name"]]
rows.sort()
lo_f, hi_f = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (1 / 3, 2 / 3)
t_lo = rows[int(len(rows) * lo_f)][0]
t_hi = rows[int(len(rows) * hi_f)][1]
ev = []
for s, e, _ in rows:
    s, e = max(s, t_lo), min(e, t_hi)
    if e > s:
        ev += [(s, 1), (e, -1)]
ev.sort()
cur, last, hist, gaps = 0, t_lo, {}, []
for t, d in ev:
    hist[min(cur, 2)] = hist.get(min(cur, 2), 0) + (t - last)
    if cur == 0 and t - last > 0:
        gaps.append(t - last)
    cur += d
    last = t
tot = sum(hist.values())
print("window %.1f ms: idle %.3f  one kernel %.3f  two or more %.3f" % (tot / 1e6, hist.get(0, 0) / tot, hist.get(1, 0) / tot, hist.get(2, 0) / tot))
gaps.sort(reverse=True)
print("idle gaps: %d, total %.1f us, longest %s us" % (len(gaps), sum(gaps) / 1e3, [round(g / 1e3, 1) for g in gaps[:8]]))
