#!/usr/bin/env python3
"""Per-kernel totals and a slice of the timeline from a rocprofv3 result database (rocpd, the default output format):
    python3 tools/rocpd_summary.py gpurun_out/<dir>/x_results.db [first_row [rows]]"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
agg = collections.defaultdict(list)
for n, s, e, _ in rows:
    agg[n.split("(")[0][:64]].append((e - s) / 1e3)
print(f"{len(rows)} dispatches")
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:64s} n={len(v):6d} avg={sum(v) / len(v):9.2f} us  total={sum(v) / 1e3:9.3f} ms")
if len(sys.argv) > 2:
    lo = int(sys.argv[2])
    hi = lo + (int(sys.argv[3]) if len(sys.argv) > 3 else 40)
    t0 = rows[lo][1]
    for n, s, e, st in rows[lo:hi]:
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  s{st}  {n.split('(')[0][:56]}")
