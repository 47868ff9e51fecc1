#!/usr/bin/env python3
"""The public call over a many-check run, alone in a process (for rocprofv3 --kernel-trace / quick A/B):
    python3 tools/full_run_probe.py [p rows batches [lanes [lookahead]]]
Prints the loop's rate and the host seconds per phase."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ls-spa_amd"))
import numpy as np  # noqa: E402
from ls_spa import ls_spa  # noqa: E402

p = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 64
lanes = sys.argv[4] if len(sys.argv) > 4 else "auto"
look = sys.argv[5] if len(sys.argv) > 5 else None
rng = np.random.default_rng(0)
Xa, Xe = rng.standard_normal((rows, p)), rng.standard_normal((rows, p))
th = rng.standard_normal(p)
ya, ye = Xa @ th + rng.standard_normal(rows), Xe @ th + rng.standard_normal(rows)
kw = dict(method="argsort", batch_size=128, num_batches=nb, tolerance=0.0, seed=42,
          lanes=lanes if lanes == "auto" else int(lanes))
if look is not None:
    kw["lookahead"] = look if look == "auto" else int(look)
for rep in range(int(os.environ.get("LSSPA_PROBE_REPS", "3"))):
    tm = {}
    t0 = time.perf_counter()
    r = ls_spa(Xa, Xe, ya, ye, _timings=tm, **kw)
    dt = time.perf_counter() - t0
    loop = tm["sampler"] + tm["estimator"] + tm["sampling"]
    print(json.dumps({"rep": rep, "seconds": round(dt, 5), "loop_s": round(loop, 5),
                      "orderings_per_s": round(2 * 128 * nb / loop), "checks": len(r.error_history),
                      "sum": float(r.attribution.sum()), "r2": float(r.r_squared),
                      "host": {k: round(v, 5) for k, v in tm.items() if k in ("sampler", "estimator", "sampling")}}))
