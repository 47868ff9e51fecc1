"""The streamed reduction from host arrays at the C3 shape: seconds per part (lsspa_reduce_timing) of repeated calls,
on a fresh engine per call (what the public ls_spa() does) and on one engine."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ls-spa_amd"))
from ls_spa._engine import HipEngine  # noqa: E402

p, n = 1000, 100000
rng = np.random.default_rng(0)
Xa, Xe = rng.standard_normal((n, p)), rng.standard_normal((n, p))
ya, ye = rng.standard_normal(n), rng.standard_normal(n)
for fresh in (True, True, True, False, False, False):
    if fresh or "eng" not in globals() or eng is None:
        t0 = time.perf_counter()
        eng = HipEngine(0)
        t_create = time.perf_counter() - t0
    t0 = time.perf_counter()
    eng.load_data(Xa, Xe, ya, ye, 0.0)
    dt = time.perf_counter() - t0
    parts = eng.reduce_timing()
    print(f"fresh engine {fresh}: create {1e3 * t_create:6.2f} ms  load_data {1e3 * dt:7.2f} ms  ({2 * Xa.nbytes / dt / 1e9:5.1f} GB/s)  "
          + "  ".join(f"{k} {1e3 * v:.3f} ms" for k, v in parts.items()))
    if fresh:
        t0 = time.perf_counter()
        eng.close()
        eng = None
        print(f"   close {1e3 * (time.perf_counter() - t0):.2f} ms")
