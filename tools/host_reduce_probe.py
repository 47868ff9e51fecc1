"""The streamed reduction from host arrays at the C3 shape: seconds per part (lsspa_reduce_timing), with page-locking of
the caller's X (default) and without (developer flag 4096)."""
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
eng = HipEngine(0)
for flags in (0, 4096, 0, 4096, 0, 4096):
    eng.set_flags(flags)
    t0 = time.perf_counter()
    eng.load_data(Xa, Xe, ya, ye, 0.0)
    dt = time.perf_counter() - t0
    parts = eng.reduce_timing()
    print(f"flags {flags:5d}: {1e3 * dt:7.2f} ms  ({2 * Xa.nbytes / dt / 1e9:5.1f} GB/s)  "
          + "  ".join(f"{k} {1e3 * v:.3f} ms" for k, v in parts.items()))
eng.close()
