"""Soak of the round-5 loop: many-check runs of the public call, one after the other, in one process -- every run must
give the first run's numbers (attribution, error history: the estimator's normals are a function of the seed), the
engine's info word must stay 0, HBM must not leak, the rate must not decay.  Shapes: C2 (the one-call group path,
eight chunks a launch), C3 (two lanes, half-chunks), a stop on the deferred path, p = 150 with three ranks' dealing
emulated by lookahead.    python3 tools/soak_full_runs.py [seconds per shape]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ls-spa_amd"))
import numpy as np, ctypes as C
from ls_spa import ls_spa, release
hip = C.CDLL("libamdhip64.so")


def free_gb():
    f, t = C.c_size_t(), C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 1e9


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
for name, p, rows, kw in (("C2 full run", 100, 10000, dict(batch_size=128, num_batches=64, tolerance=0.0)),
                          ("C3 full run", 1000, 100000, dict(batch_size=128, num_batches=32, tolerance=0.0)),
                          ("p=150 lookahead 3", 150, 3000, dict(batch_size=16, max_samples=400, tolerance=0.0, lookahead=3)),
                          ("p=300 stop", 300, 5000, dict(batch_size=32, max_samples=2048, tolerance=None))):
    rng = np.random.default_rng(p)
    Xa, Xe = rng.standard_normal((rows, p)), rng.standard_normal((rows, p))
    w = rng.standard_normal(p) / np.sqrt(p)
    ya, ye = Xa @ w + rng.standard_normal(rows), Xe @ w + rng.standard_normal(rows)
    ref, n, t_start, times = None, 0, time.perf_counter(), []
    if kw["tolerance"] is None:      # a tolerance the run meets at its fifth check or so
        probe = ls_spa(Xa, Xe, ya, ye, method="argsort", seed=3, **dict(kw, tolerance=0.0, max_samples=512))
        kw = dict(kw, tolerance=float(probe.error_history[4]) * 1.0000001)
    free0 = None
    while time.perf_counter() - t_start < budget:
        t0 = time.perf_counter()
        r = ls_spa(Xa, Xe, ya, ye, method="argsort", seed=3, **kw)
        times.append(time.perf_counter() - t0)
        if ref is None:
            ref = r
        assert np.array_equal(ref.attribution, r.attribution) and np.array_equal(ref.error_history, r.error_history), name
        assert abs(r.attribution.sum() - r.r_squared) < 1e-9
        n += 1
        if n == 2:
            free0 = free_gb()
    print(f"{name}: {n} runs, {len(ref.error_history)} checks each, seconds first / median / last "
          f"{times[0]:.4f} / {np.median(times):.4f} / {times[-1]:.4f}, free HBM after run 2 / now {free0:.2f} / {free_gb():.2f} GB")
release()
print("free after release", round(free_gb(), 2))
