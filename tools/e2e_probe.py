"""End-to-end wall time of ls_spa() at the C3 shape from host arrays (developer tool): engine creation,
reduction streamed over PCIe, sampler construction, sampling loop until the 1e-2 tolerance, final fit."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ls-spa_amd"))
from ls_spa import ls_spa
from ls_spa._engine import HipEngine
from ls_spa._driver import run_estimator, prepare_sampling
p, n = 1000, 100000
rng = np.random.default_rng(0)
Xa = rng.standard_normal((n, p)); Xe = rng.standard_normal((n, p)); w = rng.standard_normal(p)
ya = Xa @ w + rng.standard_normal(n); ye = Xe @ w + rng.standard_normal(n)
for est in ("device", "lowrank", "reference"):
    for rep in range(3):
        t0 = time.perf_counter()
        r = ls_spa(Xa, Xe, ya, ye, method="argsort", batch_size=128, num_batches=128, tolerance=1e-2, seed=42,
                   error_estimator=est)
        dt = time.perf_counter() - t0
    print(f"{est:9s}: {dt*1e3:7.1f} ms end to end, error {r.overall_error:.2e}, checks {len(r.error_history)}")
for rep in range(3):
    T = [time.perf_counter()]
    eng = HipEngine(0); T.append(time.perf_counter())
    prep = prepare_sampling(p, max_samples=128 * 128, batch_size=128, seed=42, perms=None, antithetical=True, method="argsort")
    T.append(time.perf_counter())
    eng.load_data(Xa, Xe, ya, ye, 0.0); T.append(time.perf_counter())
    run_estimator(eng, p, max_samples=128 * 128, batch_size=128, tolerance=1e-2, seed=42, perms=None, antithetical=True,
                  return_attribution_history=False, method="argsort", error_estimator="device", prepared=prep)
    T.append(time.perf_counter())
    eng.full_fit(); T.append(time.perf_counter())
    eng.close(); T.append(time.perf_counter())
    names = ["create", "prepare", "load_data", "run_estimator", "full_fit", "close"]
    print("  ".join(f"{nm} {1e3*(b-a):.1f}" for nm, a, b in zip(names, T[:-1], T[1:])), "ms")
