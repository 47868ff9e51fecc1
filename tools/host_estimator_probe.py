"""Host-side estimator cost against the BLAS thread count (developer tool; the GPU box has 256 cores and
OpenBLAS defaults to all of them, which is slow for these small factorisations)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ls-spa_amd"))
from ls_spa._stats import error_estimates, error_estimates_lowrank
from threadpoolctl import threadpool_limits
p, n = 1000, 128
rng = np.random.default_rng(0)
L = rng.standard_normal((n, p)) * 1e-3
c = L - L.mean(0)
cov = c.T @ c / n
for lim in (None, 1, 4, 8, 16, 32, 64):
    ctx = threadpool_limits(limits=lim) if lim else threadpool_limits(limits=None)
    with ctx:
        t0 = time.perf_counter(); error_estimates_lowrank(np.random.default_rng(1), c, n); t1 = time.perf_counter()
        error_estimates(np.random.default_rng(1), cov * n / (n - 1) / n); t2 = time.perf_counter()
    print(f"threads {lim}: lowrank {1e3*(t1-t0):7.1f} ms   reference {1e3*(t2-t1):7.1f} ms")
