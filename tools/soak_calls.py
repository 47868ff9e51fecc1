import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "ls-spa_amd"))
import numpy as np, ctypes as C
from ls_spa import ls_spa, release
hip = C.CDLL("libamdhip64.so")
def free_gb():
    f, t = C.c_size_t(), C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 1e9
rng = np.random.default_rng(0)
p, n = 400, 4000
Xa, Xe = rng.standard_normal((n, p)), rng.standard_normal((n, p)); w = rng.standard_normal(p)
ya, ye = Xa @ w + rng.standard_normal(n), Xe @ w + rng.standard_normal(n)
print("free before", round(free_gb(), 2))
ref = None
for i in range(12):
    t0 = time.perf_counter()
    r = ls_spa(Xa, Xe, ya, ye, method="argsort", batch_size=64, num_batches=6, tolerance=0.0, seed=3)
    dt = time.perf_counter() - t0
    if ref is None: ref = r.attribution
    assert np.array_equal(ref, r.attribution)
    print(i, round(dt, 3), "free", round(free_gb(), 2), len(r.error_history))
release()
print("free after release", round(free_gb(), 2))
