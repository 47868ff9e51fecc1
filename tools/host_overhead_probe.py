"""Where the host time of one small step goes (developer tool, GPU box only)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ls-spa_amd"))
from ls_spa._engine import HipEngine
from ls_spa import workloads
p, rows, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
d = workloads.gaussian(p, rows, rows, seed=0)
eng = HipEngine(0)
eng.load_data(*d, 0.0)
rng = np.random.default_rng(0)
perms = np.stack([rng.permutation(p) for _ in range(B)]).astype(np.int32)
def T(f, n=200):
    f(); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter(); eng.synchronize(); t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n
print("run_batch+merge      host %.1f us  wall %.1f us" % T(lambda: (eng.run_batch(perms, True), eng.merge())))
print("run_batch (no accum) host %.1f us  wall %.1f us" % T(lambda: eng.run_batch(perms, True, accumulate=False)))
print("run_batch (accum)    host %.1f us  wall %.1f us" % T(lambda: eng.run_batch(perms, True, accumulate=True)))
print("merge                host %.1f us  wall %.1f us" % T(lambda: eng.merge()))
t = eng.launch_batch(np.concatenate([perms] * 8), True)
eng.discard_batch(t)
def group():
    tk = eng.launch_batch(big, True)
    for j in range(8):
        eng.collect_batch(tk, accumulate=True, first=j * B, count=B); eng.merge()
big = np.concatenate([perms] * 8)
h, w = T(group, 50)
print("8 steps launched together: host %.1f us  wall %.1f us per step" % (h / 8, w / 8))
print("collect+merge        host %.1f us" % (T(lambda: None)[0]))
