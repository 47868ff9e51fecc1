"""Where the first ls_spa(method='argsort') call of a process spends its time (developer tool)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ls-spa_amd"))
import numpy as np
from ls_spa import ls_spa
p, rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100, 20000
rng = np.random.default_rng(0)
Xa, Xe = rng.standard_normal((rows, p)), rng.standard_normal((rows, p))
th = rng.standard_normal(p)
ya, ye = Xa @ th + rng.standard_normal(rows), Xe @ th + rng.standard_normal(rows)
for name, kw in (("random/reference", dict(method="random")), ("argsort/device #1", dict(method="argsort")),
                 ("argsort/device #2", dict(method="argsort")), ("argsort/device anti=0", dict(method="argsort", antithetical=False)),
                 ("argsort/device history", dict(method="argsort", return_history=True))):
    tm = {}
    t0 = time.perf_counter()
    ls_spa(Xa, Xe, ya, ye, max_samples=2048, batch_size=256, tolerance=1e-8, seed=42, _timings=tm, **kw)
    print(name, round(time.perf_counter() - t0, 4), {k: round(v, 4) for k, v in tm.items() if v > 0.002})
