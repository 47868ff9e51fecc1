"""Developer tool (GPU box): lifts under a set of developer flags against the default and the oracle.

    python tools/flag_parity.py 4096[,flags...]
"""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "ls-spa_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lsspa_oracle as O
from ls_spa._engine import HipEngine

flag_sets = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4096").split(",")]
for (p, n, m, prec) in [(130, 500, 400, "float64"), (300, 800, 700, "float64"), (515, 1500, 1400, "float64"),
                        (1000, 3000, 2600, "float64"), (1000, 3000, 2600, "float32"), (1300, 3000, 2900, "float64")]:
    rng = np.random.default_rng(p)
    Xa = rng.standard_normal((n, p)); Xe = rng.standard_normal((m, p))
    th = rng.standard_normal(p)
    ya = Xa @ th + rng.standard_normal(n); ye = Xe @ th + rng.standard_normal(m)
    perms = np.array([rng.permutation(p) for _ in range(4)])
    red = O.reduce(Xa, Xe, ya, ye, 1e-3)
    yy = float(ye @ ye)
    want = np.array([O.sample_lift(*red, yy, o, True) for o in perms[:2]])
    eng = HipEngine(0)
    eng.set_precision(prec)
    eng.load_data(Xa, Xe, ya, ye, 1e-3)
    eng.set_flags(1024)
    base = eng.run_batch(perms, True, want_lifts=True, accumulate=False)
    print(f"p={p} {prec}: default vs oracle {np.abs(base[:2] - want).max():.3e}", end="")
    for f in flag_sets:
        eng.set_flags(1024 | f)
        got = eng.run_batch(perms, True, want_lifts=True, accumulate=False)
        print(f" | flags {f}: vs default {np.abs(got - base).max():.3e} vs oracle {np.abs(got[:2] - want).max():.3e} info {eng.info()}", end="")
    print()
    eng.close()
