"""Time the reduction from HOST arrays (PCIe-inclusive) at the C3 shape."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ls-spa_amd"))
from ls_spa._engine import HipEngine
p, N = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(0)
Xa = rng.standard_normal((N, p)); Xe = rng.standard_normal((N, p))
w = rng.standard_normal(p); ya = Xa @ w + rng.standard_normal(N); ye = Xe @ w + rng.standard_normal(N)
eng = HipEngine(0)
for it in range(3):
    t0 = time.perf_counter(); eng.load_data(Xa, Xe, ya, ye, 0.0); eng.synchronize(); dt = time.perf_counter() - t0
    print(f"host reduce p={p} N=M={N}: {1e3*dt:.1f} ms  ({2*N*p*8/dt/1e9:.1f} GB/s over PCIe incl. Gram)")
G, g, H, h = eng.gram()
print("max |G - ref|", np.abs(G - Xa.T @ Xa / N).max(), "max |h - ref| rel", np.abs(h - Xe.T @ ye).max() / np.abs(h).max())
