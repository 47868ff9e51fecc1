"""Where does the time to tolerance go?  Wall-clock of each host-visible stage of one check
(p = 1000, one batch of 128 antithetical samples), device-side estimator."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ls-spa_amd"))
import torch
from ls_spa._engine import HipEngine
from ls_spa import _samplers as S

p, rows, B = 1000, 100_000, 128
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
Xa = torch.randn(rows, p, dtype=torch.float64, device=dev, generator=g)
Xe = torch.randn(rows, p, dtype=torch.float64, device=dev, generator=g)
th = torch.randn(p, dtype=torch.float64, device=dev, generator=g)
ya = Xa @ th + torch.randn(rows, dtype=torch.float64, device=dev, generator=g)
ye = Xe @ th + torch.randn(rows, dtype=torch.float64, device=dev, generator=g)
torch.cuda.synchronize()
eng = HipEngine(0)
eng.load_device_data(Xa.data_ptr(), p, ya.data_ptr(), rows, Xe.data_ptr(), p, ye.data_ptr(), rows, p, 0.0)
eng.synchronize()

def stage(name, fn, sync=True):
    t0 = time.perf_counter(); r = fn()
    if sync: eng.synchronize()
    print(f"  {name:28s} {(time.perf_counter() - t0) * 1e3:8.3f} ms"); return r

for rep in range(3):
    print("rep", rep)
    T0 = time.perf_counter()
    rng = np.random.default_rng(42)
    stage("reset_stats", eng.reset_stats)
    stage("history_enable", lambda: eng.history_enable(B * 128))
    src = S.ArgsortSource(p, 42, B * 128)
    chunk = stage("sampler.take", lambda: src.take(B), sync=False)
    stage("run_batch", lambda: eng.run_batch(chunk, True, want_lifts=False, accumulate=True))
    stage("merge", eng.merge)
    stage("stats(mean)", lambda: eng.stats(want_cov=False))
    xi = stage("standard_normal", lambda: rng.standard_normal((1024, B)), sync=False)
    stage("error_draws", lambda: eng.error_draws(xi, B))
    stage("error_quantiles", eng.error_quantiles)
    stage("stats(final)", lambda: eng.stats(want_cov=False))
    print(f"  total {(time.perf_counter() - T0) * 1e3:.3f} ms")
eng.close()
