"""Ad-hoc timing of the hot path at benchmark scale (developer tool, GPU box only)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ls-spa_amd"))
import torch
from ls_spa._engine import HipEngine

p = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
torch.manual_seed(0)
dev = torch.device("cuda:0")
Xa = torch.randn(N, p, dtype=torch.float64, device=dev)
Xe = torch.randn(N, p, dtype=torch.float64, device=dev)
th = torch.randn(p, dtype=torch.float64, device=dev)
ya = Xa @ th + torch.randn(N, dtype=torch.float64, device=dev)
ye = Xe @ th + torch.randn(N, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
eng = HipEngine(0)
if os.environ.get("LSSPA_F32") == "1":
    eng.set_precision("float32")
eng.profile(True)
t0 = time.perf_counter()
eng.load_device_data(Xa.data_ptr(), p, ya.data_ptr(), N, Xe.data_ptr(), p, ye.data_ptr(), N, p, 0.0)
eng.synchronize()
print(f"reduce: {1e3*(time.perf_counter()-t0):.2f} ms", eng.profile_read()["gram"])
G = (Xa.T @ Xa / N).cpu().numpy()
Gd, gd, Hd, hd = eng.gram()
print("gram max err", np.abs(Gd - G).max())
t0 = time.perf_counter(); theta, r2, info = eng.full_fit(); print(f"full_fit {1e3*(time.perf_counter()-t0):.1f} ms r2={r2:.6f} info={info}")
rng = np.random.default_rng(0)
perms = np.stack([rng.permutation(p) for _ in range(B)]).astype(np.int32)
eng.run_batch(perms, True)  # warm-up (allocations)
eng.merge(); eng.synchronize()
eng.profile(False)
t0 = time.perf_counter()
for s in range(steps):
    eng.run_batch(perms, True)
    eng.merge()
eng.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"batch of {B} antithetical samples: {1e3*dt:.2f} ms -> {2*B/dt:.0f} orderings/s (no events)")
eng.profile(True)
eng.profile_reset()
for s in range(steps):
    eng.run_batch(perms, True)
    eng.merge()
eng.synchronize()
for k, (ms, cnt) in eng.profile_read().items():
    if cnt: print(f"  {k:11s} {ms/steps:9.3f} ms/batch  ({cnt//steps} launches)")
if len(sys.argv) > 5:
    # in-process A/B of developer flag sets, interleaved rounds, per kernel class
    sets = [int(x) for x in sys.argv[5].split(",")]
    res = {f: [] for f in sets}
    for rnd in range(4):
        for f in sets:
            eng.set_flags(f)
            eng.profile_reset()
            for s in range(3):
                eng.run_batch(perms, True); eng.merge()
            eng.synchronize()
            pr = eng.profile_read()
            res[f].append({k: v[0] / 3 for k, v in pr.items() if v[1]})
    for f in sets:
        keys = res[f][0].keys()
        print("flags", f, {k: round(min(r[k] for r in res[f]), 3) for k in keys})
    # wall-clock A/B without events (the two-slice schedule only runs unprofiled)
    eng.profile(False)
    wall = {f: [] for f in sets}
    for rnd in range(4):
        for f in sets:
            eng.set_flags(f)
            eng.run_batch(perms, True); eng.merge(); eng.synchronize()
            t0 = time.perf_counter()
            for s in range(5):
                eng.run_batch(perms, True); eng.merge()
            eng.synchronize()
            wall[f].append((time.perf_counter() - t0) / 5)
    for f in sets:
        print("flags", f, f"wall {1e3*min(wall[f]):.3f} ms/batch")
    eng.profile(True)
    eng.set_flags(0)
n, mean, cov = eng.stats()
print("n", n, "sum(mean)", mean.sum(), "r2", r2)
