"""Where a C2 group's period goes on the host: time inside launch_batch / collect_chunks against the whole loop."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ls-spa_amd"))
from ls_spa._engine import HipEngine

p, n, d, b = 100, 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 8, 128
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rng = np.random.default_rng(0)
Xa, Xe = rng.standard_normal((n, p)), rng.standard_normal((n, p))
th = rng.standard_normal(p)
ya, ye = Xa @ th + rng.standard_normal(n), Xe @ th + rng.standard_normal(n)
eng = HipEngine(0)
eng.load_data(Xa, Xe, ya, ye, 1e-3)
eng.full_fit()
eng.set_lanes(lanes)
G = 80
perms = np.array([rng.permutation(p) for _ in range(G * d * b)], dtype=np.int32).reshape(G, d * b, p)
for rep in range(3):
    eng.reset_stats()
    eng.synchronize()
    tl = tc = 0.0
    t0 = time.perf_counter()
    tk = {}
    for g in range(G):
        a = time.perf_counter()
        if g not in tk:
            tk[g] = eng.launch_batch(perms[g], True)
        if lanes == 2 and g + 1 < G:
            tk[g + 1] = eng.launch_batch(perms[g + 1], True)
        c = time.perf_counter()
        eng.collect_chunks(tk.pop(g), 0, b, d, accumulate=2)
        e = time.perf_counter()
        tl += c - a
        tc += e - c
    t1 = time.perf_counter()
    eng.synchronize()
    t2 = time.perf_counter()
    print(f"d={d} lanes={lanes} per group: launch {tl / G * 1e6:.1f} us collect {tc / G * 1e6:.1f} us host loop {(t1 - t0) / G * 1e6:.1f} us "
          f"wall {(t2 - t0) / G * 1e6:.1f} us  -> {2 * d * b * G / (t2 - t0) / 1e6:.2f} M orderings/s")
eng.close()
