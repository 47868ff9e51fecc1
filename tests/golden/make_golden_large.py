"""Golden fixtures at the BENCHMARK feature counts, from the REAL reference (same rules as make_golden.py: run only
in the build container, arrays only, tests never import the reference).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_large.py

The raw data are not stored: the tests regenerate them from the seed with `large_problem` below (restated in
tests/conftest.py), a plain default_rng stream.  Stored: the orderings, the reference's lift vector for each of them
(`square_shapley`, ls_spa/ls_spa.py:256-287, on the reference's own `reduce_data` output, :290-318), theta and
r_squared of the full fit, and -- at p = 1000 -- the attribution of the reference's driver on those orderings together
with its error estimates (error_history, overall_error, attribution_errors: ls_spa/ls_spa.py:222-236, :321-341 --
statistical pins for the build's estimators, which draw other normals).

    ... make_golden_large.py large_p1000      # regenerate one fixture only

  large_p1000.npz : p = 1000, N = M = 4000, seed 1000, 8 orderings          (~100 KB)
  large_p5000.npz : p = 5000, N = M = 6000, seed 5000, 1 ordering + reverse (~100 KB)
"""
import os
import sys
import time

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import ls_spa as ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def large_problem(seed, p, n, m):
    rng = np.random.default_rng(seed)
    X_tr = rng.standard_normal((n, p))
    X_te = rng.standard_normal((m, p))
    w = rng.standard_normal(p) / np.sqrt(p)
    return X_tr, X_te, X_tr @ w + rng.standard_normal(n), X_te @ w + rng.standard_normal(m)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path)} B")


for name, seed, p, n, m, n_ord, reg in (("large_p1000", 1000, 1000, 4000, 4000, 8, 0.0),
                                        ("large_p5000", 5000, 5000, 6000, 6000, 1, 1e-2)):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    t0 = time.time()
    d = large_problem(seed, p, n, m)
    red = ref.reduce_data(*d, reg)
    yn = np.linalg.norm(d[3]) ** 2
    rng = np.random.default_rng(seed + 1)
    orders = np.array([rng.permutation(p) for _ in range(n_ord)])
    if n_ord == 1:
        orders = np.vstack([orders, orders[:, ::-1]])
    lifts = np.array([ref.square_shapley(*red, yn, o) for o in orders])
    theta = np.linalg.lstsq(red[0], red[2], rcond=None)[0]                      # :240
    r2 = (np.linalg.norm(red[3]) ** 2 - np.linalg.norm(red[3] - red[1] @ theta) ** 2) / yn   # :241-243
    pack = dict(seed=np.int64(seed), p=np.int64(p), N=np.int64(n), M=np.int64(m), reg=np.float64(reg),
                orders=orders.astype(np.int16), lifts=lifts, theta=theta, r_squared=np.float64(r2),
                y_norm_sq=np.float64(yn), y_test_head=d[3][:8], X_train_head=d[0][0, :8])
    if p == 1000:
        r = ref.ls_spa(*d, reg=reg, perms=orders, batch_size=4, tolerance=0.0)   # antithetical (the default)
        pack.update(attribution=r.attribution, drv_theta=r.theta, drv_r_squared=np.float64(r.r_squared),
                    drv_error_history=np.asarray(r.error_history), drv_overall_error=np.float64(r.overall_error),
                    drv_attribution_errors=np.asarray(r.attribution_errors), drv_batch_size=np.int64(4))
    save(name, **pack)
    print(f"  {name}: {time.time() - t0:.1f} s, sum(lift) - r2 = {lifts.sum(axis=1) - r2}")
