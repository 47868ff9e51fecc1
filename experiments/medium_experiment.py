#!/usr/bin/env python3
"""The reference's "Medium" experiment on the HIP engine, as a plain script (no marimo):

  1. data from the reference's generator (p = 100, N = M = 100000, conditioning 20, STN 5, seed 42;
     cvxgrp/ls-spa experiments/ground_truth_medium.py:15-21, :74-106);
  2. ground-truth attribution from 2^19 random orderings, antithetical, tolerance 0 (:113-116);
  3. convergence of Monte-Carlo, argsort-QMC and permutohedron-QMC sampling, each with and
     without antithetical pairing (notebooks/medium_experiment.py:348-568), as the L2 error of the
     running attribution against the ground truth (:597-603).

Writes experiments/out/gt_Medium.npy and experiments/out/convergence.csv.  Needs an MI355X.

    python experiments/medium_experiment.py [--gt-log2 19] [--samples 8192] [--rows 100000]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ls-spa_amd"))
from ls_spa import ls_spa            # noqa: E402
from ls_spa.workloads import correlated  # noqa: E402


def run(p=100, rows=100000, gt_log2=19, samples=2 ** 13, out_dir=None, log=print):
    """The whole experiment; returns (ground-truth ShapleyResults, convergence rows, per-run results)."""
    out_dir = out_dir or os.path.join(ROOT, "experiments", "out")
    os.makedirs(out_dir, exist_ok=True)

    rng = np.random.default_rng(42)
    Xa, Xe, ya, ye, _, _ = correlated(rng, p, rows, rows)

    t0 = time.perf_counter()
    n_gt = 2 ** gt_log2
    gt = ls_spa(Xa, Xe, ya, ye, perms=(rng.permutation(p) for _ in range(n_gt)), tolerance=0.0,
                batch_size=2 ** 12)
    t_gt = time.perf_counter() - t0
    np.save(os.path.join(out_dir, "gt_Medium.npy"), gt.attribution)
    log(f"ground truth: {n_gt} antithetical samples ({2 * n_gt} orderings) in {t_gt:.2f} s "
        f"-> {2 * n_gt / t_gt:.0f} orderings/s; sum = {gt.attribution.sum():.6f}, R^2 = {gt.r_squared:.6f}")

    table, runs = [], {}
    for method in ("random", "argsort", "permutohedron"):
        for anti in (False, True):
            t0 = time.perf_counter()
            r = ls_spa(Xa, Xe, ya, ye, method=method, antithetical=anti, max_samples=samples,
                       batch_size=2 ** 8, tolerance=1e-8, seed=42, return_history=True)
            dt = time.perf_counter() - t0
            runs[(method, anti)] = r
            err = np.linalg.norm(r.attribution_history - gt.attribution, axis=1)
            for n in (2 ** k for k in range(4, int(np.log2(len(err))) + 1)):
                table.append((method, int(anti), n, err[n - 1]))
            log(f"{method:14s} antithetical={int(anti)}: {len(err)} samples in {dt:.2f} s, "
                f"final L2 error {err[-1]:.3e}")
    with open(os.path.join(out_dir, "convergence.csv"), "w") as fh:
        fh.write("method,antithetical,samples,l2_error\n")
        for r in table:
            fh.write(f"{r[0]},{r[1]},{r[2]},{r[3]:.6e}\n")
    log("wrote " + os.path.join(out_dir, "convergence.csv"))
    return gt, table, runs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--p", type=int, default=100)
    ap.add_argument("--rows", type=int, default=100000)
    ap.add_argument("--gt-log2", type=int, default=19)
    ap.add_argument("--samples", type=int, default=2 ** 13)
    ap.add_argument("--out-dir", default=None)
    args = ap.parse_args()
    run(args.p, args.rows, args.gt_log2, args.samples, args.out_dir)


if __name__ == "__main__":
    main()
