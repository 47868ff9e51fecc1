"""CPU oracle for the LS-SPA hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This module is a NumPy/SciPy restatement of the reference algorithm
(cvxgrp/ls-spa @ v2, ``ls_spa/ls_spa.py``).  It keeps the reference's *cost
model* (one p x p Householder QR + one dense triangular solve + one p x p GEMM
per ordering) so that it can double as the reported CPU baseline
(``bench.py`` -> ``cpu_baseline``, kind "port").

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``ls-spa_amd/`` may.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference (``PYTHONPATH=/root/reference``) in the build container and stores
its outputs as fixtures; ``tests/test_oracle_golden.py`` checks every function
below against those fixtures (and the fixtures never need the reference again).

Each function cites the reference lines it restates (paths relative to the
reference root).
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass

import numpy as np
import scipy.linalg as sla


# --------------------------------------------------------------------------
# a1  one-time reduction                       ls_spa/ls_spa.py:290-318
# --------------------------------------------------------------------------
def reduce(X_tr, X_te, y_tr, y_te, reg):
    """Collapse (N x p, M x p) data to two small factors and two short vectors.

    Returns (R_tr, F_te, q_tr, q_te) with R_tr^T R_tr = X_tr^T X_tr / N + reg I,
    R_tr^T q_tr = X_tr^T y_tr / N, F_te^T F_te = X_te^T X_te,
    F_te^T q_te = X_te^T y_te.  (ls_spa/ls_spa.py:309-317; the ridge rows are
    stacked under the 1/sqrt(N)-scaled training rows, :309-312.)
    """
    n_obs, p = X_tr.shape
    scale = math.sqrt(n_obs)
    stacked = np.concatenate([X_tr / scale, math.sqrt(reg) * np.identity(p)], axis=0)
    target = np.concatenate([y_tr / scale, np.zeros(p)])
    q_basis, R_tr = np.linalg.qr(stacked)          # :314
    t_basis, F_te = np.linalg.qr(X_te)             # :315
    return R_tr, F_te, q_basis.T @ target, t_basis.T @ y_te   # :316-317


# --------------------------------------------------------------------------
# a2  lift vector of one ordering             ls_spa/ls_spa.py:256-287
# --------------------------------------------------------------------------
def ordering_lift(R_tr, F_te, q_tr, q_te, y_norm_sq, order):
    """R^2 increments of the p nested fits along ``order``, stored per feature.

    Steps follow the reference line by line in meaning, not in code:
    QR of the column-permuted training factor (:275); z = Q^T q_tr, prefix-masked
    (:278); all p nested coefficient vectors from one triangular solve (:279-280);
    test costs of the p+1 nested models (:282-283); R^2 and its adjacent
    differences scattered back to feature positions (:284-285).
    """
    order = np.asarray(order, dtype=np.intp)
    p = R_tr.shape[0]
    Q, R = np.linalg.qr(R_tr[:, order])
    z = Q.T @ q_tr
    prefix_rhs = np.triu(np.broadcast_to(z[:, None], (p, p)))
    coef = sla.solve_triangular(R, prefix_rhs)                 # column j: first j+1 features
    fitted = F_te[:, order] @ coef                             # (rows of F_te) x p
    base = float(q_te @ q_te)
    cost = np.empty(p + 1)
    cost[0] = base                                             # empty model
    cost[1:] = np.einsum("ij,ij->j", fitted - q_te[:, None], fitted - q_te[:, None])
    r2 = (base - cost) / y_norm_sq
    lift = np.empty(p)
    lift[order] = np.diff(r2)
    return lift


def sample_lift(R_tr, F_te, q_tr, q_te, y_norm_sq, order, antithetical):
    """One Monte-Carlo sample: the ordering, optionally averaged with its reverse
    (ls_spa/ls_spa.py:203-208)."""
    order = np.asarray(order)
    lift = ordering_lift(R_tr, F_te, q_tr, q_te, y_norm_sq, order)
    if antithetical:
        lift = 0.5 * (lift + ordering_lift(R_tr, F_te, q_tr, q_te, y_norm_sq, order[::-1]))
    return lift


# --------------------------------------------------------------------------
# a4  running statistics                      ls_spa/ls_spa.py:103-119
# --------------------------------------------------------------------------
def pooled_mean(mean_a, mean_b, n_a, n_b):
    """Weighted mean of two batch means (:103-108)."""
    n = n_a + n_b
    return mean_a * (n_a / n) + mean_b * (n_b / n)


def pooled_cov(mean_a, mean_b, cov_a, cov_b, n_a, n_b):
    """Biased covariance of the union of two batches (:111-119)."""
    n = n_a + n_b
    gap = mean_a - mean_b
    return cov_a * (n_a / n) + cov_b * (n_b / n) + (n_a / n) * (n_b / n) * np.outer(gap, gap)


# --------------------------------------------------------------------------
# a5  error estimator                         ls_spa/ls_spa.py:321-341
# --------------------------------------------------------------------------
def error_quantiles(rng, cov, draws=1024, q=0.95):
    """95th percentiles of |N(0,cov)| per feature and of its 2-norm (:332-341).

    Keeps the reference's generator call order: a Cholesky-method draw first and,
    if that raises, a second SVD-method draw (:333-336)."""
    p = cov.shape[0]
    zero = np.zeros(p)
    try:
        x = rng.multivariate_normal(zero, cov, size=draws, method="cholesky")
    except Exception:
        x = rng.multivariate_normal(zero, cov, size=draws, method="svd")
    return np.quantile(np.abs(x), q, axis=0), np.quantile(np.linalg.norm(x, axis=1), q)


# --------------------------------------------------------------------------
# a8  ordering sources
# --------------------------------------------------------------------------
def orderings_argsort(qmc, count):
    """experiments/ground_truth_medium.py:70-71."""
    return np.argsort(qmc.random(count), axis=1)


def permutohedron_basis(p):
    """Row-orthonormal (p-1) x p basis of the hyperplane sum(x)=0
    (experiments/ground_truth_medium.py:62-65): row k is (1,..,1,-(k+1),0,..,0)
    normalised."""
    U = np.tril(np.ones((p - 1, p)))
    U[np.arange(p - 1), np.arange(1, p)] = -np.arange(1, p)
    return U / np.linalg.norm(U, axis=1, keepdims=True)


def orderings_permutohedron(qmc, count, p):
    """experiments/ground_truth_medium.py:56-67."""
    pts = qmc.random(count)
    pts = pts / np.linalg.norm(pts, axis=1, keepdims=True)
    return np.argsort(pts @ permutohedron_basis(p), axis=1)


# --------------------------------------------------------------------------
# a5-a7  the estimator loop                   ls_spa/ls_spa.py:122-253
# --------------------------------------------------------------------------
@dataclass
class OracleResult:
    attribution: np.ndarray
    theta: np.ndarray
    overall_error: float
    attribution_errors: np.ndarray
    r_squared: float
    error_history: np.ndarray
    attribution_history: np.ndarray | None
    n_samples: int = 0
    cov: np.ndarray | None = None


def estimate(X_tr, X_te, y_tr, y_te, reg=0.0, max_samples=2 ** 13, batch_size=2 ** 8,
             tolerance=1e-2, seed=42, perms=None, antithetical=True,
             return_attribution_history=False):
    """Sequential restatement of the reference driver (:158-253), including its
    shared-generator interleave (:168, :175, :224) and its trigger indices
    (:222, :233-236)."""
    X_tr, X_te, y_tr, y_te = (np.array(a) for a in (X_tr, X_te, y_tr, y_te))
    p = X_tr.shape[1]
    rng = np.random.default_rng(seed)
    if perms is None:
        if p < 9:                                             # :170-173
            perms, batch_size, antithetical = itertools.permutations(range(p)), 256, False
        else:                                                 # :175 (lazy on purpose)
            perms = (rng.permutation(p) for _ in range(max_samples))
    else:
        max_samples = 2 ** 100                                # :177
    y_norm_sq = np.linalg.norm(y_te) ** 2                     # :180
    R_tr, F_te, q_tr, q_te = reduce(X_tr, X_te, y_tr, y_te, reg)

    mean = np.zeros(p)
    cov = np.zeros((p, p))
    feat_err, total_err = np.zeros(p), 0.0
    err_hist = []
    hist = [] if return_attribution_history else None
    seen, pending = 0, True
    for seen, order in enumerate(perms, 1):
        pending = True
        lift = sample_lift(R_tr, F_te, q_tr, q_te, y_norm_sq, np.array(order), antithetical)
        cov = pooled_cov(mean, lift, cov, np.zeros((p, p)), seen - 1, 1)    # :212-214
        mean = pooled_mean(mean, lift, seen - 1, 1)                         # :215-216
        if hist is not None:
            hist.append(mean.copy())
        if (seen % batch_size == 0 or seen == max_samples - 1) and p >= 9:  # :222
            feat_err, total_err = error_quantiles(rng, cov * seen / (seen - 1) / seen)
            err_hist.append(total_err)
            pending = False
            if total_err < tolerance:
                break
    if p >= 9 and pending:                                                  # :233-236
        with np.errstate(divide="ignore", invalid="ignore"):
            feat_err, total_err = error_quantiles(rng, cov * seen / (seen - 1) / seen)
        err_hist.append(total_err)

    theta = np.linalg.lstsq(R_tr, q_tr, rcond=None)[0]                      # :240
    r2 = (np.linalg.norm(q_te) ** 2 - np.linalg.norm(q_te - F_te @ theta) ** 2) / y_norm_sq
    return OracleResult(mean, theta, total_err, feat_err, r2, np.array(err_hist),
                        None if hist is None else np.array(hist).reshape(-1, p),
                        n_samples=seen, cov=cov)


# --------------------------------------------------------------------------
# independent definitions (second / third oracle)
# --------------------------------------------------------------------------
def refit_lift(X_tr, X_te, y_tr, y_te, order):
    """Lift vector by refitting every prefix on the raw data
    (notebooks/medium_experiment.py:263-275).  O(p) least-squares solves."""
    order = np.asarray(order)
    tss = float(y_te @ y_te)
    lift = np.zeros(len(order))
    prev = 0.0
    for j in range(1, len(order) + 1):
        cols = order[:j]
        beta = np.linalg.lstsq(X_tr[:, cols], y_tr, rcond=None)[0]
        r2 = (tss - float(np.sum((X_te[:, cols] @ beta - y_te) ** 2))) / tss
        lift[order[j - 1]] = r2 - prev
        prev = r2
    return lift


def brute_force_shapley(X_tr, X_te, y_tr, y_te):
    """Exact Shapley values from the 2^p subset R^2 table
    (notebooks/shapley_toy.py:100-140, without its 2-decimal rounding)."""
    p = X_tr.shape[1]
    tss = float(y_te @ y_te)
    r2 = {}
    for mask in range(1 << p):
        cols = [i for i in range(p) if mask >> i & 1]
        if not cols:
            r2[mask] = 0.0
            continue
        beta = np.linalg.lstsq(X_tr[:, cols], y_tr, rcond=None)[0]
        r2[mask] = (tss - float(np.sum((X_te[:, cols] @ beta - y_te) ** 2))) / tss
    phi = np.zeros(p)
    for order in itertools.permutations(range(p)):
        mask = 0
        for f in order:
            phi[f] += r2[mask | 1 << f] - r2[mask]
            mask |= 1 << f
    return phi / math.factorial(p)


# --------------------------------------------------------------------------
# synthetic workloads
# --------------------------------------------------------------------------
def gaussian_workload(p, n_train, n_test, seed=0, dtype=np.float64):
    """Primary benchmark data: iid N(0,1) features, y = X theta + N(0,1)."""
    rng = np.random.default_rng(seed)
    X_tr = rng.standard_normal((n_train, p))
    X_te = rng.standard_normal((n_test, p))
    theta = rng.standard_normal(p)
    y_tr = X_tr @ theta + rng.standard_normal(n_train)
    y_te = X_te @ theta + rng.standard_normal(n_test)
    return tuple(a.astype(dtype) for a in (X_tr, X_te, y_tr, y_te))


def correlated_workload(rng, p, n_train, n_test, conditioning=20.0, stn_ratio=5.0):
    """The reference's harder generator (experiments/ground_truth_medium.py:74-106):
    low-rank-plus-identity correlation, 10 % relevant features, centred by the
    training means."""
    A = rng.standard_normal((p, int(p / conditioning)))
    cov = A @ A.T + np.eye(p)
    d = np.sqrt(np.diag(cov))
    cov = cov / np.outer(d, d)
    X_tr = rng.multivariate_normal(np.zeros(p), cov, (n_train,), method="svd")
    X_te = rng.multivariate_normal(np.zeros(p), cov, (n_test,), method="svd")
    k = max((p + 1) // 10, 1)
    coef = np.zeros(p)
    coef[:k] = 2.0
    coef = rng.permutation(coef)
    noise = np.sqrt(np.sum(np.diag(cov) * coef ** 2) / stn_ratio)
    y_tr = X_tr @ coef + noise * rng.standard_normal(n_train)
    mu_x, mu_y = X_tr.mean(axis=0, keepdims=True), None
    X_tr = X_tr - mu_x
    mu_y = y_tr.mean()
    y_tr = y_tr - mu_y
    y_te = X_te @ coef + noise * rng.standard_normal(n_test)
    return X_tr, X_te - mu_x, y_tr, y_te - mu_y
