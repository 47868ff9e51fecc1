"""Synthetic workloads used by the benchmark and the experiment harness.

``gaussian`` is the primary benchmark data of SURVEY.md section 8d; ``correlated`` restates the
generator of the reference's "Medium" experiment (cvxgrp/ls-spa
``experiments/ground_truth_medium.py:74-106``): a low-rank-plus-identity correlation structure,
10 % relevant features, noise set by a signal-to-noise ratio, everything centred by the training
means.
"""
from __future__ import annotations

import numpy as np


def gaussian(p, n_train, n_test, seed=0):
    rng = np.random.default_rng(seed)
    X_tr = rng.standard_normal((n_train, p))
    X_te = rng.standard_normal((n_test, p))
    theta = rng.standard_normal(p)
    y_tr = X_tr @ theta + rng.standard_normal(n_train)
    y_te = X_te @ theta + rng.standard_normal(n_test)
    return X_tr, X_te, y_tr, y_te


def correlated(rng, p, n_train, n_test, conditioning=20.0, stn_ratio=5.0):
    """Returns (X_train, X_test, y_train, y_test, theta_true, cov)."""
    rank = max(int(p / conditioning), 1)
    A = rng.standard_normal((p, rank))
    cov = A @ A.T + np.eye(p)
    scale = np.sqrt(np.diag(cov))
    cov = cov / np.outer(scale, scale)
    X_tr = rng.multivariate_normal(np.zeros(p), cov, (n_train,), method="svd")
    X_te = rng.multivariate_normal(np.zeros(p), cov, (n_test,), method="svd")
    n_rel = max((p + 1) // 10, 1)
    theta = np.zeros(p)
    theta[:n_rel] = 2.0
    theta = rng.permutation(theta)
    noise = np.sqrt(np.sum(np.diag(cov) * theta ** 2) / stn_ratio)
    y_tr = X_tr @ theta + noise * rng.standard_normal(n_train)
    mean_x = X_tr.mean(axis=0, keepdims=True)
    mean_y = y_tr.mean()
    y_te = X_te @ theta + noise * rng.standard_normal(n_test)
    return X_tr - mean_x, X_te - mean_x, y_tr - mean_y, y_te - mean_y, theta, cov
