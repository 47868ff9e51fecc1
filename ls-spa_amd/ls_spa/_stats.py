"""Running statistics and the error estimator (host side).

``merge_sample_mean`` / ``merge_sample_cov`` keep the reference's signatures
(cvxgrp/ls-spa ``ls_spa/ls_spa.py:103-119``); the device engine applies the same
pairwise update per batch (csrc/k_lift.hip, stats_merge).  ``error_estimates``
(:321-341) stays on the host on purpose: it shares one ``numpy.random.Generator``
with the ordering sampler, and the order of draws is part of the reference's
observable behaviour.
"""
from __future__ import annotations

import contextlib
import os

import numpy as np

try:   # optional: present in this image; without it the BLAS default stands
    from threadpoolctl import ThreadpoolController
except Exception:   # pragma: no cover
    ThreadpoolController = None

_controller = None
# OpenBLAS starts one thread per core; on a 256-core GPU host that makes the p x p SVD of the reference's estimator
# 4.5x slower (684 ms against 151 ms at p = 1000) and the thin estimator 7x slower than with 16 threads
# (tools/host_estimator_probe.py)
HOST_BLAS_THREADS = int(os.environ.get("LSSPA_HOST_BLAS_THREADS", min(16, os.cpu_count() or 1)))


@contextlib.contextmanager
def host_blas_threads():
    """Run the enclosed NumPy / LAPACK calls with a sane BLAS thread count."""
    global _controller
    if ThreadpoolController is None or HOST_BLAS_THREADS <= 0:
        yield
        return
    if _controller is None:
        _controller = ThreadpoolController()
    with _controller.limit(limits=HOST_BLAS_THREADS):
        yield


def merge_sample_mean(old_mean, new_mean, old_N, new_N):
    total = old_N + new_N
    return (old_N / total) * old_mean + (new_N / total) * new_mean


def merge_sample_cov(old_mean, new_mean, old_cov, new_cov, old_N, new_N):
    total = old_N + new_N
    w_old, w_new = old_N / total, new_N / total
    shift = old_mean - new_mean
    return w_old * old_cov + w_new * new_cov + (w_old * w_new) * np.outer(shift, shift)


def error_estimates(rng, cov):
    """(per-feature, overall) 0.95-quantiles of |x| and ||x||_2 over 1024 draws x ~ N(0, cov).

    Draw order as in the reference: the Cholesky-method sampler first; if it raises
    (singular covariance), the SVD-method sampler draws again."""
    dim = cov.shape[0]
    origin = np.zeros(dim)
    with host_blas_threads():
        try:
            draws = rng.multivariate_normal(origin, cov, size=2 ** 10, method="cholesky")
        except Exception:
            draws = rng.multivariate_normal(origin, cov, size=2 ** 10, method="svd")
        per_feature = np.quantile(np.abs(draws), 0.95, axis=0)
        overall = np.quantile(np.linalg.norm(draws, axis=1), 0.95)
    return per_feature, overall


def error_estimates_lowrank(rng, centered_lifts, n_total=None):
    """Statistically equivalent estimator that never factorises the p x p covariance.

    ``centered_lifts`` is the (n, p) matrix of lift vectors minus their mean.  With
    xi ~ N(0, I_n), x = centered_lifts^T xi / sqrt(n (n-1)) has covariance
    C_unbiased / n, the matrix the reference samples from (ls_spa/ls_spa.py:223-224).
    Costs O(1024 n p) instead of an O(p^3) SVD.  The draws differ from the reference's
    stream, so this is opt-in (``error_estimator='lowrank'``)."""
    n = centered_lifts.shape[0] if n_total is None else n_total
    xi = rng.standard_normal((2 ** 10, centered_lifts.shape[0]))
    with host_blas_threads(), np.errstate(divide="ignore", invalid="ignore"):
        draws = (xi @ centered_lifts) / np.sqrt(n * (n - 1.0))
        return np.quantile(np.abs(draws), 0.95, axis=0), np.quantile(np.linalg.norm(draws, axis=1), 0.95)
