"""MI355X-native LS-SPA: a drop-in for the ``ls_spa`` package of cvxgrp/ls-spa.

Same public names as the reference's ``from .ls_spa import *`` (ls_spa/__init__.py:1):
``ls_spa``, ``ShapleyResults``, ``SizeIncompatible``, ``validate_data``,
``merge_sample_mean``, ``merge_sample_cov``, ``square_shapley``, ``reduce_data``,
``error_estimates``.  Every ordering is evaluated by hand-written HIP kernels for gfx950
behind a C ABI (include/lsspa.h); there is no CPU fallback.
"""
from ._results import ShapleyResults, SizeIncompatible, validate_data
from ._stats import error_estimates, error_estimates_lowrank, merge_sample_cov, merge_sample_mean
from ._driver import ls_spa, reduce_data, square_shapley, run_estimator, release
from ._native import LSSPANativeError
from ._rccl import NativeComm

__all__ = [
    "ls_spa", "ShapleyResults", "SizeIncompatible", "validate_data", "merge_sample_mean",
    "merge_sample_cov", "square_shapley", "reduce_data", "error_estimates",
    "error_estimates_lowrank", "run_estimator", "release", "LSSPANativeError", "NativeComm",
]
