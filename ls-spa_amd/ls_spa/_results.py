"""Result container and input validation of the LS-SPA surface.

Mirrors cvxgrp/ls-spa ``ls_spa/ls_spa.py``: ``ShapleyResults`` (:34-70, same field
order, same dashboard text), ``SizeIncompatible`` (:73-78) and ``validate_data``
(:81-100, same four checks in the same order, same messages).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


def _head(values, limit=5):
    flat = np.asarray(values).ravel()
    shown = ", ".join(f"{v:.2f}" for v in flat[:limit])
    return f"({shown}, ...)" if flat.size > limit else f"({shown})"


@dataclass
class ShapleyResults:
    attribution: np.ndarray
    theta: np.ndarray
    overall_error: float
    attribution_errors: np.ndarray
    r_squared: float
    error_history: np.ndarray | None
    attribution_history: np.ndarray | None

    def __repr__(self):
        pad = " " * 8
        lines = [
            "",
            f"{pad}p = {np.asarray(self.attribution).size}",
            f"{pad}Out-of-sample R^2 with all features: {self.r_squared:.2f}",
            "",
            f"{pad}Shapley attribution: {_head(self.attribution)}",
            f"{pad}Estimated error in Shapley attribution: {self.overall_error:.2E}",
            "",
            f"{pad}Fitted coeficients with all features: {_head(self.theta)}",
            pad,
        ]
        return "\n".join(lines)


class SizeIncompatible(Exception):
    """Raised when the shapes of the four data arrays do not fit together."""

    def __init__(self, message):
        self.message = message
        super().__init__(self.message)


_CHECKS = (
    (lambda Xa, Xe, ya, ye: Xa.shape[1] != Xe.shape[1],
     "X_train and X_test should have the same number of columns (features)."),
    (lambda Xa, Xe, ya, ye: Xa.shape[0] != ya.shape[0],
     "X_train should have the same number of rows as y_train has entries (observations)."),
    (lambda Xa, Xe, ya, ye: Xe.shape[0] != ye.shape[0],
     "X_test should have the same number of rows as y_test has entries (observations)."),
    (lambda Xa, Xe, ya, ye: Xa.shape[1] > Xa.shape[0],
     "The function works only if the number of features is at most the number of observations."),
)


def validate_data(X_train, X_test, y_train, y_test):
    for broken, message in _CHECKS:
        if broken(X_train, X_test, y_train, y_test):
            raise SizeIncompatible(message)
