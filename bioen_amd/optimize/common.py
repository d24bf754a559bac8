"""Helpers shared by the log-weights and forces front ends
(API of bioen/optimize/common.py)."""
from __future__ import print_function

import numpy as np


def _dense(x):
    return np.asarray(x, dtype=np.float64)


def chiSqrTerm(w, yTilde, YTilde):
    """0.5 * |yTilde w - YTilde|^2   (common.py:5-39).  Host numpy: this is the
    caller-side helper analyze uses for its nuisance refits, not the optimizer path."""
    v = _dense(yTilde).dot(_dense(w).reshape(-1)) - _dense(YTilde).reshape(-1)
    return 0.5 * float(v.dot(v))


def getAve(w, y):
    """Ensemble average y . w as a flat (M,) array (common.py:43-63)."""
    return np.asarray(_dense(y).dot(_dense(w).reshape(-1))).reshape(-1)


def print_highlighted(str, verbose=True):
    if verbose:
        bar = "-" * len(str)
        print(bar)
        print(str)
        print(bar)


def set_caching_heuristics(m, n):
    """True iff an M x N double matrix stays below 8 GiB (common.py:82-106).  Only kept
    so that cfg["cache_ytilde_transposed"] = "auto" resolves as in the reference; the
    device path never builds a transposed copy."""
    return not (m * n * 8 > 8 * 2 ** 30)
