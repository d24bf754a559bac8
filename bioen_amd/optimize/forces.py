"""Forces method: public API of ``bioen/optimize/forces.py`` on MI355X.

M generalised forces f parametrise the weights, w = w0 * exp(f^T yTilde) / Z:

  L(f)      = theta * sum_j w_j log(w_j / w0_j) + 0.5 |yTilde w - YTilde|^2
  dL/df_i   = sum_j (yTilde_ij - ybar_i) w_j [ theta (1 + log(w_j/w0_j)) + (yTilde^T r)_j ]

`use_c=True` paths evaluate on the device (ext.c_bioen); the ``*_base`` functions are
the reference's explicit numpy variants.
"""
from __future__ import print_function

import time

import numpy as np

from . import common
from .ext import c_bioen
from .log_weights import _run_scipy


# ------------------------------------------------------------------ synthetic data
def gen_synthetic_data(M, N, YTrue, sig_exp):
    """Observed values ~ N(YTrue, sig_exp) and their scaled form (forces.py:19-41)."""
    YObs = np.array(np.random.normal(YTrue, sig_exp))
    return YObs, YObs / sig_exp


def gen_sythetic_ensemble(M, N, YTrue, sig_exp, sig_sim):
    """Ensemble observables ~ N(YTrue, sig_sim) per structure, and yTilde = y / sig_exp
    (forces.py:44-68).  Returns (y, yTilde), both (M, N)."""
    YTrue = np.asarray(YTrue, dtype=np.float64)
    y = np.array(np.random.normal(np.repeat(YTrue[:, None], N, axis=1), sig_sim))
    yTilde = y / np.asarray(sig_exp, dtype=np.float64).reshape(-1, 1)
    return y, yTilde


def init_forces(M, val=0):
    forces = np.zeros((M, 1))
    forces[:, 0] = val
    return forces


# ------------------------------------------------------------------ numpy helpers
def get_weights_from_forces(w0, y, forces):
    """w (n,1) from forces (1,m) or (m,) (forces.py:91-111); host numpy."""
    f = np.asarray(forces, dtype=np.float64)
    if f.ndim == 1:
        f = f[None, :]
    x = f.dot(np.asarray(y, dtype=np.float64))
    e = np.exp(x - np.max(x))
    w = np.asarray(w0, dtype=np.float64).reshape(-1, 1) * e.T
    return w / w.sum()


def _kl(w, w0):
    w = np.asarray(w, dtype=np.float64).reshape(-1)
    w0 = np.asarray(w0, dtype=np.float64).reshape(-1)
    ind = w > 0                      # lim w->0 of w log w is 0
    return float(np.log(w[ind] / w0[ind]).dot(w[ind]))


def bioen_chi2_s_forces(forces, w0, yTilde, YTilde):
    """(S, chiSqr) at `forces` (m,1)/(m,): relative entropy sum w log(w/w0) and
    0.5 chi^2 (forces.py:116-136)."""
    w = get_weights_from_forces(w0, yTilde, np.asarray(forces).T)
    return _kl(w, w0), common.chiSqrTerm(w, yTilde, YTilde)


def check_params_forces(forcesInit, w0, y, yTilde, YTilde):
    """Shape contract: forcesInit (m,1); w0 (n,1); y,yTilde (m,n); YTilde (1,m) (forces.py:139-182)."""
    m, n = yTilde.shape
    expected = (("forcesInit", forcesInit, (m, 1)), ("w0", w0, (n, 1)), ("y", y, (m, n)),
                ("YTilde", YTilde, (1, m)))
    bad = False
    for name, arr, shape in expected:
        if arr.shape != shape:
            print("Unexpected shape for variable: {name}\nExpected: {expected}\nCurrent:  {current}".format(
                name=name, expected=shape, current=arr.shape))
            bad = True
    if bad:
        raise ValueError("arguments dimensionality for the 'forces' method are wrong")


# ------------------------------------------------------------------ objective / gradient
def bioen_log_posterior(forces, w0, y, yTilde, YTilde, theta, use_c=True, caching=False):
    """Negative log-posterior in the forces parametrisation (forces.py:187-212); `y` unused."""
    if use_c:
        return c_bioen.bioen_log_posterior_forces(forces, w0, yTilde, YTilde, theta)
    return bioen_log_posterior_base(forces, w0, yTilde, YTilde, theta)


def grad_bioen_log_posterior(forces, w0, y, yTilde, YTilde, theta, use_c=True, caching=False):
    """Gradient w.r.t. the m forces (forces.py:216-244); `y` unused."""
    if use_c:
        return c_bioen.grad_bioen_log_posterior_forces(forces, w0, yTilde, YTilde, theta)
    return grad_bioen_log_posterior_base(forces, w0, yTilde, YTilde, theta)


def bioen_log_posterior_base(forces, w0, yTilde, YTilde, theta, use_c=True):
    """Pure-numpy objective (forces.py:248-289)."""
    w = get_weights_from_forces(w0, yTilde, np.asarray(forces).T)
    return theta * _kl(w, w0) + common.chiSqrTerm(w, yTilde, YTilde)


def grad_bioen_log_posterior_base(forces, w0, yTilde, YTilde, theta, use_c=True):
    """Pure-numpy gradient (forces.py:293-333)."""
    yT = np.asarray(yTilde, dtype=np.float64)
    w = get_weights_from_forces(w0, yT, np.asarray(forces).T)[:, 0]
    w0v = np.asarray(w0, dtype=np.float64).reshape(-1)
    ybar = yT.dot(w)
    b = yT.T.dot(ybar - np.asarray(YTilde, dtype=np.float64).reshape(-1))
    ratio = np.where(w > 0, w / w0v, 1.0)
    t = ((np.log(ratio) + 1.0) * theta + b) * w
    return (yT - ybar[:, None]).dot(t)        # centred, as forces.py:329-332 (no cancellation)


# ------------------------------------------------------------------ optimizer
class _DeviceFdf(object):
    """One fused device evaluation per point for scipy's separate f / f' callbacks."""

    def __init__(self, w0, yTilde, YTilde, theta):
        self.ctx, self._cached = c_bioen._context_for(yTilde, YTilde)
        self.w0 = np.asarray(w0, dtype=np.float64).reshape(-1)
        self.theta = theta
        self._x = None

    def _eval(self, x):
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        if self._x is None or not np.array_equal(x, self._x):
            self._f, self._g = self.ctx.forces_fdf(x, self.w0, self.theta)
            self._x = x.copy()

    def f(self, x, *unused):
        self._eval(x)
        return self._f

    def fprime(self, x, *unused):
        self._eval(x)
        return self._g

    def close(self):
        c_bioen._release(self.ctx, self._cached)


def find_optimum(forcesInit, w0, y, yTilde, YTilde, theta, cfg):
    """Minimise over the m generalised forces (forces.py:336-548): see `_find_optimum`.  The matrix is HELD for the
    duration of the call (ext/c_bioen.py: hold): the initial objective, the optimisation, chi^2 / S and the averages of the
    optimum are served by one device copy -- at most ONE upload per call whatever the size of the matrix."""
    check_params_forces(forcesInit, w0, y, yTilde, YTilde)
    if str(cfg["minimizer"]).upper() not in ('LIBLBFGS', 'LBFGS', 'GSL', 'SCIPY'):      # rejected before anything touches the device
        raise RuntimeError("Library " + cfg["minimizer"] +
                           " not recognized (valid values =  'LIBLBFGS', 'GSL', 'scipy', 'scipy' ) ")
    if not (str(cfg["minimizer"]).upper() == 'SCIPY' and not bool(cfg["use_c_functions"])):
        with c_bioen.hold(yTilde, YTilde):
            return _find_optimum(forcesInit, w0, y, yTilde, YTilde, theta, cfg)
    return _find_optimum(forcesInit, w0, y, yTilde, YTilde, theta, cfg)


def _find_optimum(forcesInit, w0, y, yTilde, YTilde, theta, cfg):
    """Minimise over the m generalised forces (forces.py:336-548).

    Returns (wopt (n,1), yopt (m,), forces_opt (m,), fmin_initial, fmin_final,
    chiSqr (= 0.5 chi^2), S (= sum w log(w/w0)))."""

    caching = cfg["cache_ytilde_transposed"]
    if caching == "auto":
        caching = common.set_caching_heuristics(yTilde.shape[0], yTilde.shape[1])
    cfg["cache_ytilde_transposed"] = caching

    minimizer = cfg["minimizer"].upper()
    if minimizer not in ('LIBLBFGS', 'LBFGS', 'GSL', 'SCIPY'):
        raise RuntimeError("Library " + cfg["minimizer"] +
                           " not recognized (valid values =  'LIBLBFGS', 'GSL', 'scipy', 'scipy' ) ")
    use_c = bool(cfg["use_c_functions"])
    use_device = not (minimizer == 'SCIPY' and not use_c)

    if use_device:
        fmin_initial = c_bioen.bioen_log_posterior_forces(forcesInit, w0, yTilde, YTilde, theta)
    else:
        fmin_initial = bioen_log_posterior_base(forcesInit, w0, yTilde, YTilde, theta)
    if cfg["verbose"]:
        print("fmin_initial", fmin_initial)

    forces = np.asarray(forcesInit, dtype=np.float64).copy().T    # (1, m)

    start = time.time()
    if minimizer in ('LIBLBFGS', 'LBFGS'):
        common.print_highlighted("FORCES -- Library L-BFGS/HIP", cfg["verbose"])
        res = c_bioen.bioen_opt_lbfgs_forces(forces, w0, yTilde, YTilde, theta, cfg)
    elif minimizer == 'GSL':
        common.print_highlighted("FORCES -- Library GSL/C", cfg["verbose"])
        res = c_bioen.bioen_opt_bfgs_forces(forces, w0, yTilde, YTilde, theta, cfg)
    elif minimizer == 'SCIPY' and use_c:
        common.print_highlighted("FORCES -- Library scipy/HIP", cfg["verbose"])
        dev = _DeviceFdf(w0, yTilde, YTilde, theta)
        try:
            res = _run_scipy(cfg, dev.f, dev.fprime, forces, (), "c", True)
        finally:
            dev.close()
    else:
        common.print_highlighted("FORCES -- Library scipy/PY", cfg["verbose"])
        res = _run_scipy(cfg, bioen_log_posterior_base, grad_bioen_log_posterior_base, forces,
                         (w0, yTilde, YTilde, theta), "py", False)
    end = time.time()
    if cfg["verbose"]:
        print('time elapsed ', (end - start))

    # the reference's c_bioen / scipy drivers hand back a 1-D (m,) vector whose `.T` is itself (forces.py:532-535),
    # and bioen/analyze/procedure.py:77 relies on it: np.matrix(out_min[2]).T must be the (m, 1) start of the next theta
    forces_opt = np.asarray(res[0], dtype=np.float64).reshape(-1)
    fmin_final = res[1]

    if use_device:
        w, chiSqr = c_bioen.chi2_and_kl_forces(forces_opt, w0, yTilde, YTilde)
        wopt = w.reshape(-1, 1)
        S = _kl(wopt, w0)
        yopt = c_bioen.get_ave(wopt, yTilde, YTilde) if y is yTilde else common.getAve(wopt, y)
    else:
        wopt = get_weights_from_forces(w0, yTilde, forces_opt)
        yopt = common.getAve(wopt, y)
        S, chiSqr = bioen_chi2_s_forces(forces_opt, w0, yTilde, YTilde)

    if cfg["verbose"]:
        print("========================")
        print("fmin_initial           =", fmin_initial)
        print("fmin_final             =", fmin_final)
        print("========================")
    return wopt, yopt, forces_opt, fmin_initial, fmin_final, chiSqr, S
