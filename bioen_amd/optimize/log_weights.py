"""Log-weights method: public API of ``bioen/optimize/log_weights.py`` on MI355X.

Signatures, argument shapes, return tuples and exception types follow the
reference (file:line cited per function); the numerics of every `use_c=True`
path run in the HIP kernels behind ``ext.c_bioen``.  The ``*_base`` functions are
the reference's explicit pure-numpy variants (``use_c=False`` /
``use_c_functions: false``), kept because callers can ask for them by name.

  L(g)      = theta * ( sum_j w_j (g_j - G_j) - log sum e^g + log sum e^G ) + 0.5 |yTilde w - YTilde|^2
  dL/dg_k   = theta w_k [(g_k - G_k) - sum_j w_j (g_j - G_j)] + w_k [(yTilde^T r)_k - ybar . r]
"""
from __future__ import print_function

import time

import numpy as np
import scipy.optimize as sopt

from . import common
from .ext import c_bioen


def _col(x):
    """(n,1) float64 ndarray view of a vector-like (np.matrix, (n,), (n,1), (1,n))."""
    return np.asarray(x, dtype=np.float64).reshape(-1, 1)


# ------------------------------------------------------------------ legacy helpers
def getWeights(g):
    """w = exp(g) / s, s = sum exp(g)  (log_weights.py:93-110). Returns (w, s)."""
    ga = np.asarray(g, dtype=np.float64)
    shift = ga.max()
    e = np.exp(ga - shift)
    se = e.sum()
    return np.array(e / se), float(se * np.exp(shift))


def getGs(w):
    """log-weights relative to the last structure (log_weights.py:113-127); w is (n,1)."""
    g = np.log(w)
    g -= g[-1, 0]
    return g


def init_log_weights(w0):
    """(gPrime, g, G, GInit) from reference weights (log_weights.py:71-90)."""
    G = getGs(w0)
    GInit = getGs(np.array(w0))
    g = GInit.copy()
    gPrime = np.asarray(g[:-1].T)[0]
    return gPrime, g, G, GInit


def getWOpt(G, gPrimeOpt):
    """Optimal weights as an (n,1) array from optimal log-weights (log_weights.py:130-160)."""
    w, _ = getWeights(_col(gPrimeOpt))
    return w


def bioen_log_prior(w, s, g, G, theta):
    """theta * ( g.w - G.w - log s + log s0 )  (log_weights.py:18-68)."""
    _, s0 = getWeights(G)
    w, g, G = _col(w), _col(g), _col(G)
    return float(theta * ((g.T.dot(w)).item() - (G.T.dot(w)).item() - np.log(s) + np.log(s0)))


def grad_chiSqrTerm(gPrime, g, G, yTilde, YTilde, theta):
    """Gradient of the chi^2 term w.r.t. the first n-1 log-weights, the last one pinned
    to zero (legacy parametrisation, log_weights.py:164-188)."""
    np.asarray(g)[:-1, 0] = np.asarray(gPrime).reshape(-1)
    np.asarray(g)[-1, 0] = 0
    w, _ = getWeights(g)
    w = _col(w)
    yT = np.asarray(yTilde, dtype=np.float64)
    ybar = yT.dot(w)
    r = ybar - _col(YTilde)
    tmp = w[:, 0] * (yT.T.dot(r)[:, 0] - ybar.T.dot(r).item())
    return tmp[:-1]


def check_params_logweights(GInit, G, y, yTilde, YTilde):
    """Shape contract: GInit,G (n,1); y,yTilde (m,n); YTilde (1,m) (log_weights.py:191-233)."""
    m, n = yTilde.shape
    expected = (("GInit", GInit, (n, 1)), ("G", G, (n, 1)), ("y", y, (m, n)), ("YTilde", YTilde, (1, m)))
    bad = False
    for name, arr, shape in expected:
        if arr.shape != shape:
            print("Unexpected shape for variable: {name}\nExpected: {expected}\nCurrent:  {current}".format(
                name=name, expected=shape, current=arr.shape))
            bad = True
    if bad:
        raise ValueError("arguments dimensionality for the 'log_weights' method are wrong")


# ------------------------------------------------------------------ objective / gradient
def bioen_log_posterior(gPrime, g, G, yTilde, YTilde, theta, use_c=True, caching=False):
    """Negative log-posterior (log_weights.py:237-260). use_c=True -> HIP kernels."""
    if use_c:
        return c_bioen.bioen_log_posterior_logw(gPrime, g, G, yTilde, YTilde, theta, caching=caching)
    return bioen_log_posterior_base(gPrime, g, G, yTilde, YTilde, theta)


def grad_bioen_log_posterior(gPrime, g, G, yTilde, YTilde, theta, use_c=True, caching=False):
    """Gradient w.r.t. the n log-weights (log_weights.py:263-286). use_c=True -> HIP kernels."""
    if use_c:
        return c_bioen.grad_bioen_log_posterior_logw(gPrime, g, G, yTilde, YTilde, theta, caching=caching)
    return grad_bioen_log_posterior_base(gPrime, g, G, yTilde, YTilde, theta)


def bioen_log_posterior_base(gPrime, g, G, yTilde, YTilde, theta):
    """Pure-numpy objective (log_weights.py:289-328).  Like the reference it writes the
    current point into `g` in place."""
    np.asarray(g)[:, 0] = np.asarray(gPrime, dtype=np.float64).reshape(-1)   # view: works for np.matrix too
    w, s = getWeights(g)
    return bioen_log_prior(w, s, g, G, theta) + common.chiSqrTerm(w, yTilde, YTilde)


def grad_bioen_log_posterior_base(gPrime, g, G, yTilde, YTilde, theta):
    """Pure-numpy gradient (log_weights.py:332-406) in closed form.

    The reference's Python version has a sign slip in the <G> term
    (`op2 = G + G.T w`, log_weights.py:370,389); its C kernel
    (c_bioen_kernels_logw.c:216) and the mathematics have `- G + <G>`, which is what
    this function implements.  The two agree whenever G == 0 or theta == 0 -- the only
    cases the reference's tests exercise."""
    gp = _col(gPrime)
    Gc = _col(G)
    w, _ = getWeights(gp)
    yT = np.asarray(yTilde, dtype=np.float64)
    ybar = yT.dot(w)
    r = ybar - _col(YTilde)
    t = w * (yT.T.dot(r) - ybar.T.dot(r).item())
    dev = gp - Gc
    grad = theta * w * (dev - w.T.dot(dev).item()) + t
    return grad[:, 0]


# ------------------------------------------------------------------ optimizer
class _DeviceFdf(object):
    """scipy asks for f(x) and f'(x) in separate calls at the same x; one fused device
    evaluation (two matrix passes) serves both."""

    def __init__(self, G, yTilde, YTilde, theta):
        self.ctx, self._cached = c_bioen._context_for(yTilde, YTilde)
        self.G = np.asarray(G, dtype=np.float64).reshape(-1)
        self.theta = theta
        self._x = None
        self._f = None
        self._g = None

    def _eval(self, x):
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        if self._x is None or not np.array_equal(x, self._x):
            self._f, self._g = self.ctx.logw_fdf(x, self.G, self.theta)
            self._x = x.copy()

    def f(self, x, *unused):
        self._eval(x)
        return self._f

    def fprime(self, x, *unused):
        self._eval(x)
        return self._g

    def close(self):
        c_bioen._release(self.ctx, self._cached)


_SCIPY_ALGORITHMS = {
    # name -> (scipy driver, name of its gradient tolerance, label)
    "lbfgs": (sopt.fmin_l_bfgs_b, "pgtol", "L-BFGS"),
    "fmin_l_bfgs_b": (sopt.fmin_l_bfgs_b, "pgtol", "L-BFGS"),
    "bfgs": (sopt.fmin_bfgs, "gtol", "BFGS"),
    "fmin_bfgs": (sopt.fmin_bfgs, "gtol", "BFGS"),
    "cg": (sopt.fmin_cg, "gtol", "CG"),
    "fmin_cg": (sopt.fmin_cg, "gtol", "CG"),
}


def _run_scipy(cfg, f, fprime, x0, args, flavour, show_caching):
    """The three scipy drivers of log_weights.py:464-598 / forces.py:389-518 as one table."""
    key = cfg["algorithm"].lower()
    if key not in _SCIPY_ALGORITHMS:
        raise RuntimeError("Method '" + cfg["algorithm"] + "' not recognized for scipy/" + flavour +
                           " library (valid values =  'lbfgs', 'bfgs', 'cg' ) ")
    driver, tolname, label = _SCIPY_ALGORITHMS[key]
    p = cfg["params"]
    verbose = cfg["verbose"]
    common.print_highlighted('method ' + label, verbose)
    if verbose:
        print("\t", "=" * 25)
        if show_caching:
            print("\t", "caching_yTilde_transposed :     ", cfg["cache_ytilde_transposed"])
        print("\t", "epsilon                   :     ", p["epsilon"])
        print("\t", "%-26s:     " % tolname, p[tolname])
        print("\t", "maxiter                   :     ", p["max_iterations"])
        print("\t", "=" * 25)
    kw = {"args": args, "fprime": fprime, "epsilon": p["epsilon"], tolname: p[tolname],
          "maxiter": p["max_iterations"]}
    if driver is sopt.fmin_l_bfgs_b:
        if verbose:
            kw["disp"] = 1
    else:
        kw["disp"] = bool(verbose)
        kw["full_output"] = True
    return driver(f, x0, **kw)


def _uses_device(cfg):
    return not (str(cfg["minimizer"]).upper() == 'SCIPY' and not bool(cfg["use_c_functions"]))


def find_optimum(GInit, G, y, yTilde, YTilde, theta, cfg):
    """Minimise the BioEn negative log-posterior over the n log-weights (log_weights.py:409-621): see `_find_optimum`.
    The matrix is HELD for the duration of the call (ext/c_bioen.py: hold): the initial objective, the optimisation and
    the averages of the optimum are served by one device copy -- at most ONE upload per call whatever the size of the
    matrix (none inside a caller's ``with optimize.resident(yTilde):``), as the reference's one ``yTilde.T.copy()`` per
    call (c_bioen.pyx:463-473)."""
    check_params_logweights(GInit, G, y, yTilde, YTilde)
    if str(cfg["minimizer"]).upper() not in ('LIBLBFGS', 'LBFGS', 'GSL', 'SCIPY'):      # rejected before anything touches the device
        raise RuntimeError("Library " + cfg["minimizer"] +
                           " not recognized (valid values =  'LIBLBFGS', 'GSL', 'scipy', 'scipy' ) ")
    if _uses_device(cfg):
        with c_bioen.hold(yTilde, YTilde):
            return _find_optimum(GInit, G, y, yTilde, YTilde, theta, cfg)
    return _find_optimum(GInit, G, y, yTilde, YTilde, theta, cfg)


def _find_optimum(GInit, G, y, yTilde, YTilde, theta, cfg):
    """Minimise the BioEn negative log-posterior over the n log-weights
    (log_weights.py:409-621).

    Returns (wopt (n,1), yopt (m,), gopt (n,), fmin_initial, fmin_final).
    cfg comes from ``minimize.Parameters``; minimizer "lbfgs"/"liblbfgs" runs the
    device-resident L-BFGS, "scipy" drives the device (or, with
    use_c_functions False, the numpy) objective from the host; "gsl": the library's GSL-style minimizers with all vectors in HBM."""

    caching = cfg["cache_ytilde_transposed"]
    if caching == "auto":
        caching = common.set_caching_heuristics(yTilde.shape[0], yTilde.shape[1])
    cfg["cache_ytilde_transposed"] = caching      # the reference writes this back, :440

    minimizer = cfg["minimizer"].upper()
    if minimizer not in ('LIBLBFGS', 'LBFGS', 'GSL', 'SCIPY'):
        raise RuntimeError("Library " + cfg["minimizer"] +
                           " not recognized (valid values =  'LIBLBFGS', 'GSL', 'scipy', 'scipy' ) ")
    use_c = bool(cfg["use_c_functions"])
    use_device = not (minimizer == 'SCIPY' and not use_c)

    g = GInit.copy()
    gPrime = np.asarray(g[:].T)[0]

    if use_device:
        fmin_initial = c_bioen.bioen_log_posterior_logw(gPrime, g, G, yTilde, YTilde, theta)
    else:
        fmin_initial = bioen_log_posterior_base(gPrime, g, G, yTilde, YTilde, theta)
    if cfg["verbose"]:
        print("fmin_initial", fmin_initial)

    start = time.time()
    if minimizer in ('LIBLBFGS', 'LBFGS'):
        common.print_highlighted("LOGW -- Library L-BFGS/HIP", cfg["verbose"])
        res = c_bioen.bioen_opt_lbfgs_logw(gPrime, G, yTilde, YTilde, theta, cfg)
    elif minimizer == 'GSL':
        common.print_highlighted("LOGW -- Library GSL/C", cfg["verbose"])
        res = c_bioen.bioen_opt_bfgs_logw(gPrime, G, yTilde, YTilde, theta, cfg)
    elif minimizer == 'SCIPY' and use_c:
        common.print_highlighted("LOGW -- Library scipy/HIP", cfg["verbose"])
        dev = _DeviceFdf(G, yTilde, YTilde, theta)
        try:
            res = _run_scipy(cfg, dev.f, dev.fprime, gPrime, (), "c", True)
        finally:
            dev.close()
    else:
        common.print_highlighted("LOGW -- Library scipy/PY", cfg["verbose"])
        res = _run_scipy(cfg, bioen_log_posterior_base, grad_bioen_log_posterior_base, gPrime,
                         (g, G, yTilde, YTilde, theta), "py", False)
    end = time.time()
    if cfg["verbose"]:
        print('time elapsed ', (end - start))

    gopt = res[0]
    fmin_final = res[1]
    wopt = getWOpt(G, gopt)
    if use_device and y is yTilde:
        yopt = c_bioen.get_ave(wopt, yTilde, YTilde)   # yTilde . wopt from the resident matrix
    else:
        yopt = common.getAve(wopt, y)

    if cfg["verbose"]:
        print("========================")
        print("fmin_initial  = ", fmin_initial)
        print("fmin_final    = ", fmin_final)
        print("========================")
    return wopt, yopt, gopt, fmin_initial, fmin_final


def find_optimum_series(GInit, G, y, yTilde, YTilde, thetas, cfg):
    """Not in the reference: ``find_optimum`` for a whole cold-started theta series in one device call
    (what bioen/analyze/procedure.py:62-67 does one theta at a time).  Minimizer "lbfgs"/"liblbfgs"
    only; every entry of the returned list is the 5-tuple ``find_optimum`` returns for that theta --
    the optimisation itself bit for bit, since a batched run equals the single runs."""
    check_params_logweights(GInit, G, y, yTilde, YTilde)
    if cfg["minimizer"].upper() not in ('LIBLBFGS', 'LBFGS'):
        raise RuntimeError("find_optimum_series needs the lbfgs minimizer, got " + str(cfg["minimizer"]))
    g = GInit.copy()
    gPrime = np.asarray(g[:].T)[0]
    with c_bioen.hold(yTilde, YTilde):                 # one device copy for the whole series: at most one upload
        fmin_initial = [c_bioen.bioen_log_posterior_logw(gPrime, g, G, yTilde, YTilde, float(t)) for t in thetas]
        gopts, fmins = c_bioen.bioen_opt_lbfgs_logw_series(gPrime, G, yTilde, YTilde, [float(t) for t in thetas], cfg)
        out = []
        for k in range(len(fmins)):
            wopt = getWOpt(G, gopts[k])
            yopt = c_bioen.get_ave(wopt, yTilde, YTilde) if y is yTilde else common.getAve(wopt, y)
            out.append((wopt, yopt, gopts[k], fmin_initial[k], fmins[k]))
    return out
