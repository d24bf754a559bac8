"""TEST INFRASTRUCTURE ONLY -- ctypes binding to ``oracle/libbioen_oracle.so``
(the CPU restatement in ``oracle/bioen_oracle.c``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; ``bioen_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import cpus

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libbioen_oracle.so")

dp = C.POINTER(C.c_double)


class lbfgs_config(C.Structure):
    _fields_ = [("linesearch", C.c_int), ("max_iterations", C.c_int),
                ("delta", C.c_double), ("epsilon", C.c_double),
                ("ftol", C.c_double), ("gtol", C.c_double),
                ("wolfe", C.c_double), ("past", C.c_int),
                ("max_linesearch", C.c_int)]


class lbfgs_stats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("evaluations", C.c_int)]


class gsl_config(C.Structure):
    _fields_ = [("step_size", C.c_double), ("tol", C.c_double), ("max_iterations", C.c_int),
                ("algorithm", C.c_int)]


class gsl_stats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("f_evaluations", C.c_int), ("g_evaluations", C.c_int)]


GSL_ALGORITHMS = {"conjugate_fr": 0, "conjugate_pr": 1, "bfgs2": 2, "bfgs": 3, "steepest_descent": 4}
GSL_DEFAULTS = dict(algorithm="bfgs2", step_size=0.01, tol=0.001, max_iterations=5000)   # bioen_optimize.yaml

LBFGS_DEFAULTS = dict(linesearch=2, max_iterations=5000, delta=1e-6, epsilon=1e-6, ftol=1e-5,
                      gtol=0.9, wolfe=0.9, past=10, max_linesearch=100)

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libbioen_oracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_PATH):
            build()
        cpus.default_omp_threads()
        L = C.CDLL(_PATH)
        L.oracle_logw_weights.restype = C.c_double
        L.oracle_logw_weights.argtypes = [dp, dp, C.c_size_t]
        L.oracle_chi_squared.restype = C.c_double
        L.oracle_chi_squared.argtypes = [dp, dp, dp, dp, C.c_size_t, C.c_size_t]
        L.oracle_logw_fdf.restype = C.c_double
        L.oracle_logw_fdf.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, C.c_double, dp, dp]
        L.oracle_forces_weights.restype = None
        L.oracle_forces_weights.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp]
        L.oracle_forces_fdf.restype = C.c_double
        L.oracle_forces_fdf.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, C.c_double, dp, dp]
        for name in ("oracle_opt_lbfgs_logw", "oracle_opt_lbfgs_forces"):
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, C.c_double, C.POINTER(lbfgs_config), dp, dp,
                           C.POINTER(lbfgs_stats)]
        L.oracle_selftest_lbfgs.restype = C.c_int
        L.oracle_selftest_lbfgs.argtypes = [C.c_int, C.c_int, dp, C.POINTER(lbfgs_config), dp, dp,
                                            C.POINTER(lbfgs_stats)]
        for name in ("oracle_opt_gsl_logw", "oracle_opt_gsl_forces"):
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, C.c_double, C.POINTER(gsl_config), dp, dp,
                           C.POINTER(gsl_stats)]
        L.oracle_multimin_testfn_dim.restype = C.c_int
        L.oracle_multimin_testfn_dim.argtypes = [C.c_int]
        L.oracle_selftest_multimin.restype = C.c_int
        L.oracle_selftest_multimin.argtypes = [C.c_int, C.c_int, dp, dp, dp, C.POINTER(gsl_stats)]
        _lib = L
    return _lib


def _a(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(dp)


def _cfg(params):
    full = dict(LBFGS_DEFAULTS)
    full.update(params or {})
    c = lbfgs_config()
    for k in ("linesearch", "max_iterations", "past", "max_linesearch"):
        setattr(c, k, int(full[k]))
    for k in ("delta", "epsilon", "ftol", "gtol", "wolfe"):
        setattr(c, k, float(full[k]))
    return c


def logw_weights(g):
    g = _a(g).ravel()
    w = np.empty_like(g)
    logs = lib().oracle_logw_weights(_p(g), _p(w), g.size)
    return w, logs


def chi_squared(w, yTilde, YTilde):
    w, yTilde, YTilde = _a(w).ravel(), _a(yTilde), _a(YTilde).ravel()
    m, n = yTilde.shape
    yave = np.empty(m)
    val = lib().oracle_chi_squared(_p(w), _p(yTilde), _p(YTilde), _p(yave), m, n)
    return val, yave


def logw_fdf(g, G, yTilde, YTilde, theta):
    """-> (f, grad[n], w[n])"""
    g, G, yTilde, YTilde = _a(g).ravel(), _a(G).ravel(), _a(yTilde), _a(YTilde).ravel()
    m, n = yTilde.shape
    grad = np.empty(n); w = np.empty(n)
    f = lib().oracle_logw_fdf(m, n, _p(yTilde), _p(YTilde), _p(g), _p(G), float(theta), _p(grad), _p(w))
    return f, grad, w


def forces_weights(forces, w0, yTilde):
    forces, w0, yTilde = _a(forces).ravel(), _a(w0).ravel(), _a(yTilde)
    m, n = yTilde.shape
    w = np.empty(n)
    lib().oracle_forces_weights(m, n, _p(yTilde), _p(forces), _p(w0), _p(w))
    return w


def forces_fdf(forces, w0, yTilde, YTilde, theta):
    """-> (f, grad[m], w[n])"""
    forces, w0, yTilde, YTilde = _a(forces).ravel(), _a(w0).ravel(), _a(yTilde), _a(YTilde).ravel()
    m, n = yTilde.shape
    grad = np.empty(m); w = np.empty(n)
    f = lib().oracle_forces_fdf(m, n, _p(yTilde), _p(YTilde), _p(forces), _p(w0), float(theta), _p(grad), _p(w))
    return f, grad, w


def opt_lbfgs_logw(g0, G, yTilde, YTilde, theta, params=None):
    """-> (gopt, fmin, code, iterations, evaluations)"""
    g0, G, yTilde, YTilde = _a(g0).ravel(), _a(G).ravel(), _a(yTilde), _a(YTilde).ravel()
    m, n = yTilde.shape
    res = np.empty(n); fmin = C.c_double(0.0); st = lbfgs_stats()
    cfg = _cfg(params)
    code = lib().oracle_opt_lbfgs_logw(m, n, _p(yTilde), _p(YTilde), _p(g0), _p(G), float(theta),
                                       C.byref(cfg), _p(res), C.byref(fmin), C.byref(st))
    return res, fmin.value, code, st.iterations, st.evaluations


def opt_lbfgs_forces(f0, w0, yTilde, YTilde, theta, params=None):
    """-> (forces_opt, fmin, code, iterations, evaluations)"""
    f0, w0, yTilde, YTilde = _a(f0).ravel(), _a(w0).ravel(), _a(yTilde), _a(YTilde).ravel()
    m, n = yTilde.shape
    res = np.empty(m); fmin = C.c_double(0.0); st = lbfgs_stats()
    cfg = _cfg(params)
    code = lib().oracle_opt_lbfgs_forces(m, n, _p(yTilde), _p(YTilde), _p(f0), _p(w0), float(theta),
                                         C.byref(cfg), _p(res), C.byref(fmin), C.byref(st))
    return res, fmin.value, code, st.iterations, st.evaluations


def selftest_lbfgs(kind, x0, params=None):
    """-> (x, fmin, code, iterations, evaluations) for the built-in analytic objectives"""
    x0 = _a(x0).ravel()
    out = np.empty_like(x0); fmin = C.c_double(0.0); st = lbfgs_stats()
    cfg = _cfg(params)
    code = lib().oracle_selftest_lbfgs(int(kind), x0.size, _p(x0), C.byref(cfg), _p(out), C.byref(fmin), C.byref(st))
    return out, fmin.value, code, st.iterations, st.evaluations


def _gsl_cfg(params):
    full = dict(GSL_DEFAULTS)
    full.update(params or {})
    alg = full["algorithm"]
    c = gsl_config()
    c.step_size, c.tol = float(full["step_size"]), float(full["tol"])
    c.max_iterations = int(full["max_iterations"])
    c.algorithm = GSL_ALGORITHMS[alg] if isinstance(alg, str) else int(alg)
    return c


def opt_gsl_logw(g0, G, yTilde, YTilde, theta, params=None):
    """-> (gopt, fmin, gsl status, iterations, (f evaluations, gradient evaluations))"""
    g0, G, yTilde, YTilde = _a(g0).ravel(), _a(G).ravel(), _a(yTilde), _a(YTilde).ravel()
    m, n = yTilde.shape
    res = np.empty(n); fmin = C.c_double(0.0); st = gsl_stats()
    cfg = _gsl_cfg(params)
    code = lib().oracle_opt_gsl_logw(m, n, _p(yTilde), _p(YTilde), _p(g0), _p(G), float(theta), C.byref(cfg),
                                     _p(res), C.byref(fmin), C.byref(st))
    return res, fmin.value, code, st.iterations, (st.f_evaluations, st.g_evaluations)


def opt_gsl_forces(f0, w0, yTilde, YTilde, theta, params=None):
    f0, w0, yTilde, YTilde = _a(f0).ravel(), _a(w0).ravel(), _a(yTilde), _a(YTilde).ravel()
    m, n = yTilde.shape
    res = np.empty(m); fmin = C.c_double(0.0); st = gsl_stats()
    cfg = _gsl_cfg(params)
    code = lib().oracle_opt_gsl_forces(m, n, _p(yTilde), _p(YTilde), _p(f0), _p(w0), float(theta), C.byref(cfg),
                                       _p(res), C.byref(fmin), C.byref(st))
    return res, fmin.value, code, st.iterations, (st.f_evaluations, st.g_evaluations)


class ref_kernels(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("get_weights", "logw_f", "logw_df", "forces_weights", "forces_f", "forces_df")]


_REF_SYMBOLS = ("_get_weights", "_bioen_log_posterior_logw", "_grad_bioen_log_posterior_logw",
                "_get_weights_from_forces", "_bioen_log_posterior_forces", "_grad_bioen_log_posterior_forces")


def opt_gsl_refobj(ref_cdll, kind, x0, fixed, yTilde, YTilde, theta, params=None, caching=True):
    """The restated GSL minimizers on the REFERENCE's own objective: `ref_cdll` is a ctypes.CDLL of a
    build of the reference (oracle/_ref/libbioen_ref.so, or the Cython extension of a full build);
    the C glue calls its kernels the way the reference's GSL callbacks do.
    kind 'logw': x0 = GInit, fixed = G; 'forces': x0 = forces_init, fixed = w0.
    -> (x, fmin, gsl status, iterations, (f evaluations, gradient evaluations))"""
    x0, fixed, yTilde, YTilde = _a(x0).ravel(), _a(fixed).ravel(), _a(yTilde), _a(YTilde).ravel()
    m, n = yTilde.shape
    k = ref_kernels(*[C.cast(getattr(ref_cdll, name), C.c_void_p) for name in _REF_SYMBOLS])
    yT = np.ascontiguousarray(yTilde.T) if caching else None
    forces = 1 if kind == "forces" else 0
    res = np.empty(m if forces else n); fmin = C.c_double(0.0); st = gsl_stats()
    cfg = _gsl_cfg(params)
    fn = lib().oracle_opt_gsl_refobj
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(ref_kernels), C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, C.c_double,
                   C.POINTER(gsl_config), dp, C.POINTER(C.c_double), C.POINTER(gsl_stats)]
    code = fn(C.byref(k), forces, m, n, _p(yTilde), _p(yT) if caching else None, _p(YTilde), _p(x0), _p(fixed),
              float(theta), C.byref(cfg), _p(res), C.byref(fmin), C.byref(st))
    return res, fmin.value, code, st.iterations, (st.f_evaluations, st.g_evaluations)


class RefObjective(object):
    """The reference's objective as a plain C callback: `.fn` (address of oracle_refobj_eval) and
    `.handle` (its `user` argument) can be handed to any minimizer that takes
    ``int fn(void* user, const double* x, double* f, double* grad_or_NULL)``."""

    def __init__(self, ref_cdll, kind, fixed, yTilde, YTilde, theta, caching=True):
        self._keep = [_a(fixed).ravel(), _a(yTilde), _a(YTilde).ravel()]
        fixed, yTilde, YTilde = self._keep
        m, n = yTilde.shape
        yT = np.ascontiguousarray(yTilde.T) if caching else None
        self._keep.append(yT)
        k = ref_kernels(*[C.cast(getattr(ref_cdll, name), C.c_void_p) for name in _REF_SYMBOLS])
        L = lib()
        L.oracle_refobj_create.restype = C.c_void_p
        L.oracle_refobj_create.argtypes = [C.POINTER(ref_kernels), C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, C.c_double]
        L.oracle_refobj_destroy.argtypes = [C.c_void_p]
        self.handle = C.c_void_p(L.oracle_refobj_create(C.byref(k), 1 if kind == "forces" else 0, m, n, _p(yTilde),
                                                        _p(yT) if caching else None, _p(YTilde), _p(fixed), float(theta)))
        assert self.handle.value
        self.fn = C.cast(L.oracle_refobj_eval, C.c_void_p)

    def close(self):
        if self.handle is not None:
            lib().oracle_refobj_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


MULTIMIN_TESTS = {   # multimin/test_funcs.c start points
    "Roth": (0, [4.5, 3.5]), "Wood": (1, [-3.0, -1.0, -3.0, -1.0]), "Rosenbrock": (2, [-1.2, 1.0]),
    "Rosenbrock1": (2, [1.0, 1.0]), "SimpleAbs": (3, [1.0, 2.0]),
}


def selftest_multimin(algorithm, kind, x0):
    """GSL's test_fdf protocol -> (x, f, status, iterations, (f evaluations, gradient evaluations))"""
    x0 = _a(x0).ravel()
    assert x0.size == lib().oracle_multimin_testfn_dim(int(kind))
    out = np.empty_like(x0); fmin = C.c_double(0.0); st = gsl_stats()
    alg = GSL_ALGORITHMS[algorithm] if isinstance(algorithm, str) else int(algorithm)
    code = lib().oracle_selftest_multimin(alg, int(kind), _p(x0), _p(out), C.byref(fmin), C.byref(st))
    return out, fmin.value, code, st.iterations, (st.f_evaluations, st.g_evaluations)
