"""TEST INFRASTRUCTURE ONLY -- ctypes binding to ``oracle/_ref/libbioen_ref.so``.

``libbioen_ref.so`` is the reference's own C path (c_bioen_common.c,
c_bioen_kernels_logw.c, c_bioen_kernels_forces.c, c_bioen_error.c and the
vendored liblbfgs 1.10) compiled by ``oracle/Makefile`` straight from
``/root/reference``.  This module plays the part of the reference's Cython
layer (``bioen/optimize/ext/c_bioen.pyx``): it allocates the scratch arrays,
packs ``params_t`` (``c_bioen_common.h:44-60``) and calls the C entry points.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.  The product
(``bioen_amd``) never does.
"""
import ctypes as C
import os

import numpy as np

from . import cpus

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "libbioen_ref.so")

dp = C.POINTER(C.c_double)


class params_t(C.Structure):
    # field order of c_bioen_common.h:44-60
    _fields_ = [("forces", dp), ("w0", dp), ("g", dp), ("G", dp),
                ("yTilde", dp), ("YTilde", dp), ("w", dp), ("result", dp),
                ("theta", C.c_double), ("yTildeT", dp), ("caching", C.c_int),
                ("tmp_n", dp), ("tmp_m", dp), ("m", C.c_int), ("n", C.c_int)]


class lbfgs_config_params(C.Structure):
    # c_bioen_common.h:69-79
    _fields_ = [("linesearch", C.c_int), ("max_iterations", C.c_int),
                ("delta", C.c_double), ("epsilon", C.c_double),
                ("ftol", C.c_double), ("gtol", C.c_double),
                ("wolfe", C.c_double), ("past", C.c_int),
                ("max_linesearch", C.c_int)]


class visual_params(C.Structure):
    # c_bioen_common.h:89-92
    _fields_ = [("debug", C.c_size_t), ("verbose", C.c_size_t)]


_lib = None


def available():
    return os.path.isfile(_PATH)


def lib():
    global _lib
    if _lib is None:
        cpus.default_omp_threads()
        L = C.CDLL(_PATH)
        L._get_weights.restype = C.c_double
        L._get_weights.argtypes = [dp, dp, C.c_size_t]
        L._bioen_log_posterior_logw.restype = C.c_double
        L._bioen_log_posterior_logw.argtypes = [dp, dp, dp, dp, dp, dp, C.c_double, C.c_int, dp,
                                                dp, dp, C.c_int, C.c_int, C.c_double]
        L._grad_bioen_log_posterior_logw.restype = None
        L._grad_bioen_log_posterior_logw.argtypes = [dp, dp, dp, dp, dp, dp, C.c_double, C.c_int, dp,
                                                     dp, dp, C.c_int, C.c_int, C.c_double]
        L._opt_lbfgs_logw.restype = C.c_double
        L._opt_lbfgs_logw.argtypes = [params_t, lbfgs_config_params, visual_params, C.POINTER(C.c_int)]
        L._get_weights_from_forces.restype = None
        L._get_weights_from_forces.argtypes = [dp, dp, dp, dp, C.c_int, dp, dp, C.c_size_t, C.c_size_t]
        L._bioen_log_posterior_forces.restype = C.c_double
        L._bioen_log_posterior_forces.argtypes = [dp, dp, dp, dp, dp, C.c_double, C.c_int, dp, dp, dp,
                                                  C.c_int, C.c_int]
        L._grad_bioen_log_posterior_forces.restype = None
        L._grad_bioen_log_posterior_forces.argtypes = [dp, dp, dp, dp, dp, C.c_double, C.c_int, dp, dp, dp,
                                                       C.c_int, C.c_int]
        L._opt_lbfgs_forces.restype = C.c_double
        L._opt_lbfgs_forces.argtypes = [params_t, lbfgs_config_params, visual_params, C.POINTER(C.c_int)]
        L._bioen_chi_squared.restype = C.c_double
        L._bioen_chi_squared.argtypes = [dp, dp, dp, dp, C.c_size_t, C.c_size_t]
        L._set_fast_openmp_flag.argtypes = [C.c_int]
        L._get_fast_openmp_flag.restype = C.c_int
        L._omp_set_num_threads.argtypes = [C.c_int]
        L.lbfgs_strerror.restype = C.c_char_p
        L.lbfgs_strerror.argtypes = [C.c_int]
        _lib = L
    return _lib


def _a(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(dp)


def set_fast_openmp_flag(flag):
    lib()._set_fast_openmp_flag(int(flag))


def omp_set_num_threads(n):
    lib()._omp_set_num_threads(int(n))


def get_weights(g):
    g = _a(g).ravel()
    w = np.empty_like(g)
    s = lib()._get_weights(_p(g), _p(w), g.size)
    return w, s


def logw_f(gPrime, G, yTilde, YTilde, theta):
    """A1 + A5 with the TRUE G (i.e. what interface_lbfgs_logw evaluates,
    c_bioen_kernels_logw.c:550-554) -- not the A7-buggy Cython wrapper."""
    gPrime, G, yTilde, YTilde = _a(gPrime).ravel(), _a(G).ravel(), _a(yTilde), _a(YTilde).ravel()
    m, n = yTilde.shape
    w = np.empty(n); tmp_n = np.empty(n); tmp_m = np.empty(m)
    L = lib()
    s = L._get_weights(_p(gPrime), _p(w), n)
    return L._bioen_log_posterior_logw(_p(gPrime), _p(G), _p(yTilde), _p(YTilde), _p(w), None,
                                       float(theta), 0, None, _p(tmp_n), _p(tmp_m), m, n, s)


def logw_df(gPrime, G, yTilde, YTilde, theta, caching=True):
    gPrime, G, yTilde, YTilde = _a(gPrime).ravel(), _a(G).ravel(), _a(yTilde), _a(YTilde).ravel()
    m, n = yTilde.shape
    w = np.empty(n); tmp_n = np.empty(n); tmp_m = np.empty(m); grad = np.empty(n)
    yT = np.ascontiguousarray(yTilde.T) if caching else np.empty(1)
    L = lib()
    s = L._get_weights(_p(gPrime), _p(w), n)
    L._grad_bioen_log_posterior_logw(_p(gPrime), _p(G), _p(yTilde), _p(YTilde), _p(w), _p(grad),
                                     float(theta), 1 if caching else 0, _p(yT), _p(tmp_n), _p(tmp_m),
                                     m, n, s)
    return grad


def _lbfgs_cfg(params):
    c = lbfgs_config_params()
    for k in ("linesearch", "max_iterations", "past", "max_linesearch"):
        setattr(c, k, int(params[k]))
    for k in ("delta", "epsilon", "ftol", "gtol", "wolfe"):
        setattr(c, k, float(params[k]))
    return c


def opt_lbfgs_logw(g0, G, yTilde, YTilde, theta, params, caching=True, verbose=0):
    """c_bioen.pyx:441-520 -> _opt_lbfgs_logw.  Returns (gopt, fmin, code)."""
    g0, G, yTilde, YTilde = _a(g0).ravel().copy(), _a(G).ravel(), _a(yTilde), _a(YTilde).ravel()
    m, n = yTilde.shape
    w = np.empty(n); tmp_n = np.empty(n); tmp_m = np.empty(m); result = np.empty(n)
    yT = np.ascontiguousarray(yTilde.T) if caching else np.empty(1)
    p = params_t()
    p.g, p.G, p.yTilde, p.YTilde, p.w, p.result = _p(g0), _p(G), _p(yTilde), _p(YTilde), _p(w), _p(result)
    p.theta, p.yTildeT, p.caching = float(theta), _p(yT), 1 if caching else 0
    p.tmp_n, p.tmp_m, p.m, p.n = _p(tmp_n), _p(tmp_m), m, n
    v = visual_params(0, int(verbose))
    err = C.c_int(0)
    fmin = lib()._opt_lbfgs_logw(p, _lbfgs_cfg(params), v, C.byref(err))
    return result, fmin, err.value


def forces_weights(forces, w0, yTilde, caching=True):
    forces, w0, yTilde = _a(forces).ravel(), _a(w0).ravel(), _a(yTilde)
    m, n = yTilde.shape
    w = np.empty(n); tmp_n = np.empty(n)
    yT = np.ascontiguousarray(yTilde.T) if caching else np.empty(1)
    lib()._get_weights_from_forces(_p(w0), _p(yTilde), _p(forces), _p(w), 1 if caching else 0, _p(yT),
                                   _p(tmp_n), m, n)
    return w


def forces_f(forces, w0, yTilde, YTilde, theta):
    forces, w0, yTilde, YTilde = _a(forces).ravel(), _a(w0).ravel(), _a(yTilde), _a(YTilde).ravel()
    m, n = yTilde.shape
    w = np.empty(n); tmp_n = np.empty(n); tmp_m = np.empty(m)
    L = lib()
    L._get_weights_from_forces(_p(w0), _p(yTilde), _p(forces), _p(w), 0, None, _p(tmp_n), m, n)
    return L._bioen_log_posterior_forces(_p(w0), _p(yTilde), _p(YTilde), _p(w), None, float(theta), 0, None,
                                         _p(tmp_n), _p(tmp_m), m, n)


def forces_df(forces, w0, yTilde, YTilde, theta, caching=True):
    forces, w0, yTilde, YTilde = _a(forces).ravel(), _a(w0).ravel(), _a(yTilde), _a(YTilde).ravel()
    m, n = yTilde.shape
    w = np.empty(n); tmp_n = np.empty(n); tmp_m = np.empty(m); grad = np.empty(m)
    yT = np.ascontiguousarray(yTilde.T) if caching else np.empty(1)
    c = 1 if caching else 0
    L = lib()
    L._get_weights_from_forces(_p(w0), _p(yTilde), _p(forces), _p(w), c, _p(yT), _p(tmp_n), m, n)
    L._grad_bioen_log_posterior_forces(_p(w0), _p(yTilde), _p(YTilde), _p(w), _p(grad), float(theta), c,
                                       _p(yT), _p(tmp_n), _p(tmp_m), m, n)
    return grad


def opt_lbfgs_forces(f0, w0, yTilde, YTilde, theta, params, caching=True, verbose=0):
    """c_bioen.pyx:719-792 -> _opt_lbfgs_forces.  Returns (forces_opt, fmin, code)."""
    f0, w0, yTilde, YTilde = _a(f0).ravel().copy(), _a(w0).ravel(), _a(yTilde), _a(YTilde).ravel()
    m, n = yTilde.shape
    w = np.empty(n); tmp_n = np.empty(n); tmp_m = np.empty(m); result = np.empty(m)
    yT = np.ascontiguousarray(yTilde.T) if caching else np.empty(1)
    p = params_t()
    p.forces, p.w0, p.yTilde, p.YTilde, p.w, p.result = _p(f0), _p(w0), _p(yTilde), _p(YTilde), _p(w), _p(result)
    p.theta, p.yTildeT, p.caching = float(theta), _p(yT), 1 if caching else 0
    p.tmp_n, p.tmp_m, p.m, p.n = _p(tmp_n), _p(tmp_m), m, n
    v = visual_params(0, int(verbose))
    err = C.c_int(0)
    fmin = lib()._opt_lbfgs_forces(p, _lbfgs_cfg(params), v, C.byref(err))
    return result, fmin, err.value


def lbfgs_strerror(code):
    return lib().lbfgs_strerror(int(code)).decode()
