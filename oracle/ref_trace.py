"""TEST INFRASTRUCTURE ONLY -- the reference's liblbfgs driven evaluation by evaluation.

``lbfgs()`` of the vendored liblbfgs 1.10 (exported by ``oracle/_ref/libbioen_ref.so``, built by ``oracle/Makefile`` from
``/root/reference/third-party/liblbfgs-1.10``) is called through ctypes callbacks, so that every evaluation and every accepted
iteration of a run can be logged -- what ``_opt_lbfgs_forces`` (c_bioen_kernels_forces.c:574-662) does, with the curtain
open.  ``RefObjective`` is interface_lbfgs_forces' three C calls (c_bioen_kernels_forces.c:43-76) on host arrays;
``exact_fdf`` the same objective and gradient in 80-bit arithmetic (the truth the noise floors are measured against).

Only ``tests/``, the golden generators, ``tools/`` probes and ``bench.py``'s checker legs may import this module; the
product (``bioen_amd``) never does.
"""
import ctypes as C

import numpy as np

from . import ref_binding as R


class lbfgs_parameter_t(C.Structure):     # third-party/liblbfgs-1.10/include/lbfgs.h:196-343
    _fields_ = [("m", C.c_int), ("epsilon", C.c_double), ("past", C.c_int), ("delta", C.c_double),
                ("max_iterations", C.c_int), ("linesearch", C.c_int), ("max_linesearch", C.c_int),
                ("min_step", C.c_double), ("max_step", C.c_double), ("ftol", C.c_double), ("wolfe", C.c_double),
                ("gtol", C.c_double), ("xtol", C.c_double), ("orthantwise_c", C.c_double),
                ("orthantwise_start", C.c_int), ("orthantwise_end", C.c_int)]


EVAL_T = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_double)
PROG_T = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double,
                     C.c_double, C.c_double, C.c_int, C.c_int, C.c_int)


def traced_lbfgs(fdf, x0, params):
    """liblbfgs' own lbfgs() on `fdf(x) -> (f, grad)`; -> (x, fx, code, evaluations[], iterations[])"""
    L = R.lib()
    L.lbfgs_parameter_init.argtypes = [C.POINTER(lbfgs_parameter_t)]
    L.lbfgs.restype = C.c_int
    L.lbfgs.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), EVAL_T, PROG_T, C.c_void_p,
                        C.POINTER(lbfgs_parameter_t)]
    n = x0.size
    evals, its = [], []

    def ev(_inst, xp, gp, nn, step):
        x = np.ctypeslib.as_array(xp, (nn,)).copy()
        f, g = fdf(x)
        np.ctypeslib.as_array(gp, (nn,))[:] = g
        evals.append({"step": step, "f": float(f), "gnorm": float(np.sqrt(np.dot(g, g))), "x": x, "g": np.array(g)})
        return float(f)

    def pr(_inst, xp, gp, fx, xnorm, gnorm, step, nn, k, ls):
        its.append({"k": k, "ls": ls, "fx": fx, "xnorm": xnorm, "gnorm": gnorm, "step": step,
                    "ratio": gnorm / max(1.0, xnorm)})
        return 0

    p = lbfgs_parameter_t()
    L.lbfgs_parameter_init(C.byref(p))
    for k in ("linesearch", "max_iterations", "past", "max_linesearch"):
        setattr(p, k, int(params[k]))
    for k in ("delta", "epsilon", "ftol", "gtol", "wolfe"):
        setattr(p, k, float(params[k]))
    x = np.array(x0, dtype=np.float64)
    fx = C.c_double(0.0)
    code = L.lbfgs(n, x.ctypes.data_as(C.POINTER(C.c_double)), C.byref(fx), EVAL_T(ev), PROG_T(pr), None, C.byref(p))
    return x, fx.value, code, evals, its


class RefObjective(object):
    """interface_lbfgs_forces' three calls on host arrays (transposed cache built once)"""

    def __init__(self, yT, YT, w0, theta):
        self.yT, self.YT, self.w0, self.theta = R._a(yT), R._a(YT).ravel(), R._a(w0).ravel(), float(theta)
        self.m, self.n = self.yT.shape
        self.yTT = np.ascontiguousarray(self.yT.T)
        self.w = np.empty(self.n); self.tmp_n = np.empty(self.n); self.tmp_m = np.empty(self.m)

    def __call__(self, x):
        L, p = R.lib(), R._p
        x = np.ascontiguousarray(x)
        g = np.empty(self.m)
        L._get_weights_from_forces(p(self.w0), p(self.yT), p(x), p(self.w), 1, p(self.yTT), p(self.tmp_n), self.m, self.n)
        f = L._bioen_log_posterior_forces(p(self.w0), p(self.yT), p(self.YT), p(self.w), None, self.theta, 1, p(self.yTT),
                                          p(self.tmp_n), p(self.tmp_m), self.m, self.n)
        L._grad_bioen_log_posterior_forces(p(self.w0), p(self.yT), p(self.YT), p(self.w), p(g), self.theta, 1, p(self.yTT),
                                           p(self.tmp_n), p(self.tmp_m), self.m, self.n)
        return f, g


def exact_fdf(yT, YT, w0, theta, x, chunk=32768):
    """the same objective and gradient in 80-bit arithmetic, column chunks -> (f, grad) as float64 of the long double values"""
    LD = np.longdouble
    m, n = yT.shape
    xl = x.astype(LD)
    # pass 1: x_j = sum_i f_i y_ij, maximum
    xs = np.empty(n, dtype=LD)
    for c0 in range(0, n, chunk):
        xs[c0:c0 + chunk] = xl @ yT[:, c0:c0 + chunk].astype(LD)
    mx = xs.max()
    e = w0.astype(LD) * np.exp(xs - mx)
    w = e / e.sum()
    ybar = np.zeros(m, dtype=LD)
    for c0 in range(0, n, chunk):
        ybar += yT[:, c0:c0 + chunk].astype(LD) @ w[c0:c0 + chunk]
    r = ybar - YT.astype(LD)
    lw = np.log(w) - np.log(w0.astype(LD))
    f = LD(theta) * (w * lw).sum() + LD(0.5) * (r * r).sum()
    grad = np.zeros(m, dtype=LD)
    for c0 in range(0, n, chunk):
        blk = yT[:, c0:c0 + chunk].astype(LD)
        b = r @ blk
        t = (LD(theta) * (1 + lw[c0:c0 + chunk]) + b) * w[c0:c0 + chunk]
        grad += (blk - ybar[:, None]) @ t
    return float(f), grad.astype(np.float64)
