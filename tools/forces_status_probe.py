#!/usr/bin/env python3
"""r06 probe: how do the forces-method runs of configs[4] END at large theta -- on the device, in the
reference, and why not the same way?

The reference's liblbfgs `lbfgs()` (exported by oracle/_ref/libbioen_ref.so) is driven from here through ctypes callbacks,
so that every evaluation and every accepted iteration can be logged; the objective behind it is either the reference's own
three C calls (what interface_lbfgs_forces does, c_bioen_kernels_forces.c:43-76) or the device's forces_fdf.  Checker-side
tooling: imports oracle/, never part of the product.

    python3 tools/forces_status_probe.py cpu   [N] [M]      # reference only, numpy matrix (build container)
    python3 tools/forces_status_probe.py gpu   [N] [M]      # device generator, device vs reference (GPU box)
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_binding as R      # noqa: E402
from oracle import cpus                  # noqa: E402

LBFGS_DEFAULTS = dict(linesearch=2, max_iterations=5000, delta=1e-6, epsilon=1e-6, ftol=1e-5, gtol=0.9,
                      wolfe=0.9, past=10, max_linesearch=100)


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


def show(tag, code, fx, evals, its):
    print("%s: code %d, fmin %.16g, %d iterations, %d evaluations" % (tag, code, fx, len(its), len(evals)))
    for it in its:
        print("    k=%d ls=%d f=%.16g |g|=%.6e |x|=%.6e |g|/max(1,|x|)=%.6e step=%.4g"
              % (it["k"], it["ls"], it["fx"], it["gnorm"], it["xnorm"], it["ratio"], it["step"]))
    tail = evals[len(evals) - min(len(evals), 12):]
    print("    last evaluations: " + ", ".join("(stp %.3g f-f0 %.3e |g| %.3e)" % (e["step"], e["f"] - evals[0]["f"], e["gnorm"])
                                               for e in tail))


def survey_matrix(M, N, seed=12345):
    rng = np.random.default_rng(seed)
    YTrue = rng.uniform(1, 10, M)
    sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
    yT = np.empty((M, N))
    for i in range(M):
        yT[i, :] = rng.normal(YTrue[i], sig_sim[i], N) / sig_exp[i]
    YT = rng.normal(YTrue, sig_exp) / sig_exp
    return yT, YT


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "cpu"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    thetas = [float(t) for t in np.logspace(3, -0.5, 8)[:3]]
    cores = cpus.usable_cpus()
    w0 = np.full(N, 1.0 / N)
    x0 = np.zeros(M)
    ctx = None
    if mode == "gpu":
        import bioen_amd
        sys.path.insert(0, ROOT)
        import bench
        YTrue, sig_sim, sig_exp, YT = bench.synthetic_targets(M)
        ctx = bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YT, seed=12345)
        yT = ctx.read_ytilde()
    else:
        yT, YT = survey_matrix(M, N)
    print("matrix %d x %d, %d cores" % (M, N, cores))
    for theta in thetas:
        print("==== theta = %g" % theta)
        obj = RefObjective(yT, YT, w0, theta)
        variants = [(1, cores, 0), (1, cores, 1), (0, cores, 0), (1, max(1, cores // 2), 0)] + ([(1, 1, 0)] if mode == "cpu" else [])
        ref_runs = {}
        for flag, nthr, rep in variants:
            R.set_fast_openmp_flag(flag)
            R.omp_set_num_threads(nthr)
            t0 = time.perf_counter()
            x, fx, code, evals, its = traced_lbfgs(obj, x0, LBFGS_DEFAULTS)
            tag = "reference fast_openmp=%d threads=%d run %d (%.1f s)" % (flag, nthr, rep, time.perf_counter() - t0)
            show(tag, code, fx, evals, its)
            ref_runs.setdefault((flag, nthr), (x, fx, code, evals, its))
            # and the reference's own driver, to see that the traced run IS its run
            _, fmin2, code2 = R.opt_lbfgs_forces(x0, w0, yT, YT, theta, LBFGS_DEFAULTS)
            print("    _opt_lbfgs_forces itself: code %d fmin %.16g" % (code2, fmin2))
        R.set_fast_openmp_flag(1)
        R.omp_set_num_threads(cores)
        x, fx, code, evals, its = ref_runs[(1, cores)]
        # how good are the reference's own numbers at its last accepted points?  80-bit truth
        for e in evals[-2:]:
            f_ex, g_ex = exact_fdf(yT, YT, w0, theta, e["x"])
            print("    reference eval: f err %.3e (abs), |g| %.6e exact |g| %.6e, |g - g_exact| %.3e"
                  % (e["f"] - f_ex, e["gnorm"], np.sqrt(g_ex @ g_ex), np.sqrt(((e["g"] - g_ex) ** 2).sum())))
        if ctx is not None:
            def dev(xx):
                return ctx.forces_fdf(xx, w0, theta)
            xd, fxd, coded, evalsd, itsd = traced_lbfgs(dev, x0, LBFGS_DEFAULTS)
            show("liblbfgs on the DEVICE objective", coded, fxd, evalsd, itsd)
            for e in evalsd[:1] + evalsd[-2:]:
                f_ex, g_ex = exact_fdf(yT, YT, w0, theta, e["x"])
                print("    device eval: f err %.3e (abs), |g| %.6e exact |g| %.6e, |g - g_exact| %.3e"
                      % (e["f"] - f_ex, e["gnorm"], np.sqrt(g_ex @ g_ex), np.sqrt(((e["g"] - g_ex) ** 2).sum())))
            # the reference's numbers at the device's points and vice versa
            for e in evals[-2:]:
                fd, gd = dev(e["x"])
                print("    at a reference point: f_dev - f_ref %.3e, |g_dev - g_ref| %.3e, |g_ref| %.6e |g_dev| %.6e"
                      % (fd - e["f"], np.sqrt(((gd - e["g"]) ** 2).sum()), e["gnorm"], np.sqrt(gd @ gd)))
            res, _, info = ctx.opt_lbfgs_forces(x0, w0, theta, LBFGS_DEFAULTS, want_weights=False)
            print("    device engine: code %d fmin %.16g iterations %d evaluations %d"
                  % (info.lbfgs_code, info.fmin, info.iterations, info.evaluations))
    if ctx is not None:
        ctx.close()


if __name__ == "__main__":
    main()
