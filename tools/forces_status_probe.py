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
from oracle.ref_trace import traced_lbfgs, RefObjective, exact_fdf      # noqa: E402
from oracle import cpus                  # noqa: E402

LBFGS_DEFAULTS = dict(linesearch=2, max_iterations=5000, delta=1e-6, epsilon=1e-6, ftol=1e-5, gtol=0.9,
                      wolfe=0.9, past=10, max_linesearch=100)


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
