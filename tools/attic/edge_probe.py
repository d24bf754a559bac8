"""Edge cases of the reference (w0 zeros / denormals, underflowing softmax, analyze-style 1e-150 weights, theta = 0
forces) device against oracle/_ref: prints the margins tests/test_hip_edgecases.py asserts."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bioen_amd                                   # noqa: E402
from oracle import ref_binding as R, cpus          # noqa: E402

LBFGS_DEFAULTS = dict(linesearch=2, max_iterations=5000, delta=1e-6, epsilon=1e-6, ftol=1e-5, gtol=0.9, wolfe=0.9,
                      past=10, max_linesearch=100)
CONV = dict(LBFGS_DEFAULTS, epsilon=1e-9, delta=0.0, past=0, max_iterations=200000)


def targets(M, seed=12345):
    rng = np.random.default_rng(seed)
    YTrue = rng.uniform(1, 10, M)
    sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
    return YTrue, sig_sim, sig_exp, rng.normal(YTrue, sig_exp) / sig_exp


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def main():
    R.set_fast_openmp_flag(0)
    R.omp_set_num_threads(cpus.usable_cpus())
    for M, N in ((96, 20000), (512, 20000), (600, 20000), (1056, 12000)):
        YTrue, sig_sim, sig_exp, YTilde = targets(M)
        rng = np.random.default_rng(M)
        w0 = rng.dirichlet(np.ones(N) * 2.0)
        w0[rng.choice(N, 50, replace=False)] = 0.0
        w0[rng.choice(N, 50, replace=False)] = 1e-310
        w0[rng.choice(N, 5, replace=False)] = 4.9e-324
        with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
            yT = np.ascontiguousarray(ctx.read_ytilde())
            for scale in (1e-3, 200.0 / (5 * np.sqrt(M))):
                forces = scale * rng.standard_normal(M)
                for theta in (0.0, 3.0):
                    f, g = ctx.forces_fdf(forces, w0, theta)
                    w_ref = R.forces_weights(forces, w0, yT)
                    fr = R.forces_f(forces, w0, yT, YTilde, theta)
                    gr = R.forces_df(forces, w0, yT, YTilde, theta)
                    nz = int((w_ref == 0).sum()), int(((w_ref > 0) & (w_ref < 2.3e-308)).sum())
                    print("forces M=%d scale=%.3g theta=%g: w zeros/denormals %s  f rel %.2e  grad rel %.2e  (f=%.6g)"
                          % (M, scale, theta, nz, rel(f, fr), np.abs(g - gr).max() / np.abs(gr).max(), fr), flush=True)
            for theta in (30.0, 3.0):
                t0 = time.time()
                fo, wo, info = ctx.opt_lbfgs_forces(np.zeros(M), w0, theta, CONV)
                t1 = time.time()
                fr_, fmin_r, code_r = R.opt_lbfgs_forces(np.zeros(M), w0, yT, YTilde, theta, CONV)
                w_r = R.forces_weights(fr_, w0, yT)
                print("  lbfgs M=%d theta=%g: codes %d/%d fmin rel %.2e  w diff %.2e max(w)  (%d it, %.2fs / %.2fs)"
                      % (M, theta, info.lbfgs_code, code_r, rel(info.fmin, fmin_r), np.abs(wo - w_r).max() / w_r.max(),
                         info.iterations, t1 - t0, time.time() - t1), flush=True)

    # analyze-style log-weights: zeros replaced by 1e-150, G = log w - log w[-1]
    for M, N, last_tiny in ((256, 100000, False), (256, 100000, True), (1024, 20000, True)):
        YTrue, sig_sim, sig_exp, YTilde = targets(M)
        rng = np.random.default_rng(5 + M + last_tiny)
        w0 = rng.dirichlet(np.ones(N) * 0.5)
        w0[rng.choice(N - 1, N // 20, replace=False)] = 1e-150
        winit = rng.dirichlet(np.ones(N) * 0.5)
        winit[rng.choice(N - 1, N // 20, replace=False)] = 1e-150
        if last_tiny:
            w0[-1] = 1e-150
            winit[-1] = 1e-150
        G = np.log(w0) - np.log(w0[-1])
        g0 = np.log(winit) - np.log(winit[-1])
        with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
            yT = np.ascontiguousarray(ctx.read_ytilde())
            for theta in (0.5, 20.0):
                f, g = ctx.logw_fdf(g0, G, theta)
                fr = R.logw_f(g0, G, yT, YTilde, theta)
                gr = R.logw_df(g0, G, yT, YTilde, theta)
                print("logw M=%d last_tiny=%s theta=%g: G range [%.0f, %.0f]  f rel %.2e  grad rel %.2e (f = %.6g)"
                      % (M, last_tiny, theta, G.min(), G.max(), rel(f, fr), np.abs(g - gr).max() / np.abs(gr).max(), fr), flush=True)
            for theta in (100.0, 31.6):
                t0 = time.time()
                go, wo, info = ctx.opt_lbfgs_logw(g0, G, theta, CONV)
                t1 = time.time()
                gr_, fmin_r, code_r = R.opt_lbfgs_logw(g0, G, yT, YTilde, theta, CONV)
                w_r, _ = R.get_weights(gr_)
                print("  lbfgs theta=%g: codes %d/%d fmin rel %.2e  w diff %.2e max(w)  (%d it, %.2fs / %.2fs)"
                      % (theta, info.lbfgs_code, code_r, rel(info.fmin, fmin_r), np.abs(wo - w_r).max() / w_r.max(),
                         info.iterations, t1 - t0, time.time() - t1), flush=True)

    # theta = 0 forces at configs[1] size
    M, N = 256, 100000
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    rng = np.random.default_rng(3)
    w0 = np.full(N, 1.0 / N)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        yT = np.ascontiguousarray(ctx.read_ytilde())
        for scale in (0.0, 1e-3, 1e-2):
            forces = scale * rng.standard_normal(M)
            f, g = ctx.forces_fdf(forces, w0, 0.0)
            fr = R.forces_f(forces, w0, yT, YTilde, 0.0)
            gr = R.forces_df(forces, w0, yT, YTilde, 0.0)
            print("theta=0 forces scale=%g: f rel %.2e grad rel %.2e (f = %.6g)" % (scale, rel(f, fr), np.abs(g - gr).max() / np.abs(gr).max(), fr), flush=True)
        for name, cfg in (("yaml", LBFGS_DEFAULTS), ("conv", dict(CONV, max_iterations=3000))):
            t0 = time.time()
            fo, wo, info = ctx.opt_lbfgs_forces(np.zeros(M), w0, 0.0, cfg)
            t1 = time.time()
            fr_, fmin_r, code_r = R.opt_lbfgs_forces(np.zeros(M), w0, yT, YTilde, 0.0, cfg)
            w_r = R.forces_weights(fr_, w0, yT)
            f_init = R.forces_f(np.zeros(M), w0, yT, YTilde, 0.0)
            print("  theta=0 lbfgs %s: codes %d/%d fmin %.6g / %.6g (f_init %.6g) w diff %.2e max(w) (%d it, %.2fs / %.2fs)"
                  % (name, info.lbfgs_code, code_r, info.fmin, fmin_r, f_init, np.abs(wo - w_r).max() / w_r.max(), info.iterations,
                     t1 - t0, time.time() - t1), flush=True)


if __name__ == "__main__":
    main()
