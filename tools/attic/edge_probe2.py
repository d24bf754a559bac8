"""Second probe for tests/test_hip_edgecases.py: (i) forces with a cluster of near-identical structures on top and a bulk
that underflows (well-conditioned gradient although some w_j < DBL_MIN), (ii) theta = 0 forces at configs[1] size under
iteration caps, (iii) analyze-style log-weights at a size the reference converges on in seconds."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bioen_amd                                   # noqa: E402
from oracle import ref_binding as R, cpus          # noqa: E402
from tools.edge_probe import targets, rel, LBFGS_DEFAULTS, CONV   # noqa: E402


def cluster_problem(M, N, seed, ncluster=200, sigma_x=30.0):
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    rng = np.random.default_rng(seed)
    y = rng.normal(YTrue[:, None], sig_sim[:, None], (M, N)) / sig_exp[:, None]
    y[:, :ncluster] = y[:, :1] + 0.002 * rng.standard_normal((M, ncluster)) * (sig_sim / sig_exp)[:, None]
    mean = y[:, ncluster:].mean(axis=1)
    f = y[:, 0] - mean
    x = f @ y
    f *= sigma_x / x[ncluster:].std()
    w0 = rng.dirichlet(np.ones(N) * 2.0)
    w0[rng.choice(N, 50, replace=False)] = 0.0
    w0[rng.choice(N, 50, replace=False)] = 1e-310
    w0[rng.choice(N, 5, replace=False)] = 4.9e-324
    w0[3] = 0.0                  # inside the cluster too
    w0[5] = 1e-310
    return y, YTilde, f, w0


def main():
    R.set_fast_openmp_flag(0)
    R.omp_set_num_threads(cpus.usable_cpus())
    for M, N in ((96, 20000), (512, 20000), (600, 20000), (1056, 12000)):
        y, YTilde, f, w0 = cluster_problem(M, N, M)
        with bioen_amd.Context(y, YTilde) as ctx:
            for theta in (0.0, 3.0):
                fd, gd = ctx.forces_fdf(f, w0, theta)
                w_ref = R.forces_weights(f, w0, y)
                fr = R.forces_f(f, w0, y, YTilde, theta)
                gr = R.forces_df(f, w0, y, YTilde, theta)
                ws = np.sort(w_ref)[::-1]
                print("cluster M=%d theta=%g: zeros %d denormals %d top w %.3g %.3g w[200] %.3g | f rel %.2e grad rel %.2e (|grad| %.3g, f %.6g)"
                      % (M, theta, (w_ref == 0).sum(), ((w_ref > 0) & (w_ref < 2.3e-308)).sum(), ws[0], ws[1], ws[200],
                         rel(fd, fr), np.abs(gd - gr).max() / np.abs(gr).max(), np.abs(gr).max(), fr), flush=True)
            # batch of 8: scaled copies of the force vector
            F = np.stack([f * s for s in (1.0, 0.9, 0.8, 0.5, 0.25, 1.05, 0.0, 0.6)])
            th = np.array([0.0, 3.0, 30.0, 0.5, 10.0, 1.0, 5.0, 100.0])
            fb, gb = ctx.forces_fdf_batch(F, w0, th)
            worst_f, worst_g = 0.0, 0.0
            for a in range(8):
                fr = R.forces_f(F[a], w0, y, YTilde, th[a])
                gr = R.forces_df(F[a], w0, y, YTilde, th[a])
                worst_f = max(worst_f, rel(fb[a], fr))
                worst_g = max(worst_g, np.abs(gb[a] - gr).max() / np.abs(gr).max())
            print("   K=8 batch: worst f rel %.2e  worst grad rel %.2e" % (worst_f, worst_g), flush=True)

    # theta = 0 forces at configs[1] size: iteration caps
    M, N = 256, 100000
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    w0 = np.full(N, 1.0 / N)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        yT = np.ascontiguousarray(ctx.read_ytilde())
        f_init = R.forces_f(np.zeros(M), w0, yT, YTilde, 0.0)
        for cap in (30, 100, 300):
            cfg = dict(LBFGS_DEFAULTS, max_iterations=cap, delta=0.0, past=0, epsilon=1e-12)
            fo, wo, info = ctx.opt_lbfgs_forces(np.zeros(M), w0, 0.0, cfg)
            t1 = time.time()
            fr_, fmin_r, code_r = R.opt_lbfgs_forces(np.zeros(M), w0, yT, YTilde, 0.0, cfg)
            w_r = R.forces_weights(fr_, w0, yT)
            print("theta=0 cap %d: codes %d/%d fmin %.10g / %.10g rel %.2e (f_init %.6g) w diff %.2e max(w) forces diff %.2e (ref %.1fs)"
                  % (cap, info.lbfgs_code, code_r, info.fmin, fmin_r, rel(info.fmin, fmin_r), f_init,
                     np.abs(wo - w_r).max() / w_r.max(), np.abs(fo - fr_).max() / np.abs(fr_).max(), time.time() - t1), flush=True)

    # analyze-style log-weights, M = 64 x N = 4000 (pinned by the reference's two line searches at 3e-8 / 2e-6)
    M, N = 64, 4000
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    for last in (True, False):
        rng = np.random.default_rng(M + N)
        y = rng.normal(YTrue[:, None], sig_sim[:, None], (M, N)) / sig_exp[:, None]
        w0 = rng.dirichlet(np.ones(N) * 2.0)
        w0[rng.choice(N - 1, N // 20, replace=False)] = 1e-150
        wi = rng.dirichlet(np.ones(N) * 2.0)
        wi[rng.choice(N - 1, N // 20, replace=False)] = 1e-150
        if last:
            w0[-1] = 1e-150
            wi[-1] = 1e-150
        G = np.log(w0) - np.log(w0[-1])
        g0 = np.log(wi) - np.log(wi[-1])
        cfg = dict(CONV, epsilon=1e-12)
        with bioen_amd.Context(y, YTilde) as ctx:
            for theta in ((100.0, 10.0) if last else (10.0,)):
                go, wo, info = ctx.opt_lbfgs_logw(g0, G, theta, cfg)
                gr_, fmin_r, code_r = R.opt_lbfgs_logw(g0, G, y, YTilde, theta, cfg)
                w_r, _ = R.get_weights(gr_)
                print("analyze-style logw last=%s theta=%g: codes %d/%d fmin rel %.2e w diff %.2e max(w) (%d it)"
                      % (last, theta, info.lbfgs_code, code_r, rel(info.fmin, fmin_r), np.abs(wo - w_r).max() / w_r.max(),
                         info.iterations), flush=True)


if __name__ == "__main__":
    main()
