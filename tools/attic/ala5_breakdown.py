"""Development aid: where the ala5-shaped warm-started forces series (bench.ala5_record) spends its time: per call wall,
evaluations, rounds, matrix-kernel time; with and without the weights handed back."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bioen_amd
from bench import synthetic_targets, ALA5_LBFGS, SEED

N, M = 50001, 28
thetas = np.logspace(5, -1, 80)
YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M, SEED)
w0 = np.full(N, 1.0 / N)
with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED) as ctx:
    for want in (True, False, True):
        for rep in range(2):
            f = np.zeros(M)
            ctx.kernel_stats_enable(rep == 1); ctx.kernel_stats_reset()
            its = evs = 0
            per = []
            ctx.synchronize()
            t0 = time.perf_counter()
            for th in thetas:
                t1 = time.perf_counter()
                f, w, info = ctx.opt_lbfgs_forces(f, w0, th, ALA5_LBFGS, want_weights=want)
                per.append(time.perf_counter() - t1)
                its += info.iterations; evs += info.evaluations
            dt = time.perf_counter() - t0
            if rep == 1:
                ks = ctx.kernel_stats()
                mat = ks["forward"]["total_ms"] + ks["adjoint"]["total_ms"]
                print("want_weights=%s (stats on): %.1f ms, %d iterations, %d evaluations, %d matrix launches, matrix kernels %.2f ms" % (
                    want, 1e3 * dt, its, evs, ks["forward"]["launches"] + ks["adjoint"]["launches"], mat))
            else:
                per = np.array(per)
                print("want_weights=%s: %.1f ms for 80 calls (%.0f us per call, min %.0f, max %.0f), %d iterations, %d evaluations -> %.0f us per evaluation" % (
                    want, 1e3 * dt, 1e6 * per.mean(), 1e6 * per.min(), 1e6 * per.max(), its, evs, 1e6 * dt / evs))
