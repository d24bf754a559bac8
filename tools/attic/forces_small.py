"""Launch-bound forces problems: time per evaluation on small sizes (single theta and 4-theta batch)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bioen_amd
P = dict(linesearch=2, max_iterations=5000, delta=1e-6, epsilon=1e-6, ftol=1e-5, gtol=0.9, wolfe=0.9, past=10,
         max_linesearch=100)
rng = np.random.default_rng(12345)
for M, N in ((96, 3000), (256, 100000)):
    YTrue = rng.uniform(1, 10, M); sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
    YTilde = rng.normal(YTrue, sig_exp) / sig_exp
    w0 = np.full(N, 1.0 / N)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=1) as ctx:
        for thetas in ([10.0], [100.0, 10.0, 1.0, 0.3]):
            ctx.opt_lbfgs_forces_batch(thetas, np.zeros(M), w0, dict(P, max_iterations=3))
            best = None
            for rep in range(3):
                ctx.synchronize(); t0 = time.perf_counter()
                res, w, infos = ctx.opt_lbfgs_forces_batch(thetas, np.zeros(M), w0, P)
                ctx.synchronize(); dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            rounds = max(i.evaluations for i in infos)
            print("M=%d N=%d thetas=%d: %.1f ms, %d rounds -> %.1f us/round" % (M, N, len(thetas), 1e3 * best, rounds, 1e6 * best / rounds), flush=True)
