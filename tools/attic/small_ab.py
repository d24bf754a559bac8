"""Development aid: per-iteration cost on launch-bound sizes (BASELINE config 2), both direction modes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
from bench import synthetic_targets, LBFGS_DEFAULTS, SEED
for (M, N) in ((256, 100000), (64, 20000)):
    YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M)
    ctx = bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED)
    G = np.zeros(N)
    for rep in range(2):
        for mode in ("twoloop", "gram"):
            ctx.set_direction_mode(mode)
            t0 = time.perf_counter()
            g, w, info = ctx.opt_lbfgs_logw(G, G, 10.0, LBFGS_DEFAULTS)
            dt = time.perf_counter() - t0
            print("M=%d N=%d %-8s %.3f s  %d it %d ev -> %.1f us/iteration  fmin %.8f (in-library %.3f s)" % (
                M, N, mode, dt, info.iterations, info.evaluations, 1e6 * info.seconds / info.iterations, info.fmin, info.seconds))
            sys.stdout.flush()
    ctx.close()
