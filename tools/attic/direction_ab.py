"""Development aid: A/B the two direction modes on the bench workload in ONE process."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
from bench import synthetic_targets, LBFGS_DEFAULTS, SEED
M, N = 1024, 1000000
thetas = np.logspace(3, -0.5, 8)
YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M)
ctx = bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED)
G = np.zeros(N)
for rep in range(2):
    for mode in ("twoloop", "gram"):
        ctx.set_direction_mode(mode)
        t0 = time.perf_counter()
        res, w, infos = ctx.opt_lbfgs_logw_batch(thetas, G, G, LBFGS_DEFAULTS, max_batch=8)
        dt = time.perf_counter() - t0
        print("%-8s sweep %.3f s  iterations %d  evaluations %d  fmin %s" % (
            mode, dt, sum(i.iterations for i in infos), sum(i.evaluations for i in infos),
            " ".join("%.6f" % i.fmin for i in infos)))
        sys.stdout.flush()
ctx.close()
