"""Development aid: the headline series through ONE call of Context.opt_lbfgs_logw_batch -- wall time of the call against the
slowest theta's time inside the engine (what the deliveries, the page faults of fresh result arrays and the return path add);
BIOEN_HIP_PIN_RESULTS=0 for the A/B of the result-array registration."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bioen_amd
from bench import synthetic_targets, LBFGS_DEFAULTS, SEED
M, N = 1024, 1000000
thetas = np.logspace(3, -0.5, 8)
YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M)
with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED) as ctx:
    G = np.zeros(N)
    ctx.opt_lbfgs_logw_batch(thetas, G, G, LBFGS_DEFAULTS)
    for rep in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        res, w, infos = ctx.opt_lbfgs_logw_batch(thetas, G, G, LBFGS_DEFAULTS)
        t1 = time.perf_counter()
        res2, w2, infos2 = ctx.opt_lbfgs_logw_batch(thetas, G, G, LBFGS_DEFAULTS, want_weights=False)
        t2 = time.perf_counter()
        print("batch call %.4f s; slowest theta inside the engine %.4f s; without the weights handed back %.4f s" %
              (t1 - t0, max(i.seconds for i in infos), t2 - t1), flush=True)
