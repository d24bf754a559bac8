"""Development aid: the headline sweep with the N-vector cache policy off / on (BIOEN_HIP_NVEC_NT=0 against the default),
printing the sweep time and the mean forward / adjoint matrix-kernel times -- run in alternating processes on one box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bioen_amd
from bioen_amd import sweep
from bench import synthetic_targets, LBFGS_DEFAULTS, SEED
M, N = 1024, 1000000
thetas = np.logspace(3, -0.5, 8)
YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M)
with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED) as ctx:
    G = np.zeros(N)
    sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS)
    for rep in range(2):
        ctx.kernel_stats_enable(True); ctx.kernel_stats_reset(); ctx.synchronize()
        t0 = time.perf_counter()
        sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        st = ctx.kernel_stats()
        print("NVEC_NT=%s sweep %.4f s fwd %.1f us adj %.1f us" % (os.environ.get("BIOEN_HIP_NVEC_NT", "auto"), dt,
              1e3 * st["forward"]["total_ms"] / st["forward"]["launches"], 1e3 * st["adjoint"]["total_ms"] / st["adjoint"]["launches"]), flush=True)
