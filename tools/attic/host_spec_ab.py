"""Development aid: the host-driven engine's shadow policy at the headline size (one process, interleaved):
slots kept back for shadows of the slowest theta (BIOEN_HIP_RESERVE) x shadows per round (BIOEN_HIP_HOST_SHADOWS)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
from bioen_amd import sweep
from bench import synthetic_targets, LBFGS_DEFAULTS, SEED
M, N = [int(v) for v in os.environ.get("SIZE", "1024:1000000").split(":")]
thetas = np.logspace(3, -0.5, 8)
YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M)
variants = [("reserve %s, <= %s shadows" % (r, n), {"BIOEN_HIP_DEVICE_LS": "0", "BIOEN_HIP_HOST_SHADOWS": str(n), "BIOEN_HIP_RESERVE": str(r)})
            for r, n in ((0, 8), (2, 8), (2, 2), (2, 3), (2, 4), (3, 3), (0, 2), (0, 0))]
with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED) as ctx:
    G = np.zeros(N)
    best = {}
    for rep in range(int(os.environ.get("REPS", "3")) + 1):
        for name, env in variants:
            os.environ.update(env)
            ctx.kernel_stats_enable(True); ctx.kernel_stats_reset(); ctx.synchronize()
            t0 = time.perf_counter()
            sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            rounds = ctx.kernel_stats()["forward"]["launches"]
            if rep:
                best[name] = min(best.get(name, (9e9, 0)), (dt, rounds))
    for name, _ in variants:
        print("M=%d N=%d %-28s best %.4f s, %d rounds, %.1f us/round" % (M, N, name, best[name][0], best[name][1], 1e6 * best[name][0] / best[name][1]))
