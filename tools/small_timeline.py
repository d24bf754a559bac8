"""Development aid: wall time per lock-step round of the log-weights sweep on launch-bound sizes
(BASELINE configs[1] and the per-rank share of the headline at 8 GPUs), next to the matrix-kernel time.
Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split.  SIZES="M:N,M:N" overrides."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
from bioen_amd import sweep
from bench import synthetic_targets, LBFGS_DEFAULTS, SEED

sizes = [tuple(int(v) for v in s.split(":")) for s in os.environ.get("SIZES", "256:100000,1024:125000,64:20000").split(",")]
thetas = np.logspace(3, -0.5, 8)
if os.environ.get("THETAS"):                       # e.g. THETAS=10 for a K = 1 series
    thetas = np.array([float(v) for v in os.environ["THETAS"].split(",")])
for (M, N) in sizes:
    YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M)
    world = int(os.environ.get("WORLD", "1"))          # WORLD=8: rank 0 of an 8-rank decomposition, exchanges mirrored
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED, rank=0, world=world) as ctx:
        if world > 1:
            ctx.set_mirror_exchange(True)
        G = np.zeros(N)
        sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS)
        ctx.synchronize()
        t0 = time.perf_counter()
        res = sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS)       # wall time without the stats' event pairs
        ctx.synchronize()
        dt = time.perf_counter() - t0
        ctx.kernel_stats_enable(True)
        ctx.kernel_stats_reset()
        sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS)
        ctx.synchronize()
        ks = ctx.kernel_stats()
        ctx.kernel_stats_enable(False)
        panels = (M + 1023) // 1024 if (M > 1024 and os.environ.get("BIOEN_HIP_PANELS") != "0") else 1
        rounds = ks["forward"]["launches"] // panels       # beyond 1024 rows a pass is one launch per row panel
        mat = ks["forward"]["total_ms"] + ks["adjoint"]["total_ms"]
        print("M=%d N=%d: sweep %.4f s, %d iterations, %d rounds -> %.1f us/round, matrix kernels %.1f us/round (%.0f %%)" % (
            M, N, dt, sum(r["iterations"] for r in res), rounds, 1e6 * dt / rounds, 1e3 * mat / rounds, 100 * mat / (1e3 * dt)))
        sys.stdout.flush()
