"""Development aid: the forces theta series of BASELINE configs[4] (N = 1e6 x M = 512, 8 thetas, one lock-step batch), twice;
run it under `rocprofv3 --kernel-trace` and feed the trace to tools/trace_gaps.py with "true, true" (the pass-1 strip
kernel) as the first kernel of a round.  SIZE="M:N" overrides."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
from bench import synthetic_targets, LBFGS_DEFAULTS, SEED

M, N = (int(v) for v in os.environ.get("SIZE", "512:1000000").split(":"))
thetas = np.logspace(3, -0.5, 8)
if os.environ.get("THETAS"):
    thetas = np.array([float(v) for v in os.environ["THETAS"].split(",")])
YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M)
w0 = np.full(N, 1.0 / N)
f0 = np.zeros(M)
with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED) as ctx:
    ctx.opt_lbfgs_forces_batch(thetas[:2], f0, w0, dict(LBFGS_DEFAULTS, max_iterations=3))
    for rep in range(int(os.environ.get("REPS", "2"))):
        ctx.synchronize()
        t0 = time.perf_counter()
        res, w, infos = ctx.opt_lbfgs_forces_batch(thetas, f0, w0, LBFGS_DEFAULTS)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        ev = max(i.evaluations for i in infos)
        print("M=%d N=%d: series %.4f s, %d iterations, longest chain %d evaluations -> <= %.1f us per round" % (
            M, N, dt, sum(i.iterations for i in infos), ev, 1e6 * dt / ev))
        sys.stdout.flush()
