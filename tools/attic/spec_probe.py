"""Development aid: rejection rates per theta (evaluations vs iterations) and sweep time under the shadow policies.
SIZES="M:N,..."; prints per theta: iterations, evaluations, rejected share."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
from bioen_amd import sweep
from bench import synthetic_targets, LBFGS_DEFAULTS, SEED
sizes = [tuple(int(v) for v in s.split(":")) for s in os.environ.get("SIZES", "256:100000").split(",")]
thetas = np.logspace(3, -0.5, 8)
for (M, N) in sizes:
    YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED) as ctx:
        G = np.zeros(N)
        sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS)
        ctx.synchronize()
        t0 = time.perf_counter()
        res = sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print("M=%d N=%d sweep %.4f s; spec %s" % (M, N, dt, ctx.speculation_stats()))
        for r in res:
            print("   theta %8.3f it %5d ev %5d rejected %.3f code %d" % (r["theta"], r["iterations"], r["evaluations"],
                  1.0 - (r["iterations"] + 1.0) / r["evaluations"], r["code"]))
