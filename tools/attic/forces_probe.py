"""BASELINE config 5 on one GPU: forces method, N = 1e6 x M = 512, 8-theta series as one lock-step
batch (cold starts), yaml-default liblbfgs; prints wall time, rounds and matrix-kernel statistics."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bioen_amd

N, M = 1000000, int(os.environ.get("FORCES_M", "512"))
P = dict(linesearch=2, max_iterations=5000, delta=1e-6, epsilon=1e-6, ftol=1e-5, gtol=0.9, wolfe=0.9, past=10,
         max_linesearch=100)
rng = np.random.default_rng(12345)
YTrue = rng.uniform(1, 10, M)
sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
YTilde = rng.normal(YTrue, sig_exp) / sig_exp
thetas = np.logspace(3, -0.5, 8)
w0 = np.full(N, 1.0 / N)
with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
    for batch in (8,):
        ctx.opt_lbfgs_forces_batch(thetas[:2], np.zeros(M), w0, dict(P, max_iterations=3), max_batch=batch)  # warm
        ctx.kernel_stats_enable(True); ctx.kernel_stats_reset(); ctx.synchronize()
        t0 = time.perf_counter()
        res, w, infos = ctx.opt_lbfgs_forces_batch(thetas, np.zeros(M), w0, P, max_batch=batch)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        st = ctx.kernel_stats()
        its = sum(i.iterations for i in infos); evs = sum(i.evaluations for i in infos)
        print(json.dumps({"max_batch": batch, "wall_s": dt, "iterations": its, "evaluations": evs,
                          "codes": [i.lbfgs_code for i in infos], "per_theta_iterations": [i.iterations for i in infos],
                          "per_theta_evaluations": [i.evaluations for i in infos],
                          "fmin": [i.fmin for i in infos], "iterNM_per_s": its * float(N) * M / dt,
                          "forward": st["forward"], "adjoint": st["adjoint"]}), flush=True)
