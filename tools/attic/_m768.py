import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bioen_amd
from bench import LBFGS_DEFAULTS
from canon_probe import targets
N = 1000000
def run(M, mode, Ks, stats):
    os.environ["BIOEN_HIP_ONE_COPY"] = mode
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    try:
        with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
            ctx.synchronize()
            G = np.zeros(N)
            for K in Ks:
                thetas = [float(t) for t in np.logspace(1, -0.5, K)]
                if stats:
                    ctx.kernel_stats_enable(True); ctx.kernel_stats_reset()
                r = ctx.opt_lbfgs_logw_batch(thetas, G, G, dict(LBFGS_DEFAULTS, max_iterations=3), max_batch=K)
                if stats:
                    ctx.kernel_stats()
            print(M, mode, Ks, stats, "ok", flush=True)
    except Exception as e:
        print(M, mode, Ks, stats, "FAILED", str(e)[:200], flush=True)
def run2(M, mode, iters, stats, fp):
    os.environ["BIOEN_HIP_ONE_COPY"] = mode
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    try:
        with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
            G = np.zeros(N)
            for K in (1, 4, 8):
                thetas = [float(t) for t in np.logspace(1, -0.5, K)]
                params = dict(LBFGS_DEFAULTS, max_iterations=iters, past=0, delta=0.0, epsilon=1e-12)
                ctx.opt_lbfgs_logw_batch(thetas, G, G, dict(params, max_iterations=2), max_batch=K)
                if stats:
                    ctx.kernel_stats_enable(True); ctx.kernel_stats_reset()
                ctx.opt_lbfgs_logw_batch(thetas, G, G, params, max_batch=K)
                if stats:
                    ctx.kernel_stats()
            if fp:
                ctx.footprint()
            print(M, mode, iters, stats, fp, "ok", flush=True)
    except Exception as e:
        print(M, mode, iters, stats, fp, "FAILED", str(e)[:160], flush=True)
which = sys.argv[1]
if which == "f":
    run2(1024, "0", 25, True, True); run2(1024, "1", 25, True, True); run2(768, "0", 25, True, True)
elif which == "g":
    run2(1024, "0", 25, False, False); run2(1024, "1", 25, False, False); run2(768, "0", 25, False, False)
elif which == "h":
    run2(1024, "0", 25, False, False); run2(1024, "0", 25, False, False); run2(768, "0", 25, False, False)
elif which == "i":
    run2(1024, "1", 25, False, False); run2(1024, "1", 25, False, False); run2(768, "0", 25, False, False)
elif which == "j":
    run2(768, "0", 25, False, False); run2(768, "0", 25, False, False); run2(768, "0", 25, False, False)
if which == "a":      # two-copy 1024 then 768
    run(1024, "0", (1, 4, 8), True); run(768, "0", (1,), False)
elif which == "b":    # one-copy 1024 then 768
    run(1024, "1", (1, 4, 8), True); run(768, "0", (1,), False)
elif which == "c":    # no stats
    run(1024, "0", (1, 4, 8), False); run(1024, "1", (1, 4, 8), False); run(768, "0", (1,), False)
elif which == "d":    # K=8 only
    run(1024, "0", (8,), False); run(768, "0", (1,), False)
elif which == "e":    # 1024 twice
    run(1024, "0", (1,), False); run(1024, "0", (1,), False); run(768, "0", (1,), False); run(1024, "0", (1,), False)
