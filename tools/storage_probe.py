"""Reduced-byte storage EXPERIMENT of the log-weights matrix passes (Context.set_storage; VERDICT r03 item 7, SURVEY 7 /
8 f4): the theta sweep with the centred matrix streamed as FP64 (8 B, the graded path), fp32 + bf16 split (6 B) and fp32
(4 B) on ONE context -- sweep time, matrix-kernel times, and how far the minima and the weights move.
SIZE=M:N (default the headline 1024:1000000)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd                                                  # noqa: E402
from bioen_amd import sweep                                       # noqa: E402
from bench import synthetic_targets, LBFGS_DEFAULTS, SEED         # noqa: E402

M, N = (int(v) for v in os.environ.get("SIZE", "1024:1000000").split(":"))
params = LBFGS_DEFAULTS if os.environ.get("CONV", "0") != "1" else dict(LBFGS_DEFAULTS, epsilon=1e-9, delta=0.0, past=0, max_iterations=200000)
thetas = np.logspace(3, -0.5, 8)
YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M)
out = {"M": M, "N": N, "settings": "converged" if params is not LBFGS_DEFAULTS else "yaml defaults", "formats": {}}
with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED) as ctx:
    G = np.zeros(N)
    base = None
    for fmt, nbytes in (("f64", 8), ("split", 6), ("fp32", 4)):
        ctx.set_storage(fmt)
        sweep.sweep_log_weights(ctx, thetas, G, G, params)                       # builds the copies, warms up
        best, res = 1e30, None
        for rep in range(int(os.environ.get("REPS", "2"))):
            ctx.kernel_stats_enable(True)
            ctx.kernel_stats_reset()
            ctx.synchronize()
            t0 = time.perf_counter()
            res = sweep.sweep_log_weights(ctx, thetas, G, G, params)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            st = ctx.kernel_stats()
            ctx.kernel_stats_enable(False)
            best = min(best, dt)
        rec = {"bytes_per_element": nbytes, "sweep_s": best, "iterations": int(sum(r["iterations"] for r in res)),
               "evaluations": int(sum(r["evaluations"] for r in res)),
               "fwd_ms": st["forward"]["total_ms"] / max(st["forward"]["launches"], 1),
               "adj_ms": st["adjoint"]["total_ms"] / max(st["adjoint"]["launches"], 1),
               "rounds": st["forward"]["launches"], "codes": [r["code"] for r in res]}
        if base is None:
            base = res
        else:
            rec["fmin_rel_diff_vs_f64"] = [abs(a["fmin"] - b["fmin"]) / abs(b["fmin"]) for a, b in zip(res, base)]
            rec["w_diff_vs_f64_over_max_w"] = [float(np.abs(a["w"] - b["w"]).max() / b["w"].max()) for a, b in zip(res, base)]
            rec["speedup_vs_f64"] = out["formats"]["f64"]["sweep_s"] / best
        out["formats"][fmt] = rec
        print(fmt, json.dumps(rec), flush=True)
    ctx.set_storage("f64")
print(json.dumps(out))
