"""One strip copy against two (r05): the log-weights adjoint on the row-sum order copy (k_strip / k_strip2, ADJ form) against
k_strip_adj on the column-sum order copy -- per-launch time of the two matrix kernels at batch widths 1, 4, 8 and several M,
N = 1e6 (capped series: 25 iterations).  python tools/attic/onecopy_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bioen_amd                      # noqa: E402
from bench import LBFGS_DEFAULTS       # noqa: E402
from canon_probe import targets        # noqa: E402

N = int(os.environ.get("N", "1000000"))
for M in [int(m) for m in os.environ.get("MS", "1024,768,600,512,256").split(",")]:
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    for mode in ("0", "1"):
        os.environ["BIOEN_HIP_ONE_COPY"] = mode
        with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
            G = np.zeros(N)
            row = []
            for K in (1, 4, 8):
                thetas = [float(t) for t in np.logspace(1, -0.5, K)]
                params = dict(LBFGS_DEFAULTS, max_iterations=25, past=0, delta=0.0, epsilon=1e-12)
                ctx.opt_lbfgs_logw_batch(thetas, G, G, dict(params, max_iterations=2), max_batch=K)
                ctx.kernel_stats_enable(True)
                ctx.kernel_stats_reset()
                ctx.opt_lbfgs_logw_batch(thetas, G, G, params, max_batch=K)
                st = ctx.kernel_stats()
                row.append("K=%d fwd %.4f adj %.4f ms" % (K, st["forward"]["total_ms"] / max(st["forward"]["launches"], 1),
                                                          st["adjoint"]["total_ms"] / max(st["adjoint"]["launches"], 1)))
            forms, nbytes = ctx.footprint()
            print("M=%d N=%d %s (%s, %.1f GB): %s" % (M, N, "ONE copy " if mode == "1" else "two copies", "+".join(sorted(forms)),
                                                    nbytes / 1e9, "  ".join(row)), flush=True)
