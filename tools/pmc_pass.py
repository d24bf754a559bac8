"""The program bench.py runs under `rocprofv3 --pmc <counter>` (one counter per pass, as MI355X_MICROARCH.md prescribes for
HBM traffic) to read the dominant kernel's bytes per launch on the box the bench line comes from: the headline matrix,
a few lock-step rounds of the theta series, nothing else.  usage: python3 tools/pmc_pass.py <logw|forces> <M> <N>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
from bench import synthetic_targets, LBFGS_DEFAULTS, SEED

method, M, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
thetas = np.logspace(3, -0.5, 8)
YTrue, sig_sim, sig_exp, YTilde = synthetic_targets(M)
few = dict(LBFGS_DEFAULTS, max_iterations=4)          # -997 (iteration cap) is the expected end of every problem
with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=SEED) as ctx:
    if method == "logw":
        G = np.zeros(N)
        ctx.opt_lbfgs_logw_batch(thetas, G, G, few, want_weights=False)
    else:
        ctx.opt_lbfgs_forces_batch(thetas, np.zeros(M), np.full(N, 1.0 / N), few, want_weights=False)
    ctx.synchronize()
print("pmc_pass done")
