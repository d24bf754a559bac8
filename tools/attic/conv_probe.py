"""How many iterations does the headline problem (N = 1e6 x M = 1024) need under converged settings?  (r05: sizing of
bench.py's full-size parity record.)  python tools/attic/conv_probe.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import bioen_amd                                  # noqa: E402
from conftest import LBFGS_DEFAULTS                # noqa: E402
from canon_probe import targets                    # noqa: E402

M, N = 1024, 1000000
YTrue, sig_sim, sig_exp, YTilde = targets(M)
with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
    G = np.zeros(N)
    for theta in (100.0, 31.622776601683793, 10.0):
        for eps in (1e-7, 1e-8, 1e-9):
            conv = dict(LBFGS_DEFAULTS, epsilon=eps, delta=0.0, past=0, max_iterations=12000)
            t0 = time.perf_counter()
            g, w, i = ctx.opt_lbfgs_logw(G, G, theta, conv)
            dt = time.perf_counter() - t0
            print("theta %g epsilon %g: code %d, %d iterations, %d evaluations, fmin %.15g, %.1f s" %
                  (theta, eps, i.lbfgs_code, i.iterations, i.evaluations, i.fmin, dt), flush=True)
            if i.lbfgs_code == -997:
                break
