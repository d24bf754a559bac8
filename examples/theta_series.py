#!/usr/bin/env python3
"""A theta series on one MI355X through the bioen.optimize-compatible API and through the
batched device interface (needs the built library and a GPU; there is no CPU path).

    make -C bioen_amd/csrc && python examples/theta_series.py [N] [M]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd                                   # noqa: E402
from bioen_amd import optimize                     # noqa: E402  (drop-in for `from bioen import optimize`)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 64

# synthetic ensemble after bioen/optimize/forces.py:19-68
rng = np.random.default_rng(12345)
YTrue = rng.uniform(1, 10, M)
sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
y = rng.normal(YTrue[:, None], sig_sim[:, None], (M, N))
yTilde = y / sig_exp[:, None]
YTilde = (rng.normal(YTrue, sig_exp) / sig_exp)[None, :]
w0 = np.full((N, 1), 1.0 / N)
G = np.log(w0)
thetas = np.logspace(3, -0.5, 8)

# 1. the reference's API, one theta at a time (the matrix is uploaded once and stays resident)
params = optimize.minimize.Parameters("lbfgs")
params["verbose"] = False
t0 = time.perf_counter()
for theta in thetas:
    wopt, yopt, gopt, f0, fmin = optimize.log_weights.find_optimum(G.copy(), G, y, yTilde, YTilde, theta, params)
    print("theta %8.3f  L %.6f -> %.6f   S = %.4f" % (theta, f0, fmin, -float(np.sum(wopt * np.log(wopt / w0)))))
print("find_optimum x %d: %.3f s" % (len(thetas), time.perf_counter() - t0))

# 2. the whole series as one lock-step batch: all thetas share every pass over yTilde
with bioen_amd.Context(yTilde, YTilde.ravel()) as ctx:
    t0 = time.perf_counter()
    g_all, w_all, infos = ctx.opt_lbfgs_logw_batch(thetas, G.ravel(), G.ravel(), params["params"])
    dt = time.perf_counter() - t0
    for theta, info in zip(thetas, infos):
        print("theta %8.3f  L = %.6f  chi2/2 = %.4f  S = %.4f  (%d iterations, status %d)"
              % (theta, info.fmin, info.chi2, -info.kl, info.iterations, info.lbfgs_code))
    print("batched series: %.3f s" % dt)
    # the forces method on the same resident matrix
    f_all, wf_all, finfos = ctx.opt_lbfgs_forces_batch(thetas[:4], np.zeros(M), w0.ravel(), params["params"])
    print("forces, 4 thetas: L =", ["%.5f" % i.fmin for i in finfos])
