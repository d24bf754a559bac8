"""Development aid (GPU box): every line search (0 More-Thuente, 1 Armijo, 2 Wolfe, 3 strong Wolfe) on clean, non-finite and
degenerate inputs -- status / counts / fmin of the device beside the reference's binary."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
from oracle import ref_binding as R
from bench import LBFGS_DEFAULTS

rng = np.random.default_rng(3)
M, N = 64, 4000
YTrue = rng.uniform(1, 10, M)
y = rng.normal(YTrue[:, None], 0.5 * YTrue[:, None], (M, N)) / (0.1 * YTrue[:, None])
YT = rng.normal(YTrue, 0.1 * YTrue) / (0.1 * YTrue)
G, w0, f0 = np.zeros(N), np.full(N, 1.0 / N), np.zeros(M)
nan = float("nan")
g_nan = G.copy(); g_nan[7] = nan
same = y.copy(); same[:, :] = same[:, :1]
cases = [("clean", y, G, 10.0), ("NaN in g0", y, g_nan, 10.0), ("identical structures", same, G, 1.0), ("theta = 0", y, G, 0.0),
         ("theta = 1e12", y, G, 1e12)]
for ls in (0, 1, 2, 3):
    params = dict(LBFGS_DEFAULTS, linesearch=ls, max_iterations=300)
    for tag, yy, g0, theta in cases:
        for forces in (False, True):
            with bioen_amd.Context(yy, YT) as ctx:
                if forces:
                    start = f0.copy()
                    if "NaN" in tag:
                        start[2] = nan
                    x, w, info = ctx.opt_lbfgs_forces(start, w0, theta, params)
                    xr, fr, cr = R.opt_lbfgs_forces(start, w0, yy, YT, theta, params)
                else:
                    x, w, info = ctx.opt_lbfgs_logw(g0, G, theta, params)
                    xr, fr, cr = R.opt_lbfgs_logw(g0, G, yy, YT, theta, params)
            same_code = info.lbfgs_code == cr
            relf = abs(info.fmin - fr) / max(abs(fr), 1e-300) if np.isfinite(fr) and np.isfinite(info.fmin) else float("nan")
            print("linesearch %d %-7s %-22s device (%5d, %3d it, %3d ev, %.12g) | reference (%5d, %.12g) | %s rel %.1e" % (
                ls, "forces" if forces else "logw", tag, info.lbfgs_code, info.iterations, info.evaluations, info.fmin, cr, fr,
                "same status" if same_code else "STATUS DIFFERS", relf))
            sys.stdout.flush()
