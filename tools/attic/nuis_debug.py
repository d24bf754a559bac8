"""Debug: DEER nuisance series in both direction modes, per-step return codes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bioen_amd
from bioen_amd import nuisance
from test_hip_api import _deer_problem
from oracle import oracle_binding as O

F, sigma, groups, Y, m_true = _deer_problem()
Ft = (F - 1.0) / sigma[:, None]; YT = Y / sigma; off = 1.0 / sigma
N = F.shape[1]; G = np.zeros(N)
params = dict(linesearch=2, max_iterations=5000, delta=1e-6, epsilon=1e-6, ftol=1e-5, gtol=0.9, wolfe=0.9,
              past=10, max_linesearch=100)
for mode in ("twoloop", "gram"):
    with bioen_amd.Context(Ft, YT) as ctx:
        ctx.set_direction_mode(mode)
        scales = [0.15, 0.15]
        for theta in (1000.0, 100.0):
            for it in range(6):
                rs = np.ones(F.shape[0])
                for s, ix in zip(scales, groups): rs[ix] = s
                ctx.set_affine(off, rs)
                g, w, info = ctx.opt_lbfgs_logw(G, G, theta, params)
                explicit = off[:, None] + rs[:, None] * Ft
                go, fo, co, ito, evo = O.opt_lbfgs_logw(G, G, explicit, YT, theta, params)
                print(mode, theta, it, "code", info.lbfgs_code, "it", info.iterations, "ev", info.evaluations,
                      "fmin %.10g" % info.fmin, "| oracle code", co, "it", ito, "ev", evo, "fmin %.10g" % fo, flush=True)
                _, yraw = ctx.chi_squared(w)
                scales = nuisance.refit_scales(yraw, YT, off, groups)
