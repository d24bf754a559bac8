"""Device GSL-style minimizers vs the oracle on every golden fixture (diagnostic)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bioen_amd
from conftest import LOGW_GOLDEN, FORCES_GOLDEN, load_golden
from oracle import oracle_binding as O
P = dict(step_size=0.01, tol=0.001, max_iterations=5000)
for name in LOGW_GOLDEN + FORCES_GOLDEN:
    d = load_golden(name)
    forces = "forces_init" in d
    with bioen_amd.Context(d["yTilde"], d["YTilde"]) as ctx:
        for alg in O.GSL_ALGORITHMS:
            t = time.time()
            if forces:
                x, w, info = ctx.opt_gsl_forces(d["forces_init"], d["w0"], d["theta"], alg, P)
                xo, fo, so, ito, evo = O.opt_gsl_forces(d["forces_init"], d["w0"], d["yTilde"], d["YTilde"], d["theta"], dict(P, algorithm=alg))
            else:
                x, w, info = ctx.opt_gsl_logw(d["GInit"], d["G"], d["theta"], alg, P)
                xo, fo, so, ito, evo = O.opt_gsl_logw(d["GInit"], d["G"], d["yTilde"], d["YTilde"], d["theta"], dict(P, algorithm=alg))
            dt = time.time() - t
            print("%-42s %-17s dev (%3d,%5d,%5d) %.12g | oracle (%3d,%5d,%5d) %.12g | rel %.1e dx %.1e  %.2fs" % (
                name, alg, info.lbfgs_code, info.iterations, info.evaluations, info.fmin, so, ito, sum(evo), fo,
                abs(info.fmin - fo) / abs(fo), np.abs(x - xo).max(), dt), flush=True)
