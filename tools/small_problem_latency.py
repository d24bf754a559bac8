"""Drop-in latency on the reference's own small fixtures (BASELINE configs[0]): wall time of optimize.log_weights.find_optimum
(lbfgs minimizer) per call -- first call on a matrix (context creation, upload, strip copies) and a repeated call (device
context reused) -- next to the reference's C + liblbfgs path on the same inputs (oracle/_ref, all granted cores)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import LOGW_GOLDEN, load_golden, LBFGS_DEFAULTS     # noqa: E402
import bioen_amd                                                  # noqa: E402
from bioen_amd import optimize                                    # noqa: E402
from bioen_amd.optimize.ext import c_bioen                        # noqa: E402
from oracle import ref_binding as R, cpus                         # noqa: E402

R.set_fast_openmp_flag(1)
R.omp_set_num_threads(cpus.usable_cpus())
params = optimize.minimize.Parameters("lbfgs")
params["verbose"] = False
for name in LOGW_GOLDEN:
    d = load_golden(name)
    YT = d["YTilde"].reshape(1, -1)
    theta = d["theta"] if d["theta"] > 0 else 1.0
    c_bioen.clear_cache()
    t = []
    for rep in range(4):
        t0 = time.perf_counter()
        try:
            out = optimize.log_weights.find_optimum(d["GInit"], d["G"], d["yTilde"], d["yTilde"], YT, theta, params)
            fmin = out[4]
        except RuntimeError as e:            # liblbfgs status outside {0, 1, 2}: the reference raises as well
            fmin = float("nan")
        t.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    g_ref, fmin_ref, code = R.opt_lbfgs_logw(d["GInit"], d["G"], d["yTilde"], d["YTilde"], theta, LBFGS_DEFAULTS)
    t_ref = time.perf_counter() - t0
    print("%-44s M x N = %4d x %5d: first call %7.2f ms, repeated %6.2f ms (min of 3); reference C path %7.2f ms (status %d); fmin %.6g / %.6g"
          % (name, d["yTilde"].shape[0], d["yTilde"].shape[1], 1e3 * t[0], 1e3 * min(t[1:]), 1e3 * t_ref, code, fmin, fmin_ref), flush=True)
