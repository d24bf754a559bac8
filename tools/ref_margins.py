"""Measured parity margins against the REFERENCE's own binary (oracle/_ref, travels with the repository) at sizes no
golden fixture reaches -- the numbers behind tests/test_hip_fullsize.py::test_configs1_* and the ala5-shape test, as a
markdown table (committed as profiles/rNN_parity_margins.md).  Both sides run the same settings; `fmin rel` is
|L_dev - L_ref| / |L_ref| (north_star: 1e-6), `w` is max|w_dev - w_ref| / max(w_ref) (north_star: 1e-5).
usage (GPU box): python tools/ref_margins.py > gpurun_out/parity_margins.md"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bioen_amd
from oracle import ref_binding as R
from oracle import cpus
from test_hip_fullsize import _targets, LBFGS_DEFAULTS
from bench import ALA5_LBFGS

R.set_fast_openmp_flag(0)          # serial sums: the same reference result on every run and box
R.omp_set_num_threads(cpus.usable_cpus())
CONV = dict(LBFGS_DEFAULTS, epsilon=1e-9, delta=0.0, past=0, max_iterations=200000)

print("# Parity margins against the reference binary (converged settings unless noted)\n")
print("| method | prior | M x N | settings | theta | status ref / device | fmin rel | w | device iterations |")
print("|---|---|---|---|---|---|---|---|---|")
for prior, M, N, eps in [("uniform", 256, 100000, 1e-9), ("random", 256, 100000, 1e-9), ("random", 256, 100000, 1e-10),
                         ("uniform", 1024, 20000, 1e-9), ("uniform", 205, 50000, 1e-9)]:
    conv = dict(CONV, epsilon=eps)
    thetas = [316.0, 100.0, 31.6] if (M, N) == (256, 100000) else [316.0, 100.0]
    YTrue, sig_sim, sig_exp, YTilde = _targets(M)
    if prior == "uniform":
        G = np.zeros(N)
        g0 = G
    else:
        rng = np.random.default_rng(99)
        G = np.log(rng.gamma(2.0, 1.0, N))
        G -= G.max()
        g0 = G + 0.3 * rng.standard_normal(N)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        res, w, infos = ctx.opt_lbfgs_logw_batch(thetas, g0, G, conv)
        yT = np.ascontiguousarray(ctx.read_ytilde())
    for k, th in enumerate(thetas):
        g_ref, fmin_ref, code = R.opt_lbfgs_logw(g0, G, yT, YTilde, th, conv)
        w_ref = np.asarray(R.get_weights(g_ref)[0]).ravel()
        print("| log-weights | %s | %d x %d | epsilon %g, delta 0 | %g | %d / %d | %.2e | %.2e | %d |" % (
            prior, M, N, eps, th, code, infos[k].lbfgs_code, abs(infos[k].fmin - fmin_ref) / abs(fmin_ref),
            np.abs(w[k] - w_ref).max() / w_ref.max(), infos[k].iterations))
    sys.stdout.flush()
for M, N in [(256, 100000), (512, 50000), (96, 30000), (1024, 20000), (600, 30000)]:
    thetas = [316.0, 100.0, 31.6]
    YTrue, sig_sim, sig_exp, YTilde = _targets(M)
    w0 = np.full(N, 1.0 / N)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        res, w, infos = ctx.opt_lbfgs_forces_batch(thetas, np.zeros(M), w0, CONV)
        yT = np.ascontiguousarray(ctx.read_ytilde())
    for k, th in enumerate(thetas):
        f_ref, fmin_ref, code = R.opt_lbfgs_forces(np.zeros(M), w0, yT, YTilde, th, CONV)
        w_ref = np.asarray(R.forces_weights(f_ref, w0, yT)).ravel()
        print("| forces | uniform | %d x %d | epsilon 1e-09, delta 0 | %g | %d / %d | %.2e | %.2e | %d |" % (
            M, N, th, code, infos[k].lbfgs_code, abs(infos[k].fmin - fmin_ref) / abs(fmin_ref),
            np.abs(w[k] - w_ref).max() / w_ref.max(), infos[k].iterations))
    sys.stdout.flush()
# the ala5 notebook's shape and protocol (warm-started chain, lbfgs_2.yaml settings)
N, M = 50001, 28
YTrue, sig_sim, sig_exp, YTilde = _targets(M)
w0 = np.full(N, 1.0 / N)
with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
    yT = np.ascontiguousarray(ctx.read_ytilde())
    f_dev, f_ref = np.zeros(M), np.zeros(M)
    for th in np.logspace(5, -1, 80)[::8]:
        f_dev, w_dev, info = ctx.opt_lbfgs_forces(f_dev, w0, th, ALA5_LBFGS)
        f_ref, fmin_ref, code = R.opt_lbfgs_forces(f_ref, w0, yT, YTilde, th, ALA5_LBFGS)
        w_ref = np.asarray(R.forces_weights(f_ref, w0, yT)).ravel()
        print("| forces (ala5 shape, warm chain) | uniform | %d x %d | lbfgs_2.yaml | %.4g | %d / %d | %.2e | %.2e | %d |" % (
            M, N, th, code, info.lbfgs_code, abs(info.fmin - fmin_ref) / abs(fmin_ref),
            np.abs(w_dev - w_ref).max() / w_ref.max(), info.iterations))
