"""Development aid (GPU box): non-finite inputs -- what the reference's binary and the device return, and how long they take.
The reference does not validate its inputs (c_bioen.pyx:274-290 reads .data as it is); liblbfgs compares with NaN."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
from oracle import ref_binding as R
from bench import LBFGS_DEFAULTS

M, N = 64, 4000
rng = np.random.default_rng(3)
YTrue = rng.uniform(1, 10, M)
sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
y = (rng.normal(YTrue[:, None], sig_sim[:, None], (M, N)) / sig_exp[:, None])
YT = rng.normal(YTrue, sig_exp) / sig_exp
G = np.zeros(N)
params = dict(LBFGS_DEFAULTS, max_iterations=200)


def run_logw(y_, YT_, g0, G_, theta, tag):
    t0 = time.perf_counter()
    try:
        with bioen_amd.Context(y_, YT_) as ctx:
            res, w, info = ctx.opt_lbfgs_logw(g0, G_, theta, params)
        dev = (info.lbfgs_code, info.iterations, info.evaluations, info.fmin, bool(np.isfinite(res).all()))
    except Exception as e:
        dev = ("exception", repr(e)[:120])
    t1 = time.perf_counter()
    try:
        out = R.opt_lbfgs_logw(g0, G_, y_, YT_, theta, params)
        ref = (out[2], out[1], bool(np.isfinite(np.asarray(out[0])).all()))
    except Exception as e:
        ref = ("exception", repr(e)[:120])
    t2 = time.perf_counter()
    print("%-34s device %s (%.3f s) | reference (code, fmin, finite) %s (%.3f s)" % (tag, dev, t1 - t0, ref, t2 - t1))
    sys.stdout.flush()


def run_forces(y_, YT_, f0, w0, theta, tag):
    t0 = time.perf_counter()
    try:
        with bioen_amd.Context(y_, YT_) as ctx:
            res, w, info = ctx.opt_lbfgs_forces(f0, w0, theta, params)
        dev = (info.lbfgs_code, info.iterations, info.evaluations, info.fmin, bool(np.isfinite(res).all()))
    except Exception as e:
        dev = ("exception", repr(e)[:120])
    t1 = time.perf_counter()
    try:
        out = R.opt_lbfgs_forces(f0, w0, y_, YT_, theta, params)
        ref = (out[2], out[1], bool(np.isfinite(np.asarray(out[0])).all()))
    except Exception as e:
        ref = ("exception", repr(e)[:120])
    t2 = time.perf_counter()
    print("%-34s device %s (%.3f s) | reference (code, fmin, finite) %s (%.3f s)" % (tag, dev, t1 - t0, ref, t2 - t1))
    sys.stdout.flush()


nan, inf = float("nan"), float("inf")
g_nan = G.copy(); g_nan[7] = nan
g_inf = G.copy(); g_inf[7] = inf
g_minf = G.copy(); g_minf[7] = -inf
y_nan = y.copy(); y_nan[3, 11] = nan
y_inf = y.copy(); y_inf[3, 11] = inf
YT_nan = YT.copy(); YT_nan[5] = nan
run_logw(y, YT, G, G, 10.0, "logw clean")
run_logw(y, YT, g_nan, G, 10.0, "logw NaN in g0")
run_logw(y, YT, g_inf, G, 10.0, "logw +inf in g0")
run_logw(y, YT, g_minf, G, 10.0, "logw -inf in g0")
run_logw(y, YT, G, g_nan, 10.0, "logw NaN in G")
run_logw(y_nan, YT, G, G, 10.0, "logw NaN in yTilde")
run_logw(y_inf, YT, G, G, 10.0, "logw inf in yTilde")
run_logw(y, YT_nan, G, G, 10.0, "logw NaN in YTilde")
run_logw(y, YT, G, G, nan, "logw theta NaN")
run_logw(y, YT, G, G, inf, "logw theta inf")
run_logw(y, YT, G, G, -1.0, "logw theta < 0")
w0 = np.full(N, 1.0 / N)
f0 = np.zeros(M)
f_nan = f0.copy(); f_nan[2] = nan
w0_nan = w0.copy(); w0_nan[9] = nan
w0_neg = w0.copy(); w0_neg[9] = -w0_neg[9]
run_forces(y, YT, f0, w0, 10.0, "forces clean")
run_forces(y, YT, f_nan, w0, 10.0, "forces NaN in forces_init")
run_forces(y, YT, f0, w0_nan, 10.0, "forces NaN in w0")
run_forces(y, YT, f0, w0_neg, 10.0, "forces negative w0 entry")
run_forces(y_nan, YT, f0, w0, 10.0, "forces NaN in yTilde")
run_forces(y, YT_nan, f0, w0, 10.0, "forces NaN in YTilde")
run_forces(y, YT, f0, w0, nan, "forces theta NaN")
