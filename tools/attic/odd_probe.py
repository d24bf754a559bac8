"""Development aid (GPU box): valid-but-odd inputs -- extreme thetas, one observable, one or two structures, constant
columns, duplicated structures -- device beside the reference's binary (status, counts, fmin, weights)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
from oracle import ref_binding as R
from bench import LBFGS_DEFAULTS

params = dict(LBFGS_DEFAULTS, max_iterations=300)


def problem(M, N, seed=3):
    rng = np.random.default_rng(seed)
    YTrue = rng.uniform(1, 10, M)
    sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
    y = rng.normal(YTrue[:, None], sig_sim[:, None], (M, N)) / sig_exp[:, None]
    return y, rng.normal(YTrue, sig_exp) / sig_exp


def both(tag, y, YT, theta, g0=None, G=None, forces=False):
    M, N = y.shape
    G = np.zeros(N) if G is None else G
    g0 = G if g0 is None else g0
    w0 = np.full(N, 1.0 / N)
    try:
        with bioen_amd.Context(y, YT) as ctx:
            if forces:
                x, w, info = ctx.opt_lbfgs_forces(np.zeros(M), w0, theta, params)
            else:
                x, w, info = ctx.opt_lbfgs_logw(g0, G, theta, params)
        dev = (info.lbfgs_code, info.iterations, info.evaluations, info.fmin)
    except Exception as e:
        dev, w = ("exception", repr(e)[:100]), None
    try:
        if forces:
            xr, fr, cr = R.opt_lbfgs_forces(np.zeros(M), w0, y, YT, theta, params)
            wr = np.asarray(R.forces_weights(xr, w0, y)).ravel()
        else:
            xr, fr, cr = R.opt_lbfgs_logw(g0, G, y, YT, theta, params)
            wr = np.exp(xr - xr.max()); wr /= wr.sum()
        ref = (cr, fr)
    except Exception as e:
        ref, wr = ("exception", repr(e)[:100]), None
    dw = None if w is None or wr is None else float(np.abs(np.asarray(w).ravel() - wr).max() / max(wr.max(), 1e-300))
    print("%-38s device %s | reference %s | max|dw|/max(w) %s" % (tag, dev, ref, dw))
    sys.stdout.flush()


y, YT = problem(64, 4000)
for th in (0.0, 1e-300, 1e-12, 1e12, 1e300):
    both("logw theta=%g" % th, y, YT, th)
    both("forces theta=%g" % th, y, YT, th, forces=True)
y1, YT1 = problem(1, 500)
both("logw M=1", y1, YT1, 1.0)
both("forces M=1", y1, YT1, 1.0, forces=True)
yn1, YTn1 = problem(16, 1)
both("logw N=1", yn1, YTn1, 1.0)
both("forces N=1", yn1, YTn1, 1.0, forces=True)
yn2, YTn2 = problem(16, 2)
both("logw N=2", yn2, YTn2, 1.0)
both("forces N=2", yn2, YTn2, 1.0, forces=True)
yc = y.copy(); yc[:, :] = yc[:, :1]                     # every structure the same: the gradient is exactly zero
both("logw identical structures", yc, YT, 1.0)
both("forces identical structures", yc, YT, 1.0, forces=True)
yz = np.zeros_like(y)
both("logw zero matrix", yz, YT, 1.0)
both("forces zero matrix", yz, YT, 1.0, forces=True)
yd = y.copy(); yd[:, 2000:] = yd[:, :2000]              # every structure twice
both("logw duplicated structures", yd, YT, 10.0)
both("forces duplicated structures", yd, YT, 10.0, forces=True)
ybig = y * 1e150
both("logw yTilde * 1e150 (chi^2 overflows)", ybig, YT, 1.0)
both("forces yTilde * 1e150", ybig, YT, 1.0, forces=True)
