"""Canonical-segment probe (r05): (1) the forward strip kernel's in-register fold against one chunk per slot, bit for bit;
(2) the bench workload's per-theta counts and minima (the numbers tests/test_hip_fullsize.py pins) with its wall time in
both forms; (3) the forces series of configs[4].  Run on the GPU box:  python tools/canon_probe.py [quick]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bioen_amd                                  # noqa: E402
from bioen_amd import sweep                        # noqa: E402
from conftest import LBFGS_DEFAULTS                # noqa: E402


def targets(M, seed=12345):
    rng = np.random.default_rng(seed)
    YTrue = rng.uniform(1, 10, M)
    sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
    YTilde = rng.normal(YTrue, sig_exp) / sig_exp
    return YTrue, sig_sim, sig_exp, YTilde


def series(M, N, thetas, fold, engine=None, method="logw", max_batch=8):
    os.environ["BIOEN_HIP_STRIP_FOLD"] = "1" if fold else "0"
    if engine is not None:
        os.environ["BIOEN_HIP_DEVICE_LS"] = engine
    try:
        YTrue, sig_sim, sig_exp, YTilde = targets(M)
        with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
            if method == "logw":
                G = np.zeros(N)
                sweep.sweep_log_weights(ctx, thetas[:2], G, G, LBFGS_DEFAULTS, max_batch=max_batch)      # copies, warm-up
                t0 = time.perf_counter()
                res = sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS, max_batch=max_batch)
                dt = time.perf_counter() - t0
                return res, dt
            w0 = np.full(N, 1.0 / N)
            f0 = np.zeros(M)
            ctx.opt_lbfgs_forces_batch(thetas[:1], f0, w0, LBFGS_DEFAULTS)
            t0 = time.perf_counter()
            res, w, infos = ctx.opt_lbfgs_forces_batch(thetas, f0, w0, LBFGS_DEFAULTS, max_batch=max_batch)
            dt = time.perf_counter() - t0
            return [dict(theta=t, iterations=i.iterations, evaluations=i.evaluations, fmin=i.fmin, code=i.lbfgs_code,
                         w=w[k], x=res[k]) for k, (t, i) in enumerate(zip(thetas, infos))], dt
    finally:
        os.environ.pop("BIOEN_HIP_STRIP_FOLD", None)
        os.environ.pop("BIOEN_HIP_DEVICE_LS", None)


def same(a, b):
    return all(x["iterations"] == y["iterations"] and x["evaluations"] == y["evaluations"] and x["fmin"] == y["fmin"]
               and np.array_equal(x["w"], y["w"]) for x, y in zip(a, b))


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    thetas = list(np.logspace(3, -0.5, 8))
    for M, N in [(64, 2000), (28, 50001), (256, 100000), (1024, 20000), (205, 50000)]:
        a, ta = series(M, N, thetas, True)
        b, tb = series(M, N, thetas, False)
        c, tc = series(M, N, thetas, True, engine="0")
        print("fold == unfolded == host engine, M = %d, N = %d: %s %s  (%.1f / %.1f / %.1f ms; %d iterations)"
              % (M, N, same(a, b), same(a, c), 1e3 * ta, 1e3 * tb, 1e3 * tc, sum(r["iterations"] for r in a)), flush=True)
    if quick:
        return
    M, N = 1024, 1000000
    a, ta = series(M, N, thetas, True)
    print("headline, fold: %.4f s" % ta)
    for r in a:
        print("    (%r, %d, %d, %r)," % (float(r["theta"]), r["iterations"], r["evaluations"], float(r["fmin"])))
    print("  total iterations %d evaluations %d" % (sum(r["iterations"] for r in a), sum(r["evaluations"] for r in a)), flush=True)
    b, tb = series(M, N, thetas, False)
    print("headline, one chunk per slot: %.4f s, same bits: %s" % (tb, same(a, b)), flush=True)
    c, tc = series(M, N, thetas, True, engine="1")
    print("headline, device engine: %.4f s, same bits: %s" % (tc, same(a, c)), flush=True)
    f, tf = series(512, N, thetas, True, method="forces")
    print("forces configs[4]: %.4f s" % tf)
    for r in f:
        print("    theta %g: %d iterations, %d evaluations, fmin %r, code %d" % (r["theta"], r["iterations"], r["evaluations"], r["fmin"], r["code"]))


if __name__ == "__main__":
    main()
