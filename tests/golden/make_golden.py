#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run in the BUILD container only).

Inputs  : the reference's own test fixtures ``/root/reference/test/optimize/data/*.pkl``
          (python-2 pickles of np.matrix; element order per
          test_find_opt_analytical_grad_logw.py:67-68 / ..._forces.py:57-58) and
          its known answers ``*.ref``; plus seeded random problems in the regime the
          reference's tests never cover (non-uniform G, GInit != G, theta > 0).
Outputs : one ``.npz`` per case holding the INPUT arrays and the values the
          REFERENCE computes for them -- obtained by calling the reference's own C
          code (oracle/_ref/libbioen_ref.so, built by oracle/Makefile from the
          reference sources) through oracle/ref_binding.py.

Only data is stored (inputs + expected outputs); no reference source text.
Usage: python tests/golden/make_golden.py
"""
import glob
import os
import pickle
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_binding as R  # noqa: E402
from oracle import oracle_binding as O  # noqa: E402

REFDATA = "/root/reference/test/optimize/data"

DEFAULTS = dict(linesearch=2, max_iterations=5000, delta=1e-6, epsilon=1e-6, ftol=1e-5, gtol=0.9,
                wolfe=0.9, past=10, max_linesearch=100)           # bioen_optimize.yaml:33-46
TIGHT = dict(DEFAULTS, epsilon=1e-7, delta=1e-11)                 # converged-optimum parity runs
MORETHUENTE = dict(DEFAULTS, linesearch=0)
STRONG = dict(DEFAULTS, linesearch=3)
# "conv": the stopping rules that let an implementation stop early are switched off (no plateau
# test, gradient test at 1e-9), so every run ends AT the optimum -- on the epsilon test or, where
# the objective is flat to the last bit first, on an exhausted line search (-998 / -1001) at the
# rounding floor.  These runs are what north_star's 1e-6 (L) / 1e-5 (weights) are asserted on.
CONV = dict(DEFAULTS, epsilon=1e-9, delta=0.0, past=0, max_iterations=200000)
CONVMT = dict(CONV, linesearch=0)


def arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def logw_case(GInit, G, y, yTilde, YTilde, theta, ref_value=None):
    GInit, G, y, yTilde, YTilde = arr(GInit), arr(G), arr(y), arr(yTilde), arr(YTilde)
    out = dict(kind="logw", GInit=GInit, G=G, y=y, yTilde=yTilde, YTilde=YTilde, theta=float(theta))
    if np.array_equal(y, yTilde):
        out["y"] = np.zeros(0)                       # y == yTilde: store once
    R.set_fast_openmp_flag(0)
    w, s = R.get_weights(GInit)
    out["w_init"] = w
    out["s_init"] = s
    out["f_init"] = R.logw_f(GInit, G, yTilde, YTilde, theta)
    out["grad_init"] = R.logw_df(GInit, G, yTilde, YTilde, theta, caching=True)
    # a second evaluation point away from the start (deterministic perturbation)
    rng = np.random.default_rng(2024)
    gp = GInit.ravel() + 0.3 * rng.standard_normal(GInit.size)
    out["g_pert"] = gp
    out["f_pert"] = R.logw_f(gp, G, yTilde, YTilde, theta)
    out["grad_pert"] = R.logw_df(gp, G, yTilde, YTilde, theta, caching=True)
    for tag, cfg in (("def", DEFAULTS), ("tight", TIGHT), ("mt", MORETHUENTE), ("strong", STRONG),
                     ("conv", CONV), ("convmt", CONVMT)):
        gopt, fmin, code = R.opt_lbfgs_logw(GInit, G, yTilde, YTilde, theta, cfg)
        out["lbfgs_%s_fmin" % tag] = fmin
        out["lbfgs_%s_code" % tag] = code
        out["lbfgs_%s_wopt" % tag] = R.get_weights(gopt)[0]
        # How well do these settings pin the optimum?  The same algorithm restated with a
        # different summation order (oracle/bioen_oracle.c) stops at a slightly different point;
        # the spread between the two is the conditioning of the stopping point itself and
        # bounds what ANY implementation can be held to on this case.
        g_o = O.opt_lbfgs_logw(GInit, G, yTilde, YTilde, theta, cfg)[0]
        w_o = O.logw_weights(g_o)[0]
        out["lbfgs_%s_wspread" % tag] = np.abs(w_o - out["lbfgs_%s_wopt" % tag]).max() / out["lbfgs_%s_wopt" % tag].max()
    if ref_value is not None:
        out["ref_fmin_scipy_bfgs"] = float(ref_value)
    return out


def forces_case(forces_init, w0, y, yTilde, YTilde, theta, ref_value=None):
    forces_init, w0, y, yTilde, YTilde = arr(forces_init), arr(w0), arr(y), arr(yTilde), arr(YTilde)
    out = dict(kind="forces", forces_init=forces_init, w0=w0, y=y, yTilde=yTilde, YTilde=YTilde,
               theta=float(theta))
    if np.array_equal(y, yTilde):
        out["y"] = np.zeros(0)
    R.set_fast_openmp_flag(0)
    out["w_init"] = R.forces_weights(forces_init, w0, yTilde)
    out["f_init"] = R.forces_f(forces_init, w0, yTilde, YTilde, theta)
    out["grad_init"] = R.forces_df(forces_init, w0, yTilde, YTilde, theta)
    rng = np.random.default_rng(2025)
    fp = forces_init.ravel() + 1e-3 * rng.standard_normal(forces_init.size)
    out["forces_pert"] = fp
    out["w_pert"] = R.forces_weights(fp, w0, yTilde)
    out["f_pert"] = R.forces_f(fp, w0, yTilde, YTilde, theta)
    out["grad_pert"] = R.forces_df(fp, w0, yTilde, YTilde, theta)
    for tag, cfg in (("def", DEFAULTS), ("tight", TIGHT), ("mt", MORETHUENTE), ("conv", CONV), ("convmt", CONVMT)):
        fopt, fmin, code = R.opt_lbfgs_forces(forces_init, w0, yTilde, YTilde, theta, cfg)
        out["lbfgs_%s_fmin" % tag] = fmin
        out["lbfgs_%s_code" % tag] = code
        out["lbfgs_%s_fopt" % tag] = fopt
        out["lbfgs_%s_wopt" % tag] = R.forces_weights(fopt, w0, yTilde)
        f_o = O.opt_lbfgs_forces(forces_init, w0, yTilde, YTilde, theta, cfg)[0]
        w_o = O.forces_weights(f_o, w0, yTilde)
        out["lbfgs_%s_wspread" % tag] = np.abs(w_o - out["lbfgs_%s_wopt" % tag]).max() / out["lbfgs_%s_wopt" % tag].max()
    if ref_value is not None:
        out["ref_fmin_scipy_bfgs"] = float(ref_value)
    return out


def synth(M, N, seed, nonuniform=True):
    """SURVEY.md 8(d) recipe (after forces.py:19-68) + non-uniform reference weights."""
    rng = np.random.default_rng(seed)
    YTrue = rng.uniform(1, 10, M)
    sig_exp = 0.1 * YTrue
    sig_sim = 0.5 * YTrue
    yTilde = np.empty((M, N))
    for i in range(M):
        yTilde[i, :] = rng.normal(YTrue[i], sig_sim[i], N) / sig_exp[i]
    YTilde = (rng.normal(YTrue, sig_exp) / sig_exp)[None, :]
    if nonuniform:
        w0 = rng.uniform(0.2, 1.8, N)
        w0 /= w0.sum()
    else:
        w0 = np.full(N, 1.0 / N)
    return yTilde, YTilde, w0[:, None], rng


def main():
    warnings.filterwarnings("ignore")
    if not R.available():
        sys.exit("oracle/_ref/libbioen_ref.so missing: run `make -C oracle ref` first")

    # --- the reference's own fixtures -------------------------------------------------
    for path in sorted(glob.glob(os.path.join(REFDATA, "*.pkl"))):
        name = os.path.splitext(os.path.basename(path))[0]
        if name == "data_M64xN64":      # malformed stray file (63-vector first), unused by the tests
            continue
        with open(path, "rb") as fp:
            x = pickle.load(fp, encoding="latin1")
        refp = os.path.join(REFDATA, name + ".ref")
        ref_value = None
        if os.path.isfile(refp):
            with open(refp, "rb") as fp:
                ref_value = pickle.load(fp, encoding="latin1")
        if len(x) == 7:
            GInit, G, y, yTilde, YTilde, w0, theta = x
            case = logw_case(GInit, G, y, yTilde, YTilde, theta, ref_value)
            case["w0"] = arr(w0)
        else:
            forces_init, w0, y, yTilde, YTilde, theta = x
            case = forces_case(forces_init, w0, y, yTilde, YTilde, theta, ref_value)
        np.savez_compressed(os.path.join(HERE, "ref_" + name + ".npz"), **case)
        print("ref_%s: %s M=%d N=%d theta=%g" % (name, case["kind"], case["yTilde"].shape[0],
                                                  case["yTilde"].shape[1], case["theta"]))

    # --- seeded problems: non-uniform G, GInit != G, theta > 0 ------------------------
    for (M, N, seed, theta) in ((37, 500, 101, 5.0), (64, 2000, 102, 0.7), (129, 257, 103, 50.0)):
        yTilde, YTilde, w0, rng = synth(M, N, seed)
        G = np.log(w0)
        G = G - G[-1, 0]                                    # log_weights.getGs, log_weights.py:113-127
        GInit = G + 0.2 * rng.standard_normal((N, 1))       # start away from the reference weights
        case = logw_case(GInit, G, yTilde, yTilde, YTilde, theta)
        case["w0"] = w0
        case["seed"] = seed
        np.savez_compressed(os.path.join(HERE, "synth_logw_M%dxN%d.npz" % (M, N)), **case)
        print("synth_logw M=%d N=%d theta=%g fmin(def)=%.12g" % (M, N, theta, case["lbfgs_def_fmin"]))

    for (M, N, seed, theta) in ((30, 1000, 201, 10.0), (96, 3000, 202, 2.0)):
        yTilde, YTilde, w0, rng = synth(M, N, seed)
        forces_init = np.zeros((M, 1))
        case = forces_case(forces_init, w0, yTilde, yTilde, YTilde, theta)
        case["seed"] = seed
        np.savez_compressed(os.path.join(HERE, "synth_forces_M%dxN%d.npz" % (M, N)), **case)
        print("synth_forces M=%d N=%d theta=%g fmin(def)=%.12g" % (M, N, theta, case["lbfgs_def_fmin"]))

    # --- error path (c_bioen.pyx:516-520; test_error_opt_logw.py:65-83) -----------------
    yTilde, YTilde, w0, rng = synth(8, 32, 301, nonuniform=False)
    G = np.zeros((32, 1))
    bad = dict(DEFAULTS, delta=-1.0)
    _, _, code_delta = R.opt_lbfgs_logw(G, G, yTilde, YTilde, 1.0, bad)
    capped = dict(DEFAULTS, max_iterations=3)
    _, fcap, code_cap = R.opt_lbfgs_logw(G, G, yTilde, YTilde, 1.0, capped)
    np.savez_compressed(os.path.join(HERE, "error_codes.npz"), yTilde=yTilde, YTilde=YTilde,
                        code_delta_neg=code_delta, msg_delta_neg=R.lbfgs_strerror(code_delta),
                        code_maxiter=code_cap, msg_maxiter=R.lbfgs_strerror(code_cap), f_maxiter=fcap,
                        msg_0=R.lbfgs_strerror(0), msg_1=R.lbfgs_strerror(1), msg_2=R.lbfgs_strerror(2))
    print("error codes:", code_delta, code_cap)


if __name__ == "__main__":
    main()
