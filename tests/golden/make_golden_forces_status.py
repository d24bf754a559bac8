#!/usr/bin/env python3
"""Golden record of how the REFERENCE's forces-method runs END on a configs[4]-shaped problem (run in the BUILD
container only; needs oracle/_ref/libbioen_ref.so, i.e. /root/reference).

Problem : SURVEY 8(d)'s synthetic recipe (numpy default_rng(12345), row-wise normals), M = 512 observables x N = 1e5
          structures, w0 = 1/N, forces_init = 0, the 8 thetas of np.logspace(3, -0.5, 8), liblbfgs at the yaml defaults
          (bioen_optimize.yaml:33-46) -- BASELINE configs[4] with a tenth of the structures, so that the matrix can be
          re-made from the seed wherever the test runs (410 MB) and the reference finishes in seconds.
Why     : at large theta the run ends at a point where the decrease the line search still asks for lies below the
          rounding noise of the objective itself (f ~ 248, its evaluation noise ~ 5e-13; remaining decrease
          1/2 g^2 / (theta var) ~ 1e-15).  Whether liblbfgs then returns 0 (the gradient test |g| / max(1, |x|) <= 1e-6 met
          after a lucky last step, lbfgs.c:503-508) or -998 (line search exhausted, lbfgs.c:727-729) is decided by
          rounding: the reference's OWN answer changes with its summation mode (fast_openmp 0 / 1,
          c_bioen_common.c:46-55), its thread count, and from run to run of the same configuration (OpenMP reduction
          order).  This script records that spread: per theta every (mode, threads, repetition) -> (status, fmin,
          iterations, evaluations), through liblbfgs' own lbfgs() with logging callbacks (oracle/ref_trace.py) and, for
          each configuration, through the reference's driver _opt_lbfgs_forces itself.
Output  : tests/golden/forces_status_cfg4_M512xN100000.json (data only).
Usage   : python tests/golden/make_golden_forces_status.py [N] [M]
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_binding as R                    # noqa: E402
from oracle.ref_trace import traced_lbfgs, RefObjective   # noqa: E402

DEFAULTS = dict(linesearch=2, max_iterations=5000, delta=1e-6, epsilon=1e-6, ftol=1e-5, gtol=0.9,
                wolfe=0.9, past=10, max_linesearch=100)           # bioen_optimize.yaml:33-46


def survey_matrix(M, N, seed=12345):
    """SURVEY 8(d) / bioen/optimize/forces.py:19-68: row-wise normals around YTrue, scaled by the experimental error"""
    rng = np.random.default_rng(seed)
    YTrue = rng.uniform(1, 10, M)
    sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
    yT = np.empty((M, N))
    for i in range(M):
        yT[i, :] = rng.normal(YTrue[i], sig_sim[i], N) / sig_exp[i]
    YT = rng.normal(YTrue, sig_exp) / sig_exp
    return yT, YT


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    yT, YT = survey_matrix(M, N)
    w0 = np.full(N, 1.0 / N)
    x0 = np.zeros(M)
    thetas = [float(t) for t in np.logspace(3, -0.5, 8)]
    ncpu = len(os.sched_getaffinity(0))
    full = []
    for threads in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4)}, reverse=True):
        for flag in (1, 0):
            for rep in range(2):
                full.append((flag, threads, rep))
    full += [(1, 1, 0)]
    # thetas below 100 end on the plateau test (status 1) long before the rounding floor: the reference is unanimous there
    # -- four configurations document that, the twenty of the large thetas are not repeated (hundreds of iterations each)
    few = [(1, ncpu, 0), (0, ncpu, 0), (1, max(1, ncpu // 2), 0), (0, max(1, ncpu // 2), 0)]
    out = {"M": M, "N": N, "seed": 12345, "lbfgs": DEFAULTS, "host_cpus": ncpu,
           "made_by": "tests/golden/make_golden_forces_status.py from oracle/_ref/libbioen_ref.so (the reference's C path + liblbfgs 1.10)",
           "per_theta": []}
    path = os.path.join(HERE, "forces_status_cfg4_M%dxN%d.json" % (M, N))
    if os.path.isfile(path):                       # resume: thetas already recorded are kept
        with open(path) as fp:
            old = json.load(fp)
        if (old.get("M"), old.get("N"), old.get("seed")) == (M, N, 12345):
            out["per_theta"] = old["per_theta"]
    done = {round(p["theta"], 9) for p in out["per_theta"]}
    for th in thetas:
        if round(th, 9) in done:
            continue
        variants = full if th >= 99.0 else (few if th >= 0.9 else few[:2])      # (the smallest theta: thousands of iterations a run)
        obj = RefObjective(yT, YT, w0, th)
        runs = []
        t0 = time.perf_counter()
        for flag, threads, rep in variants:
            R.set_fast_openmp_flag(flag)
            R.omp_set_num_threads(threads)
            _, fx, code, evals, its = traced_lbfgs(obj, x0, DEFAULTS)
            _, fmin2, code2 = R.opt_lbfgs_forces(x0, w0, yT, YT, th, DEFAULTS)
            runs.append({"fast_openmp": flag, "threads": threads, "rep": rep, "code": int(code), "fmin": float(fx),
                         "iterations": len(its), "evaluations": len(evals),
                         "last_gnorm_ratio": float(its[-1]["ratio"]) if its else None,
                         "driver_code": int(code2), "driver_fmin": float(fmin2)})
        codes = sorted({r["code"] for r in runs} | {r["driver_code"] for r in runs})
        fm = [r["fmin"] for r in runs] + [r["driver_fmin"] for r in runs]
        out["per_theta"].append({"theta": th, "codes": codes, "fmin_min": min(fm), "fmin_max": max(fm),
                                 "fmin_rel_spread": (max(fm) - min(fm)) / abs(min(fm)), "runs": runs})
        print("theta %-8.4g codes %-14s fmin %.15g  spread %.1e  iterations %s  (%.0f s)"
              % (th, codes, min(fm), (max(fm) - min(fm)) / abs(min(fm)), sorted({r["iterations"] for r in runs}),
                 time.perf_counter() - t0), flush=True)
        out["per_theta"].sort(key=lambda p: -p["theta"])
        with open(path, "w") as fp:                # after every theta
            json.dump(out, fp, indent=1)
    R.set_fast_openmp_flag(1)
    R.omp_set_num_threads(ncpu)
    print("wrote", path)


if __name__ == "__main__":
    main()
