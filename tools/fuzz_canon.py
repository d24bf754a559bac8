"""Development aid (GPU box): canonical segments under random problems -- a structure-sharded run over W in {2, 4, 8}
ranks (threads of this process on the one GPU, host-staged exchanges: sweep.ThreadComm) must return the BITS of the
single-GPU run: objective and gradient at a random point, a capped L-BFGS batch, a capped GSL-style run, both methods,
all kernel geometries (M = 5 ... 1100: two-wave and eight-wave strips, the tall strip kernel, row panels; N = 300 ... 6000:
segments with no structure at all, ragged last segments, a last rank with a handful of structures), all line searches.
SEEDS=n (default 40), FIRST=k, WORLDS=2,4,8.  Prints every violation."""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd                      # noqa: E402
from bioen_amd import _lib             # noqa: E402
from bioen_amd import sweep            # noqa: E402
from bench import LBFGS_DEFAULTS       # noqa: E402


def workload(ctx, p):
    out = {}
    out["w"], out["logs"] = ctx.logw_weights(p["g"])
    f, grad = ctx.logw_fdf(p["g"], p["G"], p["thetas"][0])
    out["f"], out["grad"] = np.float64(f), grad
    res, w, infos = ctx.opt_lbfgs_logw_batch(p["thetas"], p["G"], p["G"], p["params"], max_batch=p["max_batch"])
    out["res"], out["wopt"] = res, w
    out["stat"] = np.array([(i.fmin, i.iterations, i.evaluations, i.lbfgs_code, i.chi2, i.kl) for i in infos])
    if p["gsl"]:
        gg, wg, ig = ctx.opt_gsl_logw(p["G"], p["G"], p["thetas"][0], p["gsl"], dict(step_size=0.01, tol=1e-3, max_iterations=12))
        out["ggsl"], out["gslstat"] = gg, np.array([ig.fmin, ig.iterations, ig.evaluations, ig.lbfgs_code])
    out["fw"] = ctx.forces_weights(p["f0"], p["w0"])
    ff, fgrad = ctx.forces_fdf(p["f0"], p["w0"], p["thetas"][0])
    out["ff"], out["fgrad"] = np.float64(ff), fgrad
    fres, fwo, finfos = ctx.opt_lbfgs_forces_batch(p["thetas"], np.zeros(p["M"]), p["w0"], p["params"], max_batch=p["max_batch"])
    out["fres"], out["fwopt"] = fres, fwo
    out["fstat"] = np.array([(i.fmin, i.iterations, i.evaluations, i.lbfgs_code, i.chi2, i.kl) for i in finfos])
    if p["gsl"]:
        fg, fwg, fig = ctx.opt_gsl_forces(np.zeros(p["M"]), p["w0"], p["thetas"][0], p["gsl"], dict(step_size=0.01, tol=1e-3, max_iterations=8))
        out["fgsl"], out["fwgsl"] = fg, fwg
        out["fgslstat"] = np.array([fig.fmin, fig.iterations, fig.evaluations, fig.lbfgs_code])
    return out


def sharded(p, world):
    comms = sweep.ThreadComm.create(world, timeout=90.0)
    results, errors = [None] * world, [None] * world

    def rank_main(r):
        try:
            ctx = bioen_amd.Context(p["y"], p["YT"], device=0, rank=r, world=world)
            try:
                ctx.set_exchange(comms[r])
                results[r] = workload(ctx, p)
            finally:
                ctx.close()
        except BaseException as e:      # noqa: B902 -- reported by the caller
            errors[r] = e
            try:
                comms[r]._s.barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    if any(t.is_alive() for t in threads):
        raise RuntimeError("a rank did not finish")
    return results, errors


def run(first, nseeds, worlds):
    bad = []
    for seed in range(first, first + nseeds):
        rng = np.random.default_rng(9100 + seed)
        M = int(rng.choice([5, 16, 37, 64, 96, 205, 512, 600, 1024, 1100]))
        N = int(rng.choice([300, 897, 1000, 1025, 1793, 2049, 3000, 6000]))      # 897, 1793: ONE structure in the last of 8 segments
        if M >= 600:
            N = min(N, 2049)
        YTrue = rng.uniform(1, 10, M)
        y = rng.normal(YTrue[:, None], 0.5 * YTrue[:, None], (M, N)) / (0.1 * YTrue[:, None])
        YT = rng.normal(YTrue, 0.1 * YTrue) / (0.1 * YTrue)
        nt = int(rng.integers(1, 7))
        p = dict(M=M, N=N, y=y, YT=YT, thetas=[float(t) for t in 10.0 ** rng.uniform(-0.5, 3.0, nt)],
                 max_batch=int(rng.integers(1, 9)), G=np.log(rng.dirichlet(np.ones(N) * 2.0)), w0=rng.dirichlet(np.ones(N) * 2.0),
                 f0=1e-4 * rng.standard_normal(M),
                 params=dict(LBFGS_DEFAULTS, linesearch=int(rng.choice([0, 1, 2, 3])), max_iterations=int(rng.integers(2, 25))),
                 gsl=str(rng.choice(["", "", "bfgs2", "conjugate_pr", "steepest_descent"])))
        p["g"] = p["G"] + 0.3 * rng.standard_normal(N)
        world = int(rng.choice(worlds))
        # a decomposition that would leave a rank without a structure is refused at context creation (documented:
        # include/bioen_hip.h, bioen_hip_ctx_create_sharded) -- drawn again with fewer ranks, down to what N admits
        while world > 1 and (world - 1) * _lib.column_segments(N, world)[2] >= N:
            world //= 2
        if world == 1:
            continue
        tag = "seed %d: M=%d N=%d W=%d thetas=%d batch=%d ls=%d it<=%d gsl=%s" % (
            seed, M, N, world, nt, p["max_batch"], p["params"]["linesearch"], p["params"]["max_iterations"], p["gsl"] or "-")
        try:
            with bioen_amd.Context(y, YT) as ctx:
                single = workload(ctx, p)
            results, errors = sharded(p, world)
        except Exception as e:          # noqa: B902
            bad.append("%s: %s: %s" % (tag, type(e).__name__, e))
            print(bad[-1], flush=True)
            continue
        if any(e is not None for e in errors):
            bad.append("%s: rank errors %s" % (tag, [repr(e) for e in errors if e is not None][:2]))
            print(bad[-1], flush=True)
            continue
        for r in range(world):
            for key, val in single.items():
                if not np.array_equal(np.asarray(results[r][key]), np.asarray(val), equal_nan=True):
                    a, b = np.asarray(results[r][key], dtype=float), np.asarray(val, dtype=float)
                    bad.append("%s: rank %d: %s differs from the single-GPU run (max |diff| %.3g)" % (
                        tag, r, key, float(np.nanmax(np.abs(a - b))) if a.shape == b.shape else float("nan")))
                    print(bad[-1], flush=True)
                    break
        if (seed - first) % 10 == 9:
            print("... %d seeds, %d violations" % (seed - first + 1, len(bad)), flush=True)
    return bad


if __name__ == "__main__":
    worlds = [int(w) for w in os.environ.get("WORLDS", "2,4,8").split(",")]
    bad = run(int(os.environ.get("FIRST", "0")), int(os.environ.get("SEEDS", "40")), worlds)
    print("fuzz_canon: %d violations" % len(bad))
    sys.exit(1 if bad else 0)
