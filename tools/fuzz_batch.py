"""Development aid (GPU box): randomised batches -- 1 ... 14 thetas over 1 ... 8 batch slots (slots are reused, problems
finish at different times, shadows come and go), shared or per-theta starts, all line searches, random iteration caps, both
methods, both log-weights engines -- every problem of a batch must return the bits of its single run.
SEEDS=n (default 60), FIRST=k.  Prints every violation."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
from bench import LBFGS_DEFAULTS


def sig(x, w, info):
    bits = lambda v: np.float64(v).tobytes()                 # (NaN == NaN here)
    return (x.tobytes(), None if w is None else w.tobytes(), bits(info.fmin), info.iterations, info.evaluations, info.lbfgs_code,
            bits(info.chi2), bits(info.kl))


def run(first, nseeds):
    bad = []
    for seed in range(first, first + nseeds):
        rng = np.random.default_rng(5000 + seed)
        M = int(rng.choice([16, 28, 64, 96, 205, 512, 600, 1030]))
        N = int(rng.choice([300, 1000, 2049, 5000]))
        YTrue = rng.uniform(1, 10, M)
        y = rng.normal(YTrue[:, None], 0.5 * YTrue[:, None], (M, N)) / (0.1 * YTrue[:, None])
        YT = rng.normal(YTrue, 0.1 * YTrue) / (0.1 * YTrue)
        nt = int(rng.integers(1, 15))
        thetas = 10.0 ** rng.uniform(-1.5, 3.0, nt)
        if rng.random() < 0.3 and nt > 1:
            thetas[rng.integers(0, nt)] = thetas[0]                    # a repeated theta
        max_batch = int(rng.integers(1, 9))
        G = np.log(rng.dirichlet(np.ones(N) * 2.0))
        w0 = rng.dirichlet(np.ones(N) * 2.0)
        shared = rng.random() < 0.5
        g0 = G if shared else np.stack([G + 0.05 * k * rng.standard_normal(N) for k in range(nt)])
        f0 = np.zeros(M) if shared else np.stack([1e-4 * k * rng.standard_normal(M) for k in range(nt)])
        params = dict(LBFGS_DEFAULTS, linesearch=int(rng.choice([0, 1, 2, 3])), max_iterations=int(rng.integers(2, 40)))
        engine = str(rng.choice(["0", "1"]))
        tag = "seed %d: M=%d N=%d thetas=%d batch=%d ls=%d it<=%d engine=%s %s" % (
            seed, M, N, nt, max_batch, params["linesearch"], params["max_iterations"], engine, "shared" if shared else "own starts")
        os.environ["BIOEN_HIP_DEVICE_LS"] = engine
        try:
            with bioen_amd.Context(y, YT) as ctx:
                res, w, infos = ctx.opt_lbfgs_logw_batch(thetas, g0, G, params, max_batch=max_batch)
                fres, fw, finfos = ctx.opt_lbfgs_forces_batch(thetas, f0, w0, params, max_batch=max_batch)
                for k in range(nt):
                    one = ctx.opt_lbfgs_logw(g0 if shared else g0[k], G, thetas[k], params)
                    if sig(res[k], w[k], infos[k]) != sig(*one):
                        bad.append("%s: log-weights problem %d (theta %.4g): batch (%d, %d it, %d ev, %.15g) single (%d, %d it, %d ev, %.15g)" % (
                            tag, k, thetas[k], infos[k].lbfgs_code, infos[k].iterations, infos[k].evaluations, infos[k].fmin,
                            one[2].lbfgs_code, one[2].iterations, one[2].evaluations, one[2].fmin))
                    if not (np.isfinite(res[k]).all() and np.isfinite(infos[k].fmin)):
                        bad.append("%s: log-weights problem %d (theta %.4g) returned non-finite numbers, status %d" % (tag, k, thetas[k], infos[k].lbfgs_code))
                    fone = ctx.opt_lbfgs_forces(f0 if shared else f0[k], w0, thetas[k], params)
                    if not (np.isfinite(fres[k]).all() and np.isfinite(finfos[k].fmin)):
                        bad.append("%s: forces problem %d (theta %.4g) returned non-finite numbers, status %d" % (tag, k, thetas[k], finfos[k].lbfgs_code))
                    if sig(fres[k], fw[k], finfos[k]) != sig(*fone):
                        bad.append("%s: forces problem %d (theta %.4g): batch (%d, %d it, %d ev, %.15g) single (%d, %d it, %d ev, %.15g)" % (
                            tag, k, thetas[k], finfos[k].lbfgs_code, finfos[k].iterations, finfos[k].evaluations, finfos[k].fmin,
                            fone[2].lbfgs_code, fone[2].iterations, fone[2].evaluations, fone[2].fmin))
        except Exception as e:
            bad.append(tag + " EXCEPTION " + repr(e)[:200])
        finally:
            del os.environ["BIOEN_HIP_DEVICE_LS"]
    return bad


if __name__ == "__main__":
    n = int(os.environ.get("SEEDS", "60"))
    bad = run(int(os.environ.get("FIRST", "0")), n)
    print("seeds", n, "violations:", len(bad))
    for b in bad:
        print("  ", b)
