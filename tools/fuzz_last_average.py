"""Development aid (GPU box): the ensemble average a single-problem optimizer call leaves on the device (last_average: what
the nuisance refits read instead of the N weights) against yTilde . w of the weights that call returned -- random shapes,
thetas, line searches, caps and status codes (converged, plateau, budget, failed searches that revert to the previous
point), with and without an affine model, both methods, both engines.  SEEDS=n (default 60)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd
from bench import LBFGS_DEFAULTS


def run(first, nseeds):
    bad, codes = [], {}
    for seed in range(first, first + nseeds):
        rng = np.random.default_rng(13000 + seed)
        M = int(rng.choice([16, 28, 96, 205, 512, 600, 1100]))
        N = int(rng.choice([17, 300, 2049, 5000]))
        YTrue = rng.uniform(1, 10, M)
        y = rng.normal(YTrue[:, None], 0.5 * YTrue[:, None], (M, N)) / (0.1 * YTrue[:, None])
        YT = rng.normal(YTrue, 0.1 * YTrue) / (0.1 * YTrue)
        theta = float(10.0 ** rng.uniform(-1.5, 3.0))
        G = np.log(rng.dirichlet(np.ones(N) * 2.0))
        g0 = G + rng.uniform(0, 0.5) * rng.standard_normal(N)
        w0 = rng.dirichlet(np.ones(N) * 2.0)
        f0 = rng.uniform(0, 2e-3) * rng.standard_normal(M)
        params = dict(LBFGS_DEFAULTS, linesearch=int(rng.choice([0, 1, 2, 3])), max_iterations=int(rng.integers(1, 60)),
                      epsilon=float(rng.choice([1e-6, 1e-12])), delta=float(rng.choice([0.0, 1e-6])),
                      max_linesearch=int(rng.choice([3, 100])))
        affine = rng.random() < 0.4
        off, sc = rng.normal(0, 1, M), rng.uniform(0.5, 2.0, M)
        engine = str(rng.choice(["0", "1"]))
        tag = "seed %d: M=%d N=%d theta=%.3g ls=%d it<=%d mls=%d engine=%s%s" % (
            seed, M, N, theta, params["linesearch"], params["max_iterations"], params["max_linesearch"], engine, " affine" if affine else "")
        os.environ["BIOEN_HIP_DEVICE_LS"] = engine
        try:
            with bioen_amd.Context(y, YT) as ctx:
                if affine:
                    ctx.set_affine(off, sc)
                x, w, info = ctx.opt_lbfgs_logw(g0, G, theta, params)
                yraw, yeff = ctx.last_average()
                codes[info.lbfgs_code] = codes.get(info.lbfgs_code, 0) + 1
                ref = y.dot(w)
                if not np.abs(yraw - ref).max() <= 1e-12 * np.abs(ref).max():
                    bad.append("%s: log-weights (status %d): last_average off by %.3g" % (tag, info.lbfgs_code, np.abs(yraw - ref).max() / np.abs(ref).max()))
                if affine and not np.abs(yeff - (off + sc * ref)).max() <= 1e-12 * np.abs(off + sc * ref).max():
                    bad.append("%s: log-weights: effective average off" % tag)
                if not affine:
                    fx, fw, finfo = ctx.opt_lbfgs_forces(f0, w0, theta, params)
                    yraw, _ = ctx.last_average()
                    codes[finfo.lbfgs_code] = codes.get(finfo.lbfgs_code, 0) + 1
                    ref = y.dot(fw)
                    if not np.abs(yraw - ref).max() <= 1e-12 * np.abs(ref).max():
                        bad.append("%s: forces (status %d): last_average off by %.3g" % (tag, finfo.lbfgs_code, np.abs(yraw - ref).max() / np.abs(ref).max()))
        except Exception as e:
            bad.append(tag + " EXCEPTION " + repr(e)[:200])
        finally:
            del os.environ["BIOEN_HIP_DEVICE_LS"]
    return bad, codes


if __name__ == "__main__":
    n = int(os.environ.get("SEEDS", "60"))
    bad, codes = run(int(os.environ.get("FIRST", "0")), n)
    print("seeds", n, "status codes seen", codes, "violations:", len(bad))
    for b in bad:
        print("  ", b)
