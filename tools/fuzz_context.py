"""Development aid (GPU box): randomised checks of the context's data paths -- assembly from raw observables (row-major and
structure-major), read-back of arbitrary blocks before and after the strip copies replace the matrix, the affine model
against an explicitly rebuilt matrix, a changed target -- against numpy.  SEEDS=n (default 40)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bioen_amd


def run(first, nseeds):
    bad = []
    for seed in range(first, first + nseeds):
        rng = np.random.default_rng(11000 + seed)
        M = int(rng.choice([1, 7, 16, 28, 65, 205, 512, 513, 1024, 1100]))
        N = int(rng.choice([1, 15, 129, 1000, 2049, 7777]))
        sim = rng.normal(5.0, 2.0, (M, N))
        exp = rng.normal(5.0, 1.0, M)
        err = rng.uniform(0.05, 0.5, M)
        tag = "seed %d: M=%d N=%d" % (seed, M, N)
        y = sim / err[:, None]
        YT = exp / err
        try:
            sm = rng.random() < 0.5
            with bioen_amd.Context.from_raw(np.ascontiguousarray(sim.T) if sm else sim, exp, err, structure_major=sm) as ctx:
                def block_ok(where):
                    r0 = int(rng.integers(0, M)); rows = int(rng.integers(1, M - r0 + 1))
                    c0 = int(rng.integers(0, N)); cols = int(rng.integers(1, N - c0 + 1))
                    got = ctx.read_ytilde(row0=r0, rows=rows, col0=c0, cols=cols)
                    if not np.array_equal(got, y[r0:r0 + rows, c0:c0 + cols]):
                        bad.append("%s: read-back of block (%d+%d, %d+%d) %s differs (structure_major=%s)" % (tag, r0, rows, c0, cols, where, sm))
                block_ok("of the row-major matrix")
                w = rng.dirichlet(np.ones(N))
                chi2, yave = ctx.chi_squared(w)
                ref = y.dot(w)
                if not np.abs(yave - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-300):
                    bad.append("%s: ensemble average off by %.3g" % (tag, np.abs(yave - ref).max() / np.abs(ref).max()))
                g = 0.3 * rng.standard_normal(N)
                G = np.zeros(N)
                f, grad = ctx.logw_fdf(g, G, 3.0)
                block_ok("after the evaluations")
                block_ok("after the evaluations")
                # affine model: off + sc * y  against a context on the rebuilt matrix
                off = rng.normal(0, 1, M); sc = rng.uniform(0.5, 2.0, M)
                ctx.set_affine(off, sc)
                fa, ga = ctx.logw_fdf(g, G, 3.0)
                ctx.set_affine(None, None)
                f2, g2 = ctx.logw_fdf(g, G, 3.0)
                if f2 != f or not np.array_equal(g2, grad):
                    bad.append("%s: the plain model does not return its bits after an affine model was removed" % tag)
                # a changed target
                YT2 = YT + rng.normal(0, 0.5, M)
                ctx.set_target(YT2)
                ft, gt = ctx.logw_fdf(g, G, 3.0)
            with bioen_amd.Context(off[:, None] + sc[:, None] * y, YT) as cb:
                fb, gb = cb.logw_fdf(g, G, 3.0)
            if not (abs(fa - fb) <= 1e-11 * abs(fb) and np.abs(ga - gb).max() <= 1e-9 * max(np.abs(gb).max(), 1e-2 * abs(fb))):
                bad.append("%s: affine model vs rebuilt matrix: f %.3g grad %.3g" % (tag, abs(fa - fb) / abs(fb), np.abs(ga - gb).max() / max(np.abs(gb).max(), 1e-300)))
            with bioen_amd.Context(y, YT2) as cc:
                fc, gc = cc.logw_fdf(g, G, 3.0)
            if not (abs(ft - fc) <= 1e-12 * abs(fc) and np.abs(gt - gc).max() <= 1e-10 * max(np.abs(gc).max(), 1e-2 * abs(fc))):
                bad.append("%s: changed target vs fresh context: f %.3g grad %.3g" % (tag, abs(ft - fc) / abs(fc), np.abs(gt - gc).max() / max(np.abs(gc).max(), 1e-300)))
        except Exception as e:
            bad.append(tag + " EXCEPTION " + repr(e)[:200])
    return bad


if __name__ == "__main__":
    n = int(os.environ.get("SEEDS", "40"))
    bad = run(int(os.environ.get("FIRST", "0")), n)
    print("seeds", n, "violations:", len(bad))
    for b in bad:
        print("  ", b)
