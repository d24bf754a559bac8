"""GPU tests at BASELINE.json's FULL sizes (config 3: N = 1e6 x M = 1024 log-weights, 8 thetas;
config 5: N = 1e6 x M = 512 forces), where the oracle would need minutes per evaluation: the
device results are checked through size-independent properties instead --

  * sparse weights pick single columns: yTilde . w must equal the columns read back (forward pass,
    every column index, exact to rounding);
  * the gradient of sampled structures against the closed form built from those columns and the
    device's own residual (adjoint pass);
  * linearity of the ensemble average, normalisation, gauge invariance, directional derivatives;
  * a batched theta series equals the single runs bit for bit, and the objective decreases.
"""
import numpy as np
import pytest

from conftest import require_reference

pytestmark = pytest.mark.gpu

LBFGS_DEFAULTS = dict(linesearch=2, max_iterations=5000, delta=1e-6, epsilon=1e-6, ftol=1e-5, gtol=0.9, wolfe=0.9,
                      past=10, max_linesearch=100)


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def _targets(M, seed=12345):
    rng = np.random.default_rng(seed)
    YTrue = rng.uniform(1, 10, M)
    sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
    YTilde = rng.normal(YTrue, sig_exp) / sig_exp
    return YTrue, sig_sim, sig_exp, YTilde


def _columns(ctx, cols):
    return np.hstack([ctx.read_ytilde(col0=int(j), cols=1) for j in cols])


def test_logw_full_size_properties():
    import bioen_amd
    M, N = 1024, 1000000
    YTrue, sig_sim, sig_exp, YTilde = _targets(M)
    rng = np.random.default_rng(99)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        cols = np.array([0, 1, 127, 128, 65535, 500000, 999871, 999998, 999999])
        Y = _columns(ctx, cols)                                   # (M, 9)
        # forward pass: sparse weights
        wts = rng.dirichlet(np.ones(cols.size))
        w = np.zeros(N)
        w[cols] = wts
        chi2, yave = ctx.chi_squared(w)
        expect = Y.dot(wts)
        assert np.abs(yave - expect).max() <= 4e-16 * np.abs(expect).max() * cols.size
        assert rel(chi2, 0.5 * np.sum((expect - YTilde) ** 2)) < 1e-13
        # linearity of the ensemble average in w
        w1, w2 = rng.dirichlet(np.ones(N)), rng.dirichlet(np.ones(N) * 0.3)
        y1, y2 = ctx.chi_squared(w1)[1], ctx.chi_squared(w2)[1]
        y12 = ctx.chi_squared(0.25 * w1 + 0.75 * w2)[1]
        assert np.abs(y12 - (0.25 * y1 + 0.75 * y2)).max() <= 1e-13 * np.abs(y12).max()
        # objective + gradient at a random point, non-uniform prior
        theta = 10.0
        G = np.log(rng.dirichlet(np.ones(N) * 2.0))
        g = G + 0.3 * rng.standard_normal(N)
        f, grad = ctx.logw_fdf(g, G, theta)
        wg, logs = ctx.logw_weights(g)
        assert abs(wg.sum() - 1.0) < 1e-12 and rel(logs, np.log(np.exp(g - g.max()).sum()) + g.max()) < 1e-14
        chi2g, ybar = ctx.chi_squared(wg)
        P = float(np.dot(wg, g - G))
        logs0 = np.log(np.exp(G - G.max()).sum()) + G.max()
        assert rel(f, theta * (P - logs + logs0) + chi2g) < 1e-12          # c_bioen_kernels_logw.c:124-147
        r = ybar - YTilde
        a = (Y - ybar[:, None]).T.dot(r)                                   # centred adjoint of the sampled columns
        expect_g = wg[cols] * (theta * ((g[cols] - G[cols]) - P) + a)      # :207-218
        assert np.abs(grad[cols] - expect_g).max() <= 1e-11 * np.abs(expect_g).max()
        assert abs(grad.sum()) < 1e-9 * np.abs(grad).sum()                 # gauge direction
        dirn = rng.standard_normal(N)
        h = 1e-4
        fp = ctx.logw_fdf(g + h * dirn, G, theta, need_grad=False)[0]
        fm = ctx.logw_fdf(g - h * dirn, G, theta, need_grad=False)[0]
        assert abs((fp - fm) / (2 * h) - grad.dot(dirn)) < 1e-5 * max(1.0, abs(grad.dot(dirn)))
        assert rel(ctx.logw_fdf(g + 3.7, G, theta, need_grad=False)[0], f) < 1e-12   # L(g + c) = L(g)

        # optimiser at full size: batched == single bit for bit, objective decreases, budget code
        G0 = np.zeros(N)
        f0 = ctx.logw_fdf(G0, G0, 100.0, need_grad=False)[0]
        short = dict(LBFGS_DEFAULTS, max_iterations=12)
        thetas = [100.0, 3.0]
        res, wopt, infos = ctx.opt_lbfgs_logw_batch(thetas, G0, G0, short, max_batch=8)
        for k, th in enumerate(thetas):
            gs, ws, info = ctx.opt_lbfgs_logw(G0, G0, th, short)
            assert info.lbfgs_code == -997 and infos[k].lbfgs_code == -997 and info.iterations == 12
            assert infos[k].fmin == info.fmin and np.array_equal(res[k], gs) and np.array_equal(wopt[k], ws)
            assert abs(ws.sum() - 1.0) < 1e-12
            # fmin is the objective at the returned point
            assert rel(ctx.logw_fdf(gs, G0, th, need_grad=False)[0], info.fmin) < 1e-13
        assert infos[0].fmin < f0
        assert rel(infos[0].fmin, thetas[0] * infos[0].kl + infos[0].chi2) < 1e-12

        # GSL-style minimizers with all vectors resident: a few bfgs2 / conjugate_pr iterations at full size
        for alg in ("bfgs2", "conjugate_pr"):
            gg, wg2, ginfo = ctx.opt_gsl_logw(G0, G0, 100.0, alg, dict(step_size=0.01, tol=0.001, max_iterations=4))
            assert ginfo.lbfgs_code in (0, -2, 27) and ginfo.iterations <= 4
            assert ginfo.fmin < f0 and abs(wg2.sum() - 1.0) < 1e-12
            assert rel(ctx.logw_fdf(gg, G0, 100.0, need_grad=False)[0], ginfo.fmin) < 1e-13
            assert rel(ginfo.fmin, 100.0 * ginfo.kl + ginfo.chi2) < 1e-12


@pytest.mark.parametrize("M", [512, 1024, 1056])      # 256-thread strips, 512-thread strips, streaming passes
def test_forces_full_size_properties(M):
    import bioen_amd
    N = 1000000
    YTrue, sig_sim, sig_exp, YTilde = _targets(M, seed=777)
    rng = np.random.default_rng(5)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=777) as ctx:
        cols = np.array([0, 129, 4097, 333333, 999999])
        Y = _columns(ctx, cols)
        w0 = rng.dirichlet(np.ones(N) * 2.0)
        forces = 1e-3 * rng.standard_normal(M)
        theta = 10.0
        w = ctx.forces_weights(forces, w0)
        assert abs(w.sum() - 1.0) < 1e-12
        x = forces.dot(Y)                                                  # x_j = sum_i f_i yTilde_ij
        ratio = (w[cols] / w0[cols]) / (w[cols[0]] / w0[cols[0]])          # w_j / w0_j  ~  exp(x_j)
        assert np.abs(ratio - np.exp(x - x[0])).max() <= 1e-11 * ratio.max()
        f, grad = ctx.forces_fdf(forces, w0, theta)
        chi2, ybar = ctx.chi_squared(w)
        kl = float(np.sum(w * np.log(w / w0)))
        assert rel(f, theta * kl + chi2) < 1e-11                           # c_bioen_kernels_forces.c:226-270
        dirn = rng.standard_normal(M)
        h = 1e-6
        fp = ctx.forces_fdf(forces + h * dirn, w0, theta, need_grad=False)[0]
        fm = ctx.forces_fdf(forces - h * dirn, w0, theta, need_grad=False)[0]
        assert abs((fp - fm) / (2 * h) - grad.dot(dirn)) < 2e-5 * max(1.0, abs(grad.dot(dirn)))
        # theta series as one batch == single runs, bit for bit
        short = dict(LBFGS_DEFAULTS, max_iterations=6)
        thetas = [100.0, 10.0]
        res, wopt, infos = ctx.opt_lbfgs_forces_batch(thetas, np.zeros(M), w0, short)
        for k, th in enumerate(thetas):
            fs, ws, info = ctx.opt_lbfgs_forces(np.zeros(M), w0, th, short)
            assert infos[k].lbfgs_code == info.lbfgs_code and infos[k].fmin == info.fmin
            assert np.array_equal(res[k], fs) and np.array_equal(wopt[k], ws)
            assert rel(info.fmin, th * info.kl + info.chi2) < 1e-11


def _generated_columns(bioen_amd, M, N, targets, seed, cols, world=64):
    """Columns of the synthetic N-column matrix WITHOUT the big context: the generator is counter-based per (row, global
    column), so the rank of a `world`-way decomposition that owns a column generates it at a small LOCAL index."""
    out = np.empty((M, len(cols)))
    per = -(-(-(-N // world)) // 128) * 128                     # api.hip: segment_geometry (64 ranks: one segment each)
    for k, j in enumerate(cols):
        r = int(j) // per
        with bioen_amd.Context.synthetic(M, N, *targets, seed=seed, rank=r, world=world) as part:
            assert part.col0 <= j < part.col0 + part.n_local and part.n_local * M < 2 ** 31
            out[:, k] = part.read_ytilde(col0=int(j) - part.col0, cols=1)[:, 0]
    return out


@pytest.mark.parametrize("copies", [1, 2])
def test_logw_beyond_2_pow_32_matrix_elements(copies, monkeypatch):
    """Maximum sizes: M x N = 1024 x 4.3e6 = 4.4e9 elements (35 GB a copy) -- every element offset beyond 2^31 and 2^32
    has to come out of 64-bit index arithmetic, in the row-major matrix, in both strip copies, in the read-back and in
    the passes.  The columns on either side of those boundaries are checked against the generator run at small local
    indices (another decomposition's ranks), the passes against those columns.  Both forms: ONE strip copy (r06: the
    default of a matrix this large -- the segments' strips interleaved, the adjoint through the LDS-image kernel) and two
    (BIOEN_HIP_ONE_COPY=0: the column-sum order copy and its kernel)."""
    import bioen_amd
    if copies == 2:
        monkeypatch.setenv("BIOEN_HIP_ONE_COPY", "0")
    M, N = 1024, 4300032
    assert M * N > 2 ** 32
    targets = _targets(M)
    YTilde = targets[3]
    rng = np.random.default_rng(31)
    # row-major: offset = row * ld + col crosses 2^31 at row 499 and 2^32 at row 998; strip-major: strip * 16384 crosses
    # them at columns 2^21 and 2^22
    cols = np.array([0, 2 ** 21 - 1, 2 ** 21, 2 ** 22 - 1, 2 ** 22, 2 ** 22 + 12345, N - 2, N - 1])
    Y = _generated_columns(bioen_amd, M, N, targets, 12345, cols)
    with bioen_amd.Context.synthetic(M, N, *targets, seed=12345) as ctx:
        forms, nbytes = ctx.footprint()
        assert "rowmajor" in forms and nbytes >= 8 * M * N
        assert np.array_equal(_columns(ctx, cols), Y)                       # read-back of the row-major matrix
        wts = rng.dirichlet(np.ones(cols.size))
        w = np.zeros(N)
        w[cols] = wts
        chi2, yave = ctx.chi_squared(w)                                     # builds the row-sum order copy
        expect = Y.dot(wts)
        assert np.abs(yave - expect).max() <= 4e-16 * np.abs(expect).max() * cols.size
        assert rel(chi2, 0.5 * np.sum((expect - YTilde) ** 2)) < 1e-13
        theta = 10.0
        G = np.log(rng.dirichlet(np.ones(N) * 2.0))
        g = G + 0.3 * rng.standard_normal(N)
        f, grad = ctx.logw_fdf(g, G, theta)                                 # ... and the column-sum order copy
        forms, _ = ctx.footprint()
        assert forms == ({"strips"} if copies == 1 else {"strips", "strips_colsum"})    # the copies have replaced the matrix
        assert ctx.layout()["interleave"] == 8
        assert np.array_equal(_columns(ctx, cols), Y)                       # read-back out of the strip copy
        wg, logs = ctx.logw_weights(g)
        chi2g, ybar = ctx.chi_squared(wg)
        P = float(np.dot(wg, g - G))
        logs0 = np.log(np.exp(G - G.max()).sum()) + G.max()
        assert rel(f, theta * (P - logs + logs0) + chi2g) < 1e-12
        a = (Y - ybar[:, None]).T.dot(ybar - YTilde)
        expect_g = wg[cols] * (theta * ((g[cols] - G[cols]) - P) + a)
        assert np.abs(grad[cols] - expect_g).max() <= 1e-11 * np.abs(expect_g).max()
        assert abs(grad.sum()) < 1e-9 * np.abs(grad).sum()
        # a short batched run on it: equal to the single runs bit for bit
        G0 = np.zeros(N)
        short = dict(LBFGS_DEFAULTS, max_iterations=3)
        res, wopt, infos = ctx.opt_lbfgs_logw_batch([100.0, 3.0], G0, G0, short, max_batch=8)
        gs, ws, info = ctx.opt_lbfgs_logw(G0, G0, 3.0, short)
        assert infos[1].fmin == info.fmin and np.array_equal(res[1], gs) and np.array_equal(wopt[1], ws)
        assert abs(ws.sum() - 1.0) < 1e-12


def test_forces_beyond_2_pow_32_matrix_elements():
    """The same for the forces method's strip passes: M x N = 512 x 8.6e6 (35 GB, one copy)."""
    import bioen_amd
    M, N = 512, 8600064
    assert M * N > 2 ** 32
    targets = _targets(M, seed=777)
    rng = np.random.default_rng(6)
    cols = np.array([0, 2 ** 22 - 1, 2 ** 22, 2 ** 23 - 1, 2 ** 23, N - 1])  # strip * 8192 crosses 2^31 / 2^32 at 2^22 / 2^23
    Y = _generated_columns(bioen_amd, M, N, targets, 777, cols)
    with bioen_amd.Context.synthetic(M, N, *targets, seed=777) as ctx:
        w0 = rng.dirichlet(np.ones(N) * 2.0)
        forces = 1e-3 * rng.standard_normal(M)
        theta = 10.0
        w = ctx.forces_weights(forces, w0)
        assert abs(w.sum() - 1.0) < 1e-12
        x = forces.dot(Y)
        ratio = (w[cols] / w0[cols]) / (w[cols[0]] / w0[cols[0]])
        assert np.abs(ratio - np.exp(x - x[0])).max() <= 1e-11 * ratio.max()
        f, grad = ctx.forces_fdf(forces, w0, theta)                          # the strip passes
        assert np.array_equal(_columns(ctx, cols), Y)
        chi2, ybar = ctx.chi_squared(w)
        kl = float(np.sum(w * np.log(w / w0)))
        assert rel(f, theta * kl + chi2) < 1e-11
        dirn = rng.standard_normal(M)
        h = 1e-6
        fp = ctx.forces_fdf(forces + h * dirn, w0, theta, need_grad=False)[0]
        fm = ctx.forces_fdf(forces - h * dirn, w0, theta, need_grad=False)[0]
        assert abs((fp - fm) / (2 * h) - grad.dot(dirn)) < 2e-5 * max(1.0, abs(grad.dot(dirn)))
        # the gradient's rows against the sampled columns cannot be isolated (every column enters every row): a sparse
        # prior does it -- w0 concentrated on the sampled columns makes them the whole ensemble
        w0s = np.full(N, 1e-300)
        w0s[cols] = 1.0 / cols.size
        ws = ctx.forces_weights(forces, w0s)
        es = np.exp(x - x.max()) / np.exp(x - x.max()).sum()
        assert np.abs(ws[cols] - es).max() <= 1e-12
        fs, grads = ctx.forces_fdf(forces, w0s, theta)
        ybs = Y.dot(es)
        t = (theta * (1.0 + np.log(es * cols.size)) + Y.T.dot(ybs - targets[3])) * es
        expect = (Y - ybs[:, None]).dot(t)                                   # c_bioen_kernels_forces.c:280-340
        assert np.abs(grads - expect).max() <= 1e-9 * np.abs(expect).max()


def test_deer_nuisance_series_at_config4_scale():
    """BASELINE config 4: DEER refinement with a modulation-depth nuisance parameter, N = 5e5
    rotamers x M = 205 time points.  The matrix F~ = (F - 1)/sigma is uploaded once;
    every refit of m is one GEMV on the device plus a closed-form 1-D least-squares step.
    Checked through properties: the data were generated with m_true, so the refits must move m
    from the start value towards it; the affine model must equal an explicitly rebuilt matrix on a
    column sample; the joint objective L(w, m) must not increase from one refit to the next
    (alternating minimisation: the refit lowers chi^2 at fixed w, the optimiser lowers L at fixed m)."""
    import bioen_amd
    from bioen_amd import nuisance
    from bench import deer_inputs          # the workload SURVEY 8(d) names: Fresnel-form traces on the 205-point time axis of
    N = 500000                             # exp-370-292-signal-deer.dat (tests/golden/deer_exp_370_292.npz), sigma = 0.01
    rng = np.random.default_rng(2024)
    sigma = 0.01
    Ft, YT_measured, off = deer_inputs(N, 2024, sigma)
    M = Ft.shape[0]
    assert M == 205
    m_true = 0.23
    w_true = rng.dirichlet(np.ones(N) * 0.5)
    # targets generated WITH m_true (so that the refits have something to recover): Y = 1 - m + m F.w + noise
    YT = off + m_true * Ft.dot(w_true) + rng.standard_normal(M)
    G = np.zeros(N)
    with bioen_amd.Context(Ft, YT) as ctx:
        # affine model == explicit matrix, on the objective (value) and on sampled gradient entries
        m0 = 0.15
        ctx.set_affine(off, np.full(M, m0))
        g = 0.2 * rng.standard_normal(N)
        f, grad = ctx.logw_fdf(g, G, 50.0)
        w, logs = ctx.logw_weights(g)
        ybar_eff = off + m0 * Ft.dot(w)
        chi2 = 0.5 * np.sum((ybar_eff - YT) ** 2)
        P = float(np.dot(w, g - G))
        assert rel(f, 50.0 * (P - logs + np.log(N)) + chi2) < 1e-11
        cols = np.array([0, 77, 123456, 499999])
        Yeff = off[:, None] + m0 * Ft[:, cols]
        a = (Yeff - ybar_eff[:, None]).T.dot(ybar_eff - YT)
        expect = w[cols] * (50.0 * ((g[cols] - G[cols]) - P) + a)
        assert np.abs(grad[cols] - expect).max() <= 1e-10 * np.abs(expect).max()
        # the series: 2 thetas x 4 refits, yaml-default L-BFGS
        res = nuisance.series(ctx, [100.0, 10.0], G, G, LBFGS_DEFAULTS, YT, row_offset=off, scale0=m0, iterations=4)
    for r in res:
        fm = [s["fmin"] for s in r["trace"]]
        # every optimisation restarts cold (procedure.py:46,66) and ends on the yaml-default plateau test
        # (delta = 1e-6 over 10 iterations), whose end points scatter by ~1e-4 relative
        assert all(b <= a * (1 + 3e-4) for a, b in zip(fm, fm[1:])), fm
        assert fm[-1] <= fm[0]
        assert abs(r["w"].sum() - 1.0) < 1e-12
    m_fit = res[-1]["scales"][0]
    assert abs(m_fit - m_true) < abs(m0 - m_true) and abs(m_fit - m_true) < 0.03
    # ... and against the MEASURED trace (the bench's DEER record): the alternation still descends, the depth stays physical
    with bioen_amd.Context(Ft, YT_measured) as ctx:
        res = nuisance.series(ctx, [100.0], G, G, LBFGS_DEFAULTS, YT_measured, row_offset=off, scale0=0.15, iterations=3)
    fm = [s_["fmin"] for s_ in res[0]["trace"]]
    assert all(b <= a * (1 + 3e-4) for a, b in zip(fm, fm[1:])), fm
    assert 0.0 < res[0]["scales"][0] < 1.0


# ---------------------------------------------------------------------------------------
# "bitwise reproducible" means exactly this: the bench workload (BASELINE configs[2]; generator seed, sizes,
# thetas and settings of bench.py) gives the same per-theta iteration / evaluation counts and the same fmin
# to the last digit on every box and in every run of ONE build.  The constants belong to the kernel sources
# at this commit (the sums' fixed reduction shapes are part of them): a change of a reduction order moves
# them and must update them here, together with profiles/.
# ---------------------------------------------------------------------------------------
BENCH_PINNED = [   # theta, iterations, evaluations, fmin
    # r06: ONE strip copy is the default of a matrix this large (include/bioen_hip.h: bioen_hip_ctx_set_one_copy) -- the
    # adjoint sums over rows in the LDS-image kernel's order, last bits differ from the two-copy form below.  Canonical
    # 8-segment reduction shape: these are the bits of 1, 2, 4 AND 8 GPUs.
    (1000.0, 8, 13, 502.23205525980626),
    (316.2277660168379, 76, 127, 477.09441969228806),
    (100.0, 43, 63, 411.9475274073285),
    (31.622776601683793, 203, 235, 291.058383858162),
    (10.0, 324, 368, 181.45436663275478),
    (3.1622776601683795, 367, 407, 133.27496797041636),
    (1.0, 286, 323, 116.9335766404019),
    (0.31622776601683794, 471, 530, 111.56954022348951),
]
BENCH_PINNED_TWO_COPIES = [   # BIOEN_HIP_ONE_COPY=0: the r05 default, bit for bit what r05 pinned
    # (r02-r04, one-GPU shape: 1605 iterations, the smallest theta stopping on an early plateau at 111.626; sharded runs took 1702-1758.)
    (1000.0, 8, 13, 502.2320552598044),
    (316.2277660168379, 76, 129, 477.094413061891),
    (100.0, 43, 63, 411.9475274073507),
    (31.622776601683793, 217, 249, 291.0563472174132),
    (10.0, 313, 364, 181.4598549770388),
    (3.1622776601683795, 333, 373, 133.3020042080253),
    (1.0, 299, 336, 116.93078839671978),
    (0.31622776601683794, 474, 518, 111.5691402586845),
]


def test_bench_workload_is_pinned(monkeypatch):
    import bioen_amd
    from bioen_amd import sweep
    M, N = 1024, 1000000
    YTrue, sig_sim, sig_exp, YTilde = _targets(M)
    thetas = np.logspace(3, -0.5, 8)
    G = np.zeros(N)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        res = sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS, max_batch=8)
        assert ctx.footprint()[0] == {"strips"} and ctx.layout()["one_copy"] == 1       # 8.2 GB resident, not 16.4
        again = sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS, max_batch=3)     # another batch schedule, same bits
        import os
        os.environ["BIOEN_HIP_DEVICE_LS"] = "1"      # the device-resident engine (the default below 4 GB per round): same bits
        try:
            dev = sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS, max_batch=8)
        finally:
            os.environ.pop("BIOEN_HIP_DEVICE_LS", None)
    assert sum(r["iterations"] for r in res) == sum(p[1] for p in BENCH_PINNED) == 1778
    for r, r2, (theta, it, ev, fmin) in zip(res, again, BENCH_PINNED):
        assert rel(r["theta"], theta) < 1e-15 and r["code"] in (0, 1)
        assert (r["iterations"], r["evaluations"]) == (it, ev), (theta, r["iterations"], r["evaluations"])
        assert r["fmin"] == fmin, (theta, repr(r["fmin"]))
        assert (r2["iterations"], r2["evaluations"], r2["fmin"]) == (it, ev, fmin)
    for r, r3 in zip(res, dev):
        assert (r3["iterations"], r3["evaluations"], r3["fmin"], r3["chi2"], r3["S"]) == \
               (r["iterations"], r["evaluations"], r["fmin"], r["chi2"], r["S"])
        assert np.array_equal(r["w"], r3["w"])
    monkeypatch.setenv("BIOEN_HIP_ONE_COPY", "0")     # two copies, as until r05: its pinned bits, and the same minima to the plateau stops' spread
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        two = sweep.sweep_log_weights(ctx, thetas, G, G, LBFGS_DEFAULTS, max_batch=8)
        assert ctx.footprint()[0] == {"strips", "strips_colsum"}
    for r, (theta, it, ev, fmin) in zip(two, BENCH_PINNED_TWO_COPIES):
        assert (r["iterations"], r["evaluations"], r["fmin"]) == (it, ev, fmin), (theta, r["iterations"], r["evaluations"], repr(r["fmin"]))
    for r, r2 in zip(res, two):
        assert rel(r["fmin"], r2["fmin"]) < 3e-4


_AT_OPTIMUM = (0, -998, -1000, -1001)      # epsilon test | line search out of trials / below min_step / rounding errors


@pytest.mark.parametrize("prior,M,N", [("uniform", 256, 100000), ("random", 256, 100000),
                                       ("uniform", 1024, 20000),      # the headline's 16-wave strip geometry
                                       ("uniform", 205, 50000)])      # configs[3]'s row count (padded strips)
def test_configs1_converged_against_the_reference_binary(prior, M, N):
    """BASELINE configs[1] (N = 1e5 x M = 256) against the REFERENCE's own C + liblbfgs path (oracle/_ref, built from
    /root/reference in the build container; the .so travels with the repository), both run to convergence
    (epsilon = 1e-9, delta = 0, past = 0): north_star's tolerances as they stand -- 1e-6 on the negative
    log-posterior, 1e-5 max(w) on the weights -- at a size no golden fixture reaches.  Three thetas the reference
    converges on in seconds; the device solves them as one lock-step batch."""
    import bioen_amd
    from oracle import cpus
    R = require_reference()             # absent on a GPU box = failure, not skip (conftest.py)
    conv = dict(LBFGS_DEFAULTS, epsilon=1e-9, delta=0.0, past=0, max_iterations=200000)
    thetas = [316.0, 100.0, 31.6] if (M, N) == (256, 100000) else [316.0, 100.0]
    YTrue, sig_sim, sig_exp, YTilde = _targets(M)
    R.set_fast_openmp_flag(0)           # serial sums instead of OpenMP reductions: the same result on every run and box
    R.omp_set_num_threads(cpus.usable_cpus())
    if prior == "uniform":
        G = np.zeros(N)
        g0 = G
        strict = [(conv, thetas)]
    else:
        # Non-uniform reference weights and a start away from them: the regime the reference's tests never enter.  Here
        # |x| ~ 600, so the gradient test |g| <= epsilon max(1, |x|) stops BOTH codes on a slope: at epsilon = 1e-9 the
        # negative log-posteriors of theta = 31.6 end 1.2e-6 apart and the weights of theta = 100 1.02e-5 max(w).  What is
        # pinned at THAT setting is asserted as it is (below: both ends satisfy the stopping rule, and the device's
        # minimum is not above the reference's); north_star's 1e-6 / 1e-5 are asserted where the stopping rule lets
        # both codes reach the optimum: epsilon = 1e-10, theta = 316 and 100.
        rng = np.random.default_rng(99)
        G = np.log(rng.gamma(2.0, 1.0, N))
        G -= G.max()
        g0 = G + 0.3 * rng.standard_normal(N)
        strict = [(dict(conv, epsilon=1e-10), thetas[:2])]
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        yT = np.ascontiguousarray(ctx.read_ytilde())
        if prior != "uniform":
            res, w, infos = ctx.opt_lbfgs_logw_batch(thetas, g0, G, conv)
            for k, theta in enumerate(thetas):
                g_ref, fmin_ref, code_ref = R.opt_lbfgs_logw(g0, G, yT, YTilde, theta, conv)
                assert code_ref in _AT_OPTIMUM and infos[k].lbfgs_code in _AT_OPTIMUM, (theta, code_ref, infos[k].lbfgs_code)
                # (i) the device's minimum is not worse than the reference's beyond the tolerance
                assert infos[k].fmin <= fmin_ref * (1.0 + 1e-6), (theta, infos[k].fmin, fmin_ref)
                # (ii) both end points sit where liblbfgs' gradient test put them: |grad L| <= epsilon max(1, |x|) (for
                # status 0; a search that gave up at the rounding floor is within a small factor of it), evaluated by
                # the device at BOTH points -- and the reference's own end point is not a better one for the device's objective
                for x_end, code in ((res[k], infos[k].lbfgs_code), (g_ref, code_ref)):
                    f_end, grad_end = ctx.logw_fdf(x_end, G, theta)
                    slope = np.linalg.norm(grad_end) / max(1.0, np.linalg.norm(x_end))
                    assert slope <= (1.0 if code == 0 else 50.0) * 1.001 * conv["epsilon"], (theta, code, slope)
                f_at_ref, _ = ctx.logw_fdf(g_ref, G, theta)
                assert rel(f_at_ref, fmin_ref) < 1e-11          # the same objective on both sides, to rounding
        for cfg, ths in strict:
            res, w, infos = ctx.opt_lbfgs_logw_batch(ths, g0, G, cfg)
            for k, theta in enumerate(ths):
                g_ref, fmin_ref, code_ref = R.opt_lbfgs_logw(g0, G, yT, YTilde, theta, cfg)
                # converged (0) or stopped at the rounding floor of the line search (with fast_openmp = 1 the reference's
                # OpenMP reductions made even this status vary from run to run on the same inputs)
                assert code_ref in _AT_OPTIMUM and infos[k].lbfgs_code in _AT_OPTIMUM, (theta, code_ref, infos[k].lbfgs_code)
                assert rel(infos[k].fmin, fmin_ref) < 1e-6, (theta, infos[k].fmin, fmin_ref)
                w_ref = np.asarray(R.get_weights(g_ref)[0]).ravel()          # the reference's own softmax (_get_weights)
                assert abs(w_ref.sum() - 1.0) < 1e-9
                assert np.abs(w[k] - w_ref).max() <= 1e-5 * w_ref.max(), (theta, np.abs(w[k] - w_ref).max() / w_ref.max())


def test_ala5_shape_forces_series_against_the_reference_binary():
    """The ala5 notebook's workload shape -- N = 50001 structures x M = 28 observables, forces method, liblbfgs with the
    settings of examples/ala5_optimize/lbfgs_2.yaml, thetas of thetas2.dat (every eighth of the 80), each theta
    warm-started from the previous optimum as run_theta_series does -- on synthetic data, against the reference's
    _opt_lbfgs_forces fed the same chain: 1e-6 on the negative log-posterior, 1e-5 max(w) on the weights, as they stand."""
    import bioen_amd
    from oracle import cpus
    R = require_reference()             # absent on a GPU box = failure, not skip (conftest.py)
    from bench import ALA5_LBFGS
    N, M = 50001, 28
    thetas = np.logspace(5, -1, 80)[::8]
    YTrue, sig_sim, sig_exp, YTilde = _targets(M)
    w0 = np.full(N, 1.0 / N)
    R.set_fast_openmp_flag(0)
    R.omp_set_num_threads(cpus.usable_cpus())
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        yT = np.ascontiguousarray(ctx.read_ytilde())
        f_dev, f_ref = np.zeros(M), np.zeros(M)
        for theta in thetas:
            f_dev, w_dev, info = ctx.opt_lbfgs_forces(f_dev, w0, theta, ALA5_LBFGS)
            f_ref, fmin_ref, code_ref = R.opt_lbfgs_forces(f_ref, w0, yT, YTilde, theta, ALA5_LBFGS)
            # (on this synthetic ensemble the largest thetas sit at the rounding floor after a few iterations: the search
            # runs out of trials, -998, in the reference as on the device -- the real ala5 data do not do that)
            assert info.lbfgs_code in (0, 1, 2, -998) and code_ref in (0, 1, 2, -998), (theta, info.lbfgs_code, code_ref)
            assert rel(info.fmin, fmin_ref) < 1e-6, (theta, info.fmin, fmin_ref)
            w_ref = np.asarray(R.forces_weights(f_ref, w0, yT)).ravel()
            assert np.abs(w_dev - w_ref).max() <= 1e-5 * w_ref.max(), (theta, np.abs(w_dev - w_ref).max() / w_ref.max())
        # the cold-started series as one lock-step batch lands on the same minima
        res, w, infos = ctx.opt_lbfgs_forces_batch(thetas, np.zeros(M), w0, ALA5_LBFGS)
        for theta, i in zip(thetas, infos):
            f_c, fmin_c, code_c = R.opt_lbfgs_forces(np.zeros(M), w0, yT, YTilde, theta, ALA5_LBFGS)
            assert rel(i.fmin, fmin_c) < 1e-6, (theta, i.fmin, fmin_c)


@pytest.mark.parametrize("M,N", [(256, 100000), (512, 50000), (96, 30000), (1024, 20000), (600, 30000)])   # k_strip | k_strip2
def test_configs1_forces_converged_against_the_reference_binary(M, N):
    """The same size through the forces method: the reference's _opt_lbfgs_forces and the device's lock-step batch,
    both with epsilon = 1e-9, delta = 0, past = 0.  Both end at the rounding floor of the line search (-998) or on
    the gradient test; 1e-6 on the negative log-posterior and 1e-5 max(w) on the weights, as they stand."""
    import bioen_amd
    from oracle import cpus
    R = require_reference()             # absent on a GPU box = failure, not skip (conftest.py)
    conv = dict(LBFGS_DEFAULTS, epsilon=1e-9, delta=0.0, past=0, max_iterations=200000)
    thetas = [316.0, 100.0, 31.6]
    YTrue, sig_sim, sig_exp, YTilde = _targets(M)
    w0 = np.full(N, 1.0 / N)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        res, w, infos = ctx.opt_lbfgs_forces_batch(thetas, np.zeros(M), w0, conv)
        yT = np.ascontiguousarray(ctx.read_ytilde())
    R.set_fast_openmp_flag(0)           # serial sums instead of OpenMP reductions: the same result on every run and box
    R.omp_set_num_threads(cpus.usable_cpus())
    for k, theta in enumerate(thetas):
        f_ref, fmin_ref, code_ref = R.opt_lbfgs_forces(np.zeros(M), w0, yT, YTilde, theta, conv)
        assert code_ref in _AT_OPTIMUM and infos[k].lbfgs_code in _AT_OPTIMUM, (theta, code_ref, infos[k].lbfgs_code)
        assert rel(infos[k].fmin, fmin_ref) < 1e-6, (theta, infos[k].fmin, fmin_ref)
        w_ref = np.asarray(R.forces_weights(f_ref, w0, yT)).ravel()
        assert np.abs(w[k] - w_ref).max() <= 1e-5 * w_ref.max(), (theta, np.abs(w[k] - w_ref).max() / w_ref.max())


@pytest.mark.parametrize("M,N", [(256, 100000), (1024, 20000), (512, 50000), (205, 50000), (600, 30000), (1056, 12000)])
def test_objective_and_gradient_against_the_reference_binary(M, N):
    """One evaluation of both methods at random points with a non-uniform prior, device against the reference's C
    functions (_bioen_log_posterior_*, _grad_bioen_log_posterior_*): every matrix-pass variant (strip kernels with 4, 8,
    16 waves per strip and padded rows, the forces kernels with 64 and with 128 rows per wave; streaming kernels at
    M = 1056) at sizes the golden fixtures do not reach."""
    import bioen_amd
    from oracle import cpus
    R = require_reference()             # absent on a GPU box = failure, not skip (conftest.py)
    YTrue, sig_sim, sig_exp, YTilde = _targets(M)
    rng = np.random.default_rng(7 + M)
    G = np.log(rng.gamma(2.0, 1.0, N))
    G -= G.max()
    g = G + 0.5 * rng.standard_normal(N)
    w0 = rng.dirichlet(np.ones(N) * 2.0)
    forces = 1e-3 * rng.standard_normal(M)
    R.set_fast_openmp_flag(0)           # serial sums instead of OpenMP reductions: the same result on every run and box
    R.omp_set_num_threads(cpus.usable_cpus())
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        yT = np.ascontiguousarray(ctx.read_ytilde())
        for theta in (0.7, 40.0):
            f, grad = ctx.logw_fdf(g, G, theta)
            f_ref = R.logw_f(g, G, yT, YTilde, theta)
            grad_ref = np.asarray(R.logw_df(g, G, yT, YTilde, theta)).ravel()
            assert rel(f, f_ref) < 1e-12, (theta, f, f_ref)
            assert np.abs(grad - grad_ref).max() <= 1e-10 * np.abs(grad_ref).max(), theta
            ff, fgrad = ctx.forces_fdf(forces, w0, theta)
            ff_ref = R.forces_f(forces, w0, yT, YTilde, theta)
            fgrad_ref = np.asarray(R.forces_df(forces, w0, yT, YTilde, theta)).ravel()
            assert rel(ff, ff_ref) < 1e-12, (theta, ff, ff_ref)
            assert np.abs(fgrad - fgrad_ref).max() <= 1e-10 * np.abs(fgrad_ref).max(), theta


def test_configs4_shape_forces_endings_against_the_reference_golden():
    """how the forces-method runs of a configs[4]-shaped problem END -- status, minimum, iteration
    count per theta -- against what the REFERENCE's own binary does with the same inputs (tests/golden/
    forces_status_cfg4_M512xN100000.json, made by make_golden_forces_status.py from oracle/_ref: SURVEY 8(d)'s numpy
    stream, M = 512 x N = 1e5, the 8 thetas of the series, yaml-default liblbfgs, every summation mode x thread count x
    repetition recorded).

    Where the reference is unanimous (theta <= 31.6: the plateau test, status 1) the device must end with the same status
    after the same number of iterations where the reference's own count is unanimous too.  At theta >= 100 the reference is NOT a function of its inputs:
    the runs end where the decrease the line search still asks for (1/2 g^2 / (theta var) ~ 1e-15) lies below the rounding
    noise of the objective (f ~ 250: 5e-13), so 0 or -998 (or -1000) is decided by rounding, and the golden holds both
    for the same inputs (fast_openmp 0 / 1, 8 / 4 / 2 / 1 threads, repeated runs).  There the two sides are held to what IS
    determined: the same minimum to 1e-12, and an ending out of the rounding-floor set.  (profiles/
    r06_forces_status_probe_full.txt: the same at N = 1e6, with liblbfgs' own binary on the device's objective.)"""
    import json
    import os
    import bioen_amd
    from bench import survey_inputs
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "forces_status_cfg4_M512xN100000.json")) as fp:
        gold = json.load(fp)
    M, N = gold["M"], gold["N"]
    assert len(gold["per_theta"]) == 8
    y, YT = survey_inputs(M, N, gold["seed"])
    w0 = np.full(N, 1.0 / N)
    thetas = [p["theta"] for p in gold["per_theta"]]
    with bioen_amd.Context(y, YT) as ctx:
        _, _, infos = ctx.opt_lbfgs_forces_batch(thetas, np.zeros(M), w0, gold["lbfgs"], want_weights=False)
        for k in (0, 1, 2):                          # the large thetas one by one: the bits of the batch
            _, _, one = ctx.opt_lbfgs_forces(np.zeros(M), w0, thetas[k], gold["lbfgs"], want_weights=False)
            assert (one.lbfgs_code, one.iterations, one.evaluations, one.fmin) == \
                   (infos[k].lbfgs_code, infos[k].iterations, infos[k].evaluations, infos[k].fmin)
    floor = {0, -998, -1000, -1001}                  # converged | the line search's three ways of giving up at the rounding floor
    undetermined = 0
    for g, i in zip(gold["per_theta"], infos):
        codes = set(g["codes"])
        its = {r["iterations"] for r in g["runs"]}
        tol = max(1e-12, 10.0 * g["fmin_rel_spread"])      # (a plateau stop moves with the summation order: the reference's own spread)
        assert rel(i.fmin, g["fmin_min"]) <= tol, (g["theta"], i.fmin, g["fmin_min"], tol)
        if len(codes) == 1:
            assert i.lbfgs_code in codes, (g["theta"], i.lbfgs_code, codes)
            if len(its) == 1:
                assert i.iterations in its, (g["theta"], i.iterations, its)
        else:
            undetermined += 1
            assert g["theta"] >= 99.0 and codes <= floor and i.lbfgs_code in floor, (g["theta"], i.lbfgs_code, codes)
            assert rel(i.fmin, g["fmin_min"]) <= 1e-12
    assert undetermined == 3                         # theta = 1000, 316, 100: the reference's own coin flips (the golden's content)
