"""GPU parity tests for the reference's edge cases, device against the reference's own binary (oracle/_ref):

(a) forces method with reference weights w0 holding exact zeros and denormals and force vectors that drive weights below
    DBL_MIN -- the reference zeroes those terms of the prior and of the gradient
    (/root/reference/bioen/optimize/ext/c_bioen_kernels_forces.c:250-254, 320-328) -- on all pass families (k_strip with 2
    and 8 waves, k_strip2, row panels), K = 1 and K = 8, objective, gradient and L-BFGS;
(b) log-weights with a prior and a start whose weights span 1e-150 ... 1 as bioen.analyze manufactures them
    (`wopt[wopt == 0] = 1e-150`, /root/reference/bioen/analyze/procedure.py:37-38, 78-79, through getGs,
    /root/reference/bioen/optimize/log_weights.py:113-127: G = log w - log w[-1], down to -345 or up to +345);
(c) theta = 0 forces at BASELINE configs[1] size.

Tolerances: 1e-12 on the objective and 1e-6 / 1e-5 max(w) on converged runs, as everywhere.  The gradient is held to
1e-10 max|grad| wherever it is well conditioned, and otherwise to its CONDITION: both codes form
grad_i = sum_j (y_ij - ybar_i) t_j; the reference subtracts ybar_i element by element, the device's matrix-core product
works on operands centred on the targets Y_i and takes the difference (ybar_i - Y_i) T off afterwards, i.e.
its rounding error is a few ulp of  cond_i = sum_j |y_ij - Y_i| |t_j| + |ybar_i - Y_i| |T|.  Where the weights collapse
on one structure (or on a cluster of near-identical ones) while ybar is far from Y, the true gradient is many orders
below cond_i; there the device is held against an 80-bit evaluation of the closed form (gradient_ok): within 1e-10 max|grad| of it, or
no further from it than twice the REFERENCE's own distance (in the cluster regime the reference's gradient is itself
2e-9 ... 3e-8 max|grad| off the truth, the device's 1e-8 ... 2e-8; on well-conditioned points the device is at 1e-15, the
reference at 1e-13 ... 1e-14), or within 64 ulp of cond_i row by row."""
import numpy as np
import pytest

from conftest import LBFGS_DEFAULTS, LBFGS_CONV, require_reference

pytestmark = pytest.mark.gpu

DBL_MIN = 2.2250738585072014e-308
SHAPES = [(96, 20000), (512, 20000), (600, 20000), (1056, 12000)]      # k_strip 2 waves | 8 waves | k_strip2 | row panels


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def targets(M, seed=12345):
    rng = np.random.default_rng(seed)
    YTrue = rng.uniform(1, 10, M)
    sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
    return YTrue, sig_sim, sig_exp, rng.normal(YTrue, sig_exp) / sig_exp


def holed_w0(rng, N):
    w0 = rng.dirichlet(np.ones(N) * 2.0)
    w0[rng.choice(N, 50, replace=False)] = 0.0
    w0[rng.choice(N, 50, replace=False)] = 1e-310        # denormal
    w0[rng.choice(N, 5, replace=False)] = 4.9e-324       # the smallest one
    return w0


def forces_truth(y, YTilde, f, w0, theta):
    """closed form of F1-F3 in 80-bit arithmetic, with the reference's DBL_MIN guards; -> (L, grad, cond)"""
    L = np.longdouble
    yl, fl, w0l, Yl = y.astype(L), f.astype(L), w0.astype(L), YTilde.astype(L)
    x = fl @ yl
    e = w0l * np.exp(x - x.max())
    w = e / e.sum()
    ybar = yl @ w
    r = ybar - Yl
    b = r @ yl
    ok = (w >= DBL_MIN) & (w0l >= DBL_MIN)
    lw = np.where(ok, np.log(np.where(ok, w, 1)) - np.log(np.where(ok, w0l, 1)), L(0))
    t = (L(theta) * (1 + lw) + b) * w
    grad = (yl - ybar[:, None]) @ t
    cond = np.abs(yl - Yl[:, None]) @ np.abs(t) + np.abs(r) * abs(t.sum())
    obj = L(theta) * (lw * w).sum() + L(0.5) * (r * r).sum()
    return float(obj), grad.astype(np.float64), cond.astype(np.float64)


def gradient_ok(g_dev, g_ref, g_true, cond):
    """the device's gradient is within 1e-10 max|grad| of the 80-bit truth (the plain statement), or -- where the sum
    cancels -- no further from it than twice the reference's own distance, or within 64 ulp of its condition row by row"""
    err_dev, err_ref = np.abs(g_dev - g_true), np.abs(g_ref - g_true)
    return bool(err_dev.max() <= 1e-10 * np.abs(g_true).max() or err_dev.max() <= 2.0 * err_ref.max() or
                (err_dev <= 64 * 2.0 ** -52 * cond).all())


def cluster_problem(M, N, seed, ncluster=200, sigma_x=30.0):
    """200 near-identical structures on top of the softmax (weights of the same order), the bulk 660 +- 4 x 30 below:
    thousands of weights underflow to denormals or zero"""
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    rng = np.random.default_rng(seed)
    y = rng.normal(YTrue[:, None], sig_sim[:, None], (M, N)) / sig_exp[:, None]
    y[:, :ncluster] = y[:, :1] + 0.002 * rng.standard_normal((M, ncluster)) * (sig_sim / sig_exp)[:, None]
    f = y[:, 0] - y[:, ncluster:].mean(axis=1)
    f *= sigma_x / (f @ y)[ncluster:].std()
    w0 = holed_w0(rng, N)
    w0[3], w0[5] = 0.0, 1e-310                            # inside the cluster too
    return y, YTilde, f, w0


@pytest.mark.parametrize("M,N", SHAPES)
def test_forces_underflowing_weights_and_holed_prior_against_the_reference_binary(M, N):
    import bioen_amd
    from oracle import cpus
    R = require_reference()
    R.set_fast_openmp_flag(0)
    R.omp_set_num_threads(cpus.usable_cpus())
    ulp = 2.0 ** -52
    y, YTilde, f, w0 = cluster_problem(M, N, M)
    rng = np.random.default_rng(100 + M)
    with bioen_amd.Context(y, YTilde) as ctx:
        # cluster regime, K = 1; and the regime in which ONE structure takes the whole weight (x spread over +- 4 x 200)
        f_delta = 200.0 / (5.0 * np.sqrt(M)) * rng.standard_normal(M)
        for name, fv in (("cluster", f), ("delta", f_delta), ("small", 1e-3 * rng.standard_normal(M))):
            for theta in (0.0, 3.0):
                fd, gd = ctx.forces_fdf(fv, w0, theta)
                w_ref = np.asarray(R.forces_weights(fv, w0, y)).ravel()
                f_ref = R.forces_f(fv, w0, y, YTilde, theta)
                g_ref = np.asarray(R.forces_df(fv, w0, y, YTilde, theta)).ravel()
                if name != "small":
                    assert (w_ref == 0).sum() > 50 and ((w_ref > 0) & (w_ref < DBL_MIN)).sum() >= 1, name
                assert rel(fd, f_ref) < 1e-12, (name, theta, fd, f_ref)
                f_true, g_true, cond = forces_truth(y, YTilde, fv, w0, theta)
                assert rel(fd, f_true) < 1e-12
                err_dev, err_ref = np.abs(gd - g_true), np.abs(g_ref - g_true)
                print("M=%d %s theta=%g: |grad| %.3g, device %.2e, reference %.2e of it; device %.1f ulp of cond"
                      % (M, name, theta, np.abs(g_true).max(), err_dev.max() / np.abs(g_true).max(),
                         err_ref.max() / np.abs(g_true).max(), (err_dev / (ulp * cond)).max()))
                assert gradient_ok(gd, g_ref, g_true, cond), (name, theta)
                if name == "small":      # well conditioned: the plain statement
                    assert np.abs(gd - g_ref).max() <= 1e-10 * np.abs(g_ref).max()
        # K = 8: every column of the batch against the reference's single evaluation
        F = np.stack([f * s for s in (1.0, 0.9, 0.8, 0.5, 0.25, 1.05, 0.0, 0.6)])
        th = np.array([0.0, 3.0, 30.0, 0.5, 10.0, 1.0, 5.0, 100.0])
        fb, gb = ctx.forces_fdf_batch(F, w0, th)
        for a in range(8):
            f_true, g_true, cond = forces_truth(y, YTilde, F[a], w0, th[a])
            assert rel(fb[a], R.forces_f(F[a], w0, y, YTilde, th[a])) < 1e-12, a
            assert gradient_ok(gb[a], np.asarray(R.forces_df(F[a], w0, y, YTilde, th[a])).ravel(), g_true, cond), a
            f1, g1 = ctx.forces_fdf(F[a], w0, th[a])
            assert f1 == fb[a] and np.array_equal(g1, gb[a])          # a batch column = the single call, bit for bit


@pytest.mark.parametrize("M,N,thetas", [(96, 20000, (30.0, 3.0)), (512, 20000, (30.0,)), (600, 20000, (30.0,)),
                                        (1056, 12000, (30.0,))])
def test_forces_lbfgs_with_zeros_and_denormals_in_the_prior_against_the_reference_binary(M, N, thetas):
    import bioen_amd
    from oracle import cpus
    R = require_reference()
    R.set_fast_openmp_flag(0)
    R.omp_set_num_threads(cpus.usable_cpus())
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    w0 = holed_w0(np.random.default_rng(M), N)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        yT = np.ascontiguousarray(ctx.read_ytilde())
        for theta in thetas:
            fo, wo, info = ctx.opt_lbfgs_forces(np.zeros(M), w0, theta, LBFGS_CONV)
            f_ref, fmin_ref, code_ref = R.opt_lbfgs_forces(np.zeros(M), w0, yT, YTilde, theta, LBFGS_CONV)
            w_ref = np.asarray(R.forces_weights(f_ref, w0, yT)).ravel()
            assert info.lbfgs_code in (0, -998, -1000, -1001) and code_ref in (0, -998, -1000, -1001)
            assert rel(info.fmin, fmin_ref) < 1e-6, (theta, info.fmin, fmin_ref)
            assert np.abs(wo - w_ref).max() <= 1e-5 * w_ref.max(), (theta, np.abs(wo - w_ref).max() / w_ref.max())
            assert (wo[w0 == 0.0] == 0.0).all() and np.isfinite(wo).all() and abs(wo.sum() - 1.0) < 1e-12


def analyze_style(rng, N, last_tiny):
    w = rng.dirichlet(np.ones(N) * 2.0)
    w[rng.choice(N - 1, N // 20, replace=False)] = 1e-150      # procedure.py:37-38: zeros become 1e-150
    if last_tiny:
        w[-1] = 1e-150                                         # getGs subtracts log w[-1]: G up to +345
    return np.log(w) - np.log(w[-1])


@pytest.mark.parametrize("M,N", [(256, 100000), (1024, 20000)])
@pytest.mark.parametrize("last_tiny", [False, True])
def test_logw_analyze_style_log_weights_evaluation_against_the_reference_binary(M, N, last_tiny):
    import bioen_amd
    from oracle import cpus
    R = require_reference()
    R.set_fast_openmp_flag(0)
    R.omp_set_num_threads(cpus.usable_cpus())
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    rng = np.random.default_rng(5 + M + last_tiny)
    G, g0 = analyze_style(rng, N, last_tiny), analyze_style(rng, N, last_tiny)
    assert (G.min() < -300.0) if not last_tiny else (G.max() > 300.0)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        yT = np.ascontiguousarray(ctx.read_ytilde())
        for theta in (0.5, 20.0):
            f, grad = ctx.logw_fdf(g0, G, theta)
            f_ref = R.logw_f(g0, G, yT, YTilde, theta)
            grad_ref = np.asarray(R.logw_df(g0, G, yT, YTilde, theta)).ravel()
            assert rel(f, f_ref) < 1e-12, (theta, f, f_ref)
            assert np.abs(grad - grad_ref).max() <= 1e-10 * np.abs(grad_ref).max(), theta
        w, logs = ctx.logw_weights(g0)
        w_ref, s_ref = R.get_weights(g0)
        assert np.abs(w - np.asarray(w_ref).ravel()).max() <= 1e-13 * np.max(w_ref) and rel(logs, np.log(s_ref)) < 1e-13


@pytest.mark.parametrize("last_tiny,thetas", [(True, (100.0, 10.0)), (False, (10.0,))])
def test_logw_analyze_style_converged_against_the_reference_binary(last_tiny, thetas):
    """M = 64 x N = 4000, epsilon = 1e-12 (|x| ~ 2e4: at 1e-9 the gradient test stops both codes on a slope), where the
    reference's two line searches pin the optimum to 3e-8 / 2e-6 among themselves (tools/attic/edge_probe2.py)"""
    import bioen_amd
    from oracle import cpus
    R = require_reference()
    R.set_fast_openmp_flag(0)
    R.omp_set_num_threads(cpus.usable_cpus())
    M, N = 64, 4000
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    rng = np.random.default_rng(M + N)
    y = rng.normal(YTrue[:, None], sig_sim[:, None], (M, N)) / sig_exp[:, None]
    G, g0 = analyze_style(rng, N, last_tiny), analyze_style(rng, N, last_tiny)
    cfg = dict(LBFGS_CONV, epsilon=1e-12)
    with bioen_amd.Context(y, YTilde) as ctx:
        for theta in thetas:
            go, wo, info = ctx.opt_lbfgs_logw(g0, G, theta, cfg)
            g_ref, fmin_ref, code_ref = R.opt_lbfgs_logw(g0, G, y, YTilde, theta, cfg)
            w_ref = np.asarray(R.get_weights(g_ref)[0]).ravel()
            assert info.lbfgs_code in (0, -998, -1000, -1001) and code_ref in (0, -998, -1000, -1001)
            assert rel(info.fmin, fmin_ref) < 1e-6, (theta, info.fmin, fmin_ref)
            assert np.abs(wo - w_ref).max() <= 1e-5 * w_ref.max(), (theta, np.abs(wo - w_ref).max() / w_ref.max())


def test_forces_theta_zero_at_configs1_size_against_the_reference_binary():
    """theta = 0 (no prior): N = 1e5 columns span the 256 targets many times over, chi^2 -> 0 along a flat valley and no
    minimiser is pinned -- so: objective and gradient at 1e-12 / 1e-10, and the first 30 L-BFGS iterations (both codes
    report -997, LBFGSERR_MAXIMUMITERATION, at the cap) step for step: fmin to 1e-9, weights to 1e-8 max(w)."""
    import bioen_amd
    from oracle import cpus
    R = require_reference()
    R.set_fast_openmp_flag(0)
    R.omp_set_num_threads(cpus.usable_cpus())
    M, N = 256, 100000
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    rng = np.random.default_rng(3)
    w0 = np.full(N, 1.0 / N)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        yT = np.ascontiguousarray(ctx.read_ytilde())
        for scale in (0.0, 1e-3, 1e-2):
            forces = scale * rng.standard_normal(M)
            f, g = ctx.forces_fdf(forces, w0, 0.0)
            f_ref = R.forces_f(forces, w0, yT, YTilde, 0.0)
            g_ref = np.asarray(R.forces_df(forces, w0, yT, YTilde, 0.0)).ravel()
            assert rel(f, f_ref) < 1e-12 and np.abs(g - g_ref).max() <= 1e-10 * np.abs(g_ref).max(), scale
        cfg = dict(LBFGS_DEFAULTS, max_iterations=30, delta=0.0, past=0, epsilon=1e-12)
        fo, wo, info = ctx.opt_lbfgs_forces(np.zeros(M), w0, 0.0, cfg)
        f_r, fmin_ref, code_ref = R.opt_lbfgs_forces(np.zeros(M), w0, yT, YTilde, 0.0, cfg)
        w_ref = np.asarray(R.forces_weights(f_r, w0, yT)).ravel()
        assert info.lbfgs_code == code_ref == -997 and info.iterations == 30
        assert rel(info.fmin, fmin_ref) < 1e-9 and info.fmin < 0.02 * f_ref
        assert np.abs(wo - w_ref).max() <= 1e-8 * w_ref.max()
        assert np.abs(fo - f_r).max() <= 1e-8 * np.abs(f_r).max()


# ---- non-finite inputs ------------------------------------------------------------------------------------------------
from test_oracle import NON_FINITE          # the same cases the CPU suite runs through the restatement


@pytest.mark.parametrize("case", sorted(NON_FINITE))
def test_non_finite_inputs_end_as_in_the_reference_binary(case):
    """NaN / inf in any input: the reference's binary (liblbfgs built with -ffast-math: its start test lets NaN through as
    "already minimal") returns status 2, the start point and a non-finite fmin after one evaluation.  The device does the
    same -- it does not iterate on NaN until max_iterations, and nothing hangs."""
    import bioen_amd as hip
    R = require_reference()
    method, kw = NON_FINITE[case]
    params = dict(LBFGS_DEFAULTS, max_iterations=50)
    with hip.Context(kw["yTilde"], kw["YTilde"]) as ctx:
        if method == "logw":
            x_r, f_r, code_r = R.opt_lbfgs_logw(kw["g0"], kw["G"], kw["yTilde"], kw["YTilde"], kw["theta"], params)
            x_d, w_d, info = ctx.opt_lbfgs_logw(kw["g0"], kw["G"], kw["theta"], params)
            start = kw["g0"]
        else:
            x_r, f_r, code_r = R.opt_lbfgs_forces(kw["f0"], kw["w0"], kw["yTilde"], kw["YTilde"], kw["theta"], params)
            x_d, w_d, info = ctx.opt_lbfgs_forces(kw["f0"], kw["w0"], kw["theta"], params)
            start = kw["f0"]
    assert code_r == 2 and info.lbfgs_code == 2 and info.iterations == 0 and info.evaluations == 1
    assert not np.isfinite(info.fmin) and np.isnan(info.fmin) == np.isnan(f_r)
    assert np.array_equal(np.asarray(x_r).ravel(), start, equal_nan=True)
    assert np.array_equal(np.asarray(x_d).ravel(), start, equal_nan=True)


def test_a_non_finite_theta_in_a_batch_leaves_the_other_problems_alone():
    """One member of a lock-step batch ends at once on NaN; the others run on and return the bits of their single runs."""
    import bioen_amd as hip
    method, kw = NON_FINITE["logw theta NaN"]
    params = dict(LBFGS_DEFAULTS, max_iterations=40)
    thetas = [10.0, float("nan"), 1.0]
    with hip.Context(kw["yTilde"], kw["YTilde"]) as ctx:
        res, w, infos = ctx.opt_lbfgs_logw_batch(thetas, kw["g0"], kw["G"], params, max_batch=8)
        assert infos[1].lbfgs_code == 2 and infos[1].evaluations == 1 and not np.isfinite(infos[1].fmin)
        for k in (0, 2):
            gs, ws, info = ctx.opt_lbfgs_logw(kw["g0"], kw["G"], thetas[k], params)
            assert infos[k].fmin == info.fmin and infos[k].iterations == info.iterations > 0
            assert np.array_equal(res[k], gs) and np.array_equal(w[k], ws)


# ---- valid but degenerate inputs (tools/attic/odd_probe.py) ---------------------------------------------------------------------
def _odd_problem(M, N, seed=3):
    rng = np.random.default_rng(seed)
    YTrue = rng.uniform(1, 10, M)
    sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
    y = rng.normal(YTrue[:, None], sig_sim[:, None], (M, N)) / sig_exp[:, None]
    return y, rng.normal(YTrue, sig_exp) / sig_exp


def _odd_cases():
    y, YT = _odd_problem(64, 4000)
    y1, YT1 = _odd_problem(1, 500)
    yn1, YTn1 = _odd_problem(16, 1)
    same = y.copy()
    same[:, :] = same[:, :1]
    return {
        # name -> (yTilde, YTilde, theta, statuses the two methods end with)
        "theta = 0": (y, YT, 0.0, (0, 0)),
        "theta = 1e-300": (y, YT, 1e-300, (0, 0)),
        "theta = 1e-12": (y, YT, 1e-12, (0, 0)),
        "one observable": (y1, YT1, 1.0, (0, 0)),
        "one structure": (yn1, YTn1, 1.0, (2, 2)),                      # nothing to optimise: already minimal
        "identical structures": (same, YT, 1.0, (2, 2)),               # the gradient is exactly zero
        "zero matrix": (np.zeros_like(y), YT, 1.0, (2, 2)),
        "chi^2 near overflow": (y * 1e150, YT, 1.0, (-995, 2)),        # log-weights: the first search finds no descent
    }


ODD = _odd_cases()


@pytest.mark.parametrize("case", sorted(ODD))
def test_degenerate_but_valid_inputs_against_the_reference_binary(case):
    """Status, objective and weights of both methods on inputs at the edge of the domain -- prior switched off, a single
    observable or structure, structures that cannot be told apart, an objective of 1e303 -- equal the reference binary's."""
    import bioen_amd as hip
    R = require_reference()
    y, YT, theta, (code_l, code_f) = ODD[case]
    M, N = y.shape
    G, w0, f0 = np.zeros(N), np.full(N, 1.0 / N), np.zeros(M)
    params = dict(LBFGS_DEFAULTS, max_iterations=300)
    with hip.Context(y, YT) as ctx:
        g_d, w_d, info_l = ctx.opt_lbfgs_logw(G, G, theta, params)
        f_d, wf_d, info_f = ctx.opt_lbfgs_forces(f0, w0, theta, params)
    g_r, fmin_l, cl = R.opt_lbfgs_logw(G, G, y, YT, theta, params)
    f_r, fmin_f, cf = R.opt_lbfgs_forces(f0, w0, y, YT, theta, params)
    assert (cl, cf) == (code_l, code_f) and (info_l.lbfgs_code, info_f.lbfgs_code) == (code_l, code_f)
    # the objective to 1e-6 as everywhere -- in ABSOLUTE terms where it is the rounding residue of a perfect fit (theta -> 0)
    assert abs(info_l.fmin - fmin_l) <= 1e-6 * max(abs(fmin_l), 1e-6)
    assert abs(info_f.fmin - fmin_f) <= 1e-6 * max(abs(fmin_f), 1e-6)
    w_r = np.exp(g_r - g_r.max())
    w_r /= w_r.sum()
    assert np.abs(np.asarray(w_d).ravel() - w_r).max() <= 1e-5 * w_r.max()
    wf_r = np.asarray(R.forces_weights(f_r, w0, y)).ravel()
    assert np.abs(np.asarray(wf_d).ravel() - wf_r).max() <= 1e-5 * wf_r.max()


@pytest.mark.parametrize("engine", ["device", "host"])
def test_a_context_is_clean_after_a_run_on_non_finite_input(engine, monkeypatch):
    """A run that ended on NaN must leave nothing behind: the next runs and evaluations on the SAME context return the
    bits a fresh context returns.  (Until r04 the weights' hand-out scaled the zero padding of e by a non-finite
    1 / sum e: NaN for good, and 0 x NaN in the matrix passes poisoned every later evaluation in that batch slot.)"""
    import bioen_amd as hip
    monkeypatch.setenv("BIOEN_HIP_DEVICE_LS", "1" if engine == "device" else "0")
    method, kw = NON_FINITE["logw NaN in g0"]
    y, YT = kw["yTilde"], kw["YTilde"]
    M, N = y.shape
    G, w0, f0 = np.zeros(N), np.full(N, 1.0 / N), np.zeros(M)
    params = dict(LBFGS_DEFAULTS, max_iterations=40)
    thetas = [10.0, 1.0]

    def clean(ctx):
        a = ctx.opt_lbfgs_logw_batch(thetas, G, G, params)
        b = ctx.opt_lbfgs_forces_batch(thetas, f0, w0, params)
        f, grad = ctx.logw_fdf(G + 0.1, G, 3.0)
        ff, fgrad = ctx.forces_fdf(f0 + 1e-3, w0, 3.0)
        return (a[0].tobytes(), a[1].tobytes(), [i.fmin for i in a[2]], b[0].tobytes(), b[1].tobytes(), [i.fmin for i in b[2]],
                f, grad.tobytes(), ff, fgrad.tobytes())

    with hip.Context(y, YT) as ctx:
        fresh = clean(ctx)
    assert all(np.isfinite(v) for v in fresh[2] + fresh[5]) and np.isfinite(fresh[6]) and np.isfinite(fresh[8])
    with hip.Context(y, YT) as ctx:
        for name in sorted(NON_FINITE):
            m, k = NON_FINITE[name]
            if k["yTilde"] is not y or k["YTilde"] is not YT:
                continue                                   # (those cases need a context of their own)
            if m == "logw":
                _, _, info = ctx.opt_lbfgs_logw(k["g0"], k["G"], k["theta"], params)
                ctx.opt_lbfgs_logw_batch([10.0, k["theta"]], k["g0"], k["G"], params)
            else:
                _, _, info = ctx.opt_lbfgs_forces(k["f0"], k["w0"], k["theta"], params)
                ctx.opt_lbfgs_forces_batch([10.0, k["theta"]], k["f0"], k["w0"], params)
            assert info.lbfgs_code == 2, name
            assert clean(ctx) == fresh, name


def test_a_search_refused_for_a_non_descent_direction_returns_the_accepted_point():
    """liblbfgs refuses a search whose direction does not descend BEFORE evaluating anything (lbfgs.c:671-674): status -994,
    the accepted point and ITS objective.  The device learns the initial slope with the first trial's results -- that
    evaluation is neither counted nor returned (r04: fmin was the trial's 139.1 instead of 81.9; tools/fuzz_parity.py, seed 69)."""
    import bioen_amd as hip
    R = require_reference()
    rng = np.random.default_rng(1069)
    M, N = 205, 12345
    rng.choice(21); rng.choice(14)                           # (the draws of tools/fuzz_parity.py's seed 69 before the data)
    YTrue = rng.uniform(1, 10, M)
    sig = rng.uniform(0.05, 0.3, M) * YTrue
    y = rng.normal(YTrue[:, None], rng.uniform(0.2, 0.8) * YTrue[:, None], (M, N)) / sig[:, None]
    YT = rng.normal(YTrue, sig) / sig
    theta = float(10.0 ** rng.uniform(-2, 3))
    G = np.log(rng.dirichlet(np.ones(N) * rng.uniform(0.3, 3.0)) + 1e-300)
    g = G + rng.uniform(0.0, 1.0) * rng.standard_normal(N)
    params = dict(LBFGS_DEFAULTS, linesearch=1, max_iterations=9, past=0, delta=0.0, epsilon=1e-9)
    with hip.Context(y, YT) as ctx:
        x_d, w_d, info = ctx.opt_lbfgs_logw(g, G, theta, params)
        f_at = ctx.logw_fdf(x_d, G, theta, need_grad=False)[0]
    x_r, fmin_r, code_r = R.opt_lbfgs_logw(g, G, y, YT, theta, params)
    assert code_r == -994 and info.lbfgs_code == -994 and info.iterations == 2 and info.evaluations == 7
    assert rel(info.fmin, fmin_r) < 1e-9 and rel(f_at, info.fmin) < 1e-13       # the objective OF the returned point
    assert np.abs(x_d - np.asarray(x_r).ravel()).max() <= 1e-7 * np.abs(x_r).max()


def test_beyond_the_rounding_floor_a_degenerate_pair_ends_the_run_as_in_the_reference_binary():
    """An Armijo-only search with the gradient test set below the gradient's rounding noise walks on at the floor until a
    pair with y.s = 0 turns the two-loop recursion's direction into NaN.  The reference's binary (its -ffast-math build
    takes a NaN slope for "not a descent direction") ends with -994 and the last accepted point; so does the device --
    until r04 it carried the NaN into the forces and reported status 0 (tools/fuzz_batch.py, seed 23)."""
    import bioen_amd as hip
    R = require_reference()
    rng = np.random.default_rng(5023)
    M, N = 28, 5000
    rng.choice(8); rng.choice(4)                              # (the draws of tools/fuzz_batch.py's seed 23 before the data)
    YTrue = rng.uniform(1, 10, M)
    y = rng.normal(YTrue[:, None], 0.5 * YTrue[:, None], (M, N)) / (0.1 * YTrue[:, None])
    YT = rng.normal(YTrue, 0.1 * YTrue) / (0.1 * YTrue)
    nt = int(rng.integers(1, 15))
    thetas = 10.0 ** rng.uniform(-1.5, 3.0, nt)
    if rng.random() < 0.3 and nt > 1:
        thetas[rng.integers(0, nt)] = thetas[0]
    rng.integers(1, 9)
    rng.dirichlet(np.ones(N) * 2.0)
    w0 = rng.dirichlet(np.ones(N) * 2.0)
    theta = float(thetas[1])
    params = dict(LBFGS_DEFAULTS, linesearch=1, epsilon=1e-12, delta=0.0, past=0, max_iterations=60)
    with hip.Context(y, YT) as ctx:
        f_d, w_d, info = ctx.opt_lbfgs_forces(np.zeros(M), w0, theta, params)
        f_at = ctx.forces_fdf(f_d, w0, theta, need_grad=False)[0]
    f_r, fmin_r, code_r = R.opt_lbfgs_forces(np.zeros(M), w0, y, YT, theta, params)
    assert code_r == -994 and info.lbfgs_code == -994
    assert np.isfinite(f_d).all() and np.isfinite(w_d).all() and abs(w_d.sum() - 1.0) < 1e-12
    assert rel(info.fmin, fmin_r) < 1e-12 and rel(f_at, info.fmin) < 1e-13
    w_r = np.asarray(R.forces_weights(f_r, w0, y)).ravel()
    assert np.abs(w_d - w_r).max() <= 1e-5 * w_r.max()
