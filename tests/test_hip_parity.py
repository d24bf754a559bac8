"""GPU parity tests: the HIP path (through the C ABI, bioen_amd._lib.Context) against
(1) the committed golden vectors = values computed by the REFERENCE's own C code, and
(2) the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star): 1e-6 relative on the negative log-posterior,
1e-5 on the final weights; single evaluations are held to ~1e-12 (the reference's own
unit tests use 5e-14 / 5e-12 between its C and Python paths, test_func_gradient_logw.py:9-10).
"""
import numpy as np
import pytest

from conftest import (LOGW_GOLDEN, FORCES_GOLDEN, LBFGS_DEFAULTS, LBFGS_TIGHT, LBFGS_CONV, LBFGS_CONVMT,
                      load_golden)

pytestmark = pytest.mark.gpu

F_RTOL = 1e-12          # single evaluation of L
G_RTOL = 1e-10          # gradient, relative to its max-norm
FMIN_RTOL = 1e-6        # north_star: negative log-posterior at the optimum
W_RTOL = 1e-5           # north_star: final weights (relative to the largest weight)


@pytest.fixture(scope="module")
def hip():
    import bioen_amd
    assert bioen_amd.device_count() >= 1, "no MI355X visible"
    return bioen_amd


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def maxrel(a, b):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


# ---------------------------------------------------------------------------------------
# single evaluations against the reference's values
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", LOGW_GOLDEN)
def test_logw_f_grad_weights_vs_reference(hip, name):
    d = load_golden(name)
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        w, logs = ctx.logw_weights(d["GInit"])
        assert maxrel(w, d["w_init"]) < 1e-13
        assert rel(logs, np.log(d["s_init"])) < 1e-13 or abs(logs - np.log(d["s_init"])) < 1e-13
        for gkey, fkey, grkey in (("GInit", "f_init", "grad_init"), ("g_pert", "f_pert", "grad_pert")):
            f, grad = ctx.logw_fdf(d[gkey], d["G"], d["theta"])
            assert rel(f, d[fkey]) < F_RTOL, (name, gkey, f, d[fkey])
            assert maxrel(grad, d[grkey]) < G_RTOL, (name, gkey)
            f_only, none = ctx.logw_fdf(d[gkey], d["G"], d["theta"], need_grad=False)
            assert none is None and f_only == f          # f-only path = same kernels, same bits


@pytest.mark.parametrize("name", FORCES_GOLDEN)
def test_forces_f_grad_weights_vs_reference(hip, name):
    d = load_golden(name)
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        for xkey, wkey, fkey, grkey in (("forces_init", "w_init", "f_init", "grad_init"),
                                        ("forces_pert", "w_pert", "f_pert", "grad_pert")):
            w = ctx.forces_weights(d[xkey], d["w0"])
            assert maxrel(w, d[wkey]) < 1e-12
            f, grad = ctx.forces_fdf(d[xkey], d["w0"], d["theta"])
            assert rel(f, d[fkey]) < F_RTOL, (name, xkey, f, d[fkey])
            # the reference's own C-vs-Python tolerance for this gradient is 5e-8
            # (test_func_gradient_forces.py:10): it is a difference of O(1e4) terms
            assert maxrel(grad, d[grkey]) < 1e-9, (name, xkey)


def test_chi_squared_and_average(hip):
    d = load_golden("ref_data_potra_part_2_logw_M808xN10.npz")
    from oracle import oracle_binding as O
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        chi2, yave = ctx.chi_squared(d["w_init"])
    chi2_o, yave_o = O.chi_squared(d["w_init"], d["yTilde"], d["YTilde"])
    assert rel(chi2, chi2_o) < 1e-13
    assert maxrel(yave, yave_o) < 1e-13


# ---------------------------------------------------------------------------------------
# L-BFGS against the reference's liblbfgs runs
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", LOGW_GOLDEN)
@pytest.mark.parametrize("tag,params", [("def", LBFGS_DEFAULTS), ("tight", LBFGS_TIGHT)])
def test_logw_lbfgs_vs_reference(hip, name, tag, params):
    d = load_golden(name)
    code_ref = int(d["lbfgs_%s_code" % tag])
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        gopt, w, info = ctx.opt_lbfgs_logw(d["GInit"], d["G"], d["theta"], params)
        if code_ref not in (0, 1, 2):
            # the reference itself fails here (line search exhausted); only the status family is pinned
            assert info.lbfgs_code < 0 or info.lbfgs_code in (0, 1)
            return
        assert info.lbfgs_code in (0, 1, 2), info.lbfgs_code
        # with the yaml defaults the delta-test stops on a 1e-6 relative plateau, so two
        # trajectories that differ in the last bits may stop an iteration apart
        tol = FMIN_RTOL if tag == "tight" else 2e-5
        assert rel(info.fmin, float(d["lbfgs_%s_fmin" % tag])) < tol, (name, info.fmin)
        # weights: 1e-5 where the settings pin the optimum that well; where the reference and
        # its own restatement (different summation order) already stop further apart than
        # that (stored as *_wspread by make_golden.py) the bound is that spread
        wref = d["lbfgs_%s_wopt" % tag]
        wtol = max(W_RTOL, 3.0 * float(d["lbfgs_%s_wspread" % tag]))
        assert np.abs(w - wref).max() <= wtol * wref.max(), (name, np.abs(w - wref).max() / wref.max(), wtol)
        # self-consistency (test_find_opt_analytical_grad_logw.py:181-188): fmin == L(gopt)
        f_again, _ = ctx.logw_fdf(gopt, d["G"], d["theta"], need_grad=False)
        assert rel(f_again, info.fmin) < 5e-14
        # known answer of the reference's regression test (*.ref, tolerance 1e-1, :134-147)
        if "ref_fmin_scipy_bfgs" in d:
            assert rel(info.fmin, float(d["ref_fmin_scipy_bfgs"])) < 1e-1
        assert info.iterations > 0 and info.evaluations >= info.iterations


@pytest.mark.parametrize("name", FORCES_GOLDEN)
def test_forces_lbfgs_vs_reference(hip, name):
    d = load_golden(name)
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        fopt, w, info = ctx.opt_lbfgs_forces(d["forces_init"], d["w0"], d["theta"], LBFGS_DEFAULTS)
        assert info.lbfgs_code in (0, 1, 2)
        assert rel(info.fmin, float(d["lbfgs_def_fmin"])) < 5e-6, (name, info.fmin)
        wref = d["lbfgs_def_wopt"]
        wtol = max(W_RTOL, 3.0 * float(d["lbfgs_def_wspread"]))
        assert np.abs(w - wref).max() <= wtol * wref.max(), (name, np.abs(w - wref).max() / wref.max(), wtol)
        f_again, _ = ctx.forces_fdf(fopt, d["w0"], d["theta"], need_grad=False)
        assert rel(f_again, info.fmin) < 5e-14
        if "ref_fmin_scipy_bfgs" in d:
            assert rel(info.fmin, float(d["ref_fmin_scipy_bfgs"])) < 1e-1


# ---------------------------------------------------------------------------------------
# north_star's numbers, un-widened: converged runs against the REFERENCE's converged runs
# ---------------------------------------------------------------------------------------
# Two fixtures are left out of the strict comparison, as the reference's own optimiser tests leave them out
# (test_find_opt_analytical_grad_logw.py:15-27: "(*) require specific tuning of parameters"; GSL conjugate_pr /
# bfgs end in NaN on the second, which tests/golden/gsl_*.npz reproduce): theta = 1e-3 resp. 808 observables on
# 100 structures make L(g) a staircase of plateaus (dL/dg_k ~ w_k vanishes wherever a weight has died), and
# the reference's liblbfgs gives up on one of them -- -998 / -1001 at 4386.96 resp. 4350.19 / 4427.69 while
# GSL bfgs2 reaches 4042.79 resp. 4159.16 and scipy's BFGS (the stored *.ref) 4313.60 resp. 4350.19.  A
# plateau is not an optimum, so there is no reference point to be within 1e-5 of; what is held there is the
# reference's own regression tolerance (1e-1 on fmin against *.ref, :134-147) and "no worse than the start".
REFERENCE_EXCLUDES = {"ref_data_potra_part_1_logw_M808xN80.npz", "ref_data_potra_part_2_logw_M808xN10.npz"}


def _converged_checks(name, tag, d, info, w, wref, theta):
    code_ref, fmin_ref = int(d["lbfgs_%s_code" % tag]), float(d["lbfgs_%s_fmin" % tag])
    # the reference ends on the epsilon test (0) or on an exhausted line search at the rounding floor
    # (-998 backtracking / -1001 More-Thuente: no representable decrease is left); both are "at the optimum"
    assert code_ref in (0, -998, -1001)
    assert info.lbfgs_code in (0, -998, -1001), (name, tag, info.lbfgs_code)
    if name in REFERENCE_EXCLUDES:
        assert rel(info.fmin, float(d["ref_fmin_scipy_bfgs"])) < 1e-1 and info.fmin < float(d["f_init"])
        return
    assert rel(info.fmin, fmin_ref) < FMIN_RTOL, (name, tag, info.fmin, fmin_ref)
    # theta = 0 (the two deer_test fixtures): no prior, 808 observables on 10 structures -- L is pinned,
    # the weights are not (a flat direction); everywhere else north_star's 1e-5 holds as is
    if theta > 0:
        assert np.abs(w - wref).max() <= W_RTOL * wref.max(), (name, tag, np.abs(w - wref).max() / wref.max())
    # secondary: RELATIVE agreement of the entries above 1e-3 max(w).  A weight's own relative
    # error is the absolute error of its log-weight, which an epsilon = 1e-9 gradient test pins only to
    # ~epsilon |x| / (w_k x curvature): the reference and its restatement already differ by 1.7e-4 on
    # synth_logw_M64xN2000 and 1.4e-5 on synth_logw_M37xN500 by this measure (2e-6 / 1e-7 by the one above)
    big = wref > 1e-3 * wref.max()
    if theta > 0:
        assert np.abs(w[big] / wref[big] - 1.0).max() <= 1e-3, (name, tag)


@pytest.mark.parametrize("name", LOGW_GOLDEN)
@pytest.mark.parametrize("tag,params", [("conv", LBFGS_CONV), ("convmt", LBFGS_CONVMT)])
def test_logw_converged_vs_reference(hip, name, tag, params):
    """max|w - w_ref| <= 1e-5 max(w_ref) and |fmin - fmin_ref| <= 1e-6 |fmin_ref| against the reference's own
    liblbfgs run, no conditioning term: epsilon = 1e-9, delta = 0, past = 0 leave no early exit."""
    d = load_golden(name)
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        gopt, w, info = ctx.opt_lbfgs_logw(d["GInit"], d["G"], d["theta"], params)
        f_again, _ = ctx.logw_fdf(gopt, d["G"], d["theta"], need_grad=False)
    _converged_checks(name, tag, d, info, w, d["lbfgs_%s_wopt" % tag], d["theta"])
    if info.lbfgs_code == 0:
        assert rel(f_again, info.fmin) < 5e-14
    else:       # after a failed search liblbfgs restores x but reports the last trial's f (lbfgs.c:470-479); on the
        # two plateau fixtures the reference itself fails on, that trial can sit a visible step away
        assert rel(f_again, info.fmin) < (1e-5 if name in REFERENCE_EXCLUDES else 1e-9)
    assert abs(w.sum() - 1.0) < 1e-12


@pytest.mark.parametrize("name", FORCES_GOLDEN)
@pytest.mark.parametrize("tag,params", [("conv", LBFGS_CONV), ("convmt", LBFGS_CONVMT)])
def test_forces_converged_vs_reference(hip, name, tag, params):
    d = load_golden(name)
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        fopt, w, info = ctx.opt_lbfgs_forces(d["forces_init"], d["w0"], d["theta"], params)
    if name == "ref_data_deer_test_forces_M808xN10.npz" and tag == "convmt":
        # theta = 0 and the reference's More-Thuente run stops 0.35 % above its backtracking run
        # (25538.77 vs 25449.08, -1001 after 14 iterations): hold to "no worse than the reference"
        assert info.lbfgs_code in (0, -998, -1001)
        assert info.fmin <= float(d["lbfgs_convmt_fmin"]) * (1 + 1e-9)
        assert info.fmin >= float(d["lbfgs_conv_fmin"]) * (1 - FMIN_RTOL)
        return
    _converged_checks(name, tag, d, info, w, d["lbfgs_%s_wopt" % tag], d["theta"])
    assert abs(w.sum() - 1.0) < 1e-12


# ... and the reference's runs under the other line searches / settings that make_golden.py stores
@pytest.mark.parametrize("name", LOGW_GOLDEN)
@pytest.mark.parametrize("tag,ls", [("mt", 0), ("strong", 3)])
def test_logw_other_linesearch_goldens_vs_reference(hip, name, tag, ls):
    """liblbfgs More-Thuente (linesearch 0) and backtracking strong-Wolfe (3) at the yaml defaults, against
    the reference's runs (lbfgs_mt_*, lbfgs_strong_*).  Default settings stop on the 1e-6 plateau, so
    fmin is pinned to the width of that plateau and the weights to the recorded conditioning of the
    stopping point; the converged tests above carry the un-widened claim."""
    d = load_golden(name)
    code_ref = int(d["lbfgs_%s_code" % tag])
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        gopt, w, info = ctx.opt_lbfgs_logw(d["GInit"], d["G"], d["theta"], dict(LBFGS_DEFAULTS, linesearch=ls))
    if code_ref not in (0, 1, 2):
        assert info.lbfgs_code < 0 or info.lbfgs_code in (0, 1)
        return
    assert info.lbfgs_code in (0, 1, 2), (name, tag, info.lbfgs_code)
    assert rel(info.fmin, float(d["lbfgs_%s_fmin" % tag])) < 2e-5, (name, tag, info.fmin)
    wref = d["lbfgs_%s_wopt" % tag]
    wtol = max(W_RTOL, 3.0 * float(d["lbfgs_%s_wspread" % tag]))
    assert np.abs(w - wref).max() <= wtol * wref.max(), (name, tag)


@pytest.mark.parametrize("name", FORCES_GOLDEN)
@pytest.mark.parametrize("tag,params", [("tight", LBFGS_TIGHT), ("mt", dict(LBFGS_DEFAULTS, linesearch=0))])
def test_forces_other_goldens_vs_reference(hip, name, tag, params):
    d = load_golden(name)
    code_ref = int(d["lbfgs_%s_code" % tag])
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        fopt, w, info = ctx.opt_lbfgs_forces(d["forces_init"], d["w0"], d["theta"], params)
    if code_ref not in (0, 1, 2):
        assert info.lbfgs_code < 0 or info.lbfgs_code in (0, 1)
        return
    assert info.lbfgs_code in (0, 1, 2), (name, tag, info.lbfgs_code)
    assert rel(info.fmin, float(d["lbfgs_%s_fmin" % tag])) < (FMIN_RTOL if tag == "tight" else 5e-6), (name, tag)
    wref = d["lbfgs_%s_wopt" % tag]
    wtol = max(W_RTOL, 3.0 * float(d["lbfgs_%s_wspread" % tag]))
    assert np.abs(w - wref).max() <= wtol * wref.max(), (name, tag)


@pytest.mark.parametrize("linesearch", [0, 1, 3])
def test_logw_other_linesearches_vs_oracle(hip, linesearch):
    from oracle import oracle_binding as O
    d = load_golden("synth_logw_M37xN500.npz")
    params = dict(LBFGS_TIGHT, linesearch=linesearch)
    g_o, fmin_o, code_o, it_o, ev_o = O.opt_lbfgs_logw(d["GInit"], d["G"], d["yTilde"], d["YTilde"], d["theta"], params)
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        gopt, w, info = ctx.opt_lbfgs_logw(d["GInit"], d["G"], d["theta"], params)
    if code_o in (0, 1, 2):
        assert info.lbfgs_code in (0, 1, 2)
        assert rel(info.fmin, fmin_o) < FMIN_RTOL
        w_o, _ = O.logw_weights(g_o)
        assert np.abs(w - w_o).max() <= W_RTOL * w_o.max()
        # same algorithm => iteration counts agree up to last-bit trajectory drift
        assert abs(info.iterations - it_o) <= max(3, it_o // 5)
    else:
        assert info.lbfgs_code < 0


def test_lbfgs_error_codes(hip):
    e = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "error_codes.npz"))
    yTilde, YTilde = e["yTilde"], e["YTilde"]
    G = np.zeros(yTilde.shape[1])
    with hip.Context(yTilde, YTilde) as ctx:
        _, _, info = ctx.opt_lbfgs_logw(G, G, 1.0, dict(LBFGS_DEFAULTS, delta=-1.0))
        assert info.lbfgs_code == int(e["code_delta_neg"]) == -1015
        assert info.evaluations == 0
        _, _, info = ctx.opt_lbfgs_logw(G, G, 1.0, dict(LBFGS_DEFAULTS, max_iterations=3))
        assert info.lbfgs_code == int(e["code_maxiter"]) == -997
        assert info.iterations == 3
        assert rel(info.fmin, float(e["f_maxiter"])) < 1e-9
    L = hip._lib.lib()
    assert L.bioen_hip_lbfgs_strerror(-1015).decode() == str(e["msg_delta_neg"])
    assert L.bioen_hip_lbfgs_strerror(-997).decode() == str(e["msg_maxiter"])
    for code in (0, 1, 2):
        assert L.bioen_hip_lbfgs_strerror(code).decode() == str(e["msg_%d" % code])


def test_bitwise_reproducible(hip):
    """Fixed-order reductions: two runs of the same problem agree to the last bit
    (the reference needs fast_openmp=0 + serial sums for this, test_logw_reproducibility.py)."""
    d = load_golden("synth_logw_M64xN2000.npz")
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        a = ctx.opt_lbfgs_logw(d["GInit"], d["G"], d["theta"], LBFGS_DEFAULTS)
        b = ctx.opt_lbfgs_logw(d["GInit"], d["G"], d["theta"], LBFGS_DEFAULTS)
    assert a[2].fmin == b[2].fmin and a[2].iterations == b[2].iterations
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ---------------------------------------------------------------------------------------
# edge shapes: ragged sizes around the 128-column / 32-row padding, tiny problems
# ---------------------------------------------------------------------------------------
# ... and around the row limits of the forces strip passes (k_forces_xy / k_forces_bt: up to 512 rows
# with 256-thread blocks, up to 1024 with 512-thread blocks, a thread's second row is t + THREADS;
# beyond 1024 rows the four streaming passes take over)
@pytest.mark.parametrize("M,N", [(1, 1), (1, 2), (3, 127), (31, 128), (32, 129), (33, 255), (65, 257), (7, 1000),
                                 (255, 300), (257, 2049), (500, 1500), (512, 4100), (513, 700), (1000, 900),
                                 (1024, 1100), (1025, 300),
                                 # beyond 1024 rows the log-weights passes run over row panels of <= 1024 rows (a 1-row
                                 # panel, two full ones, a ragged third)
                                 (1040, 129), (2048, 200), (2100, 333)])
def test_ragged_shapes_vs_oracle(hip, M, N):
    from oracle import oracle_binding as O
    rng = np.random.default_rng(1000 * M + N)
    yTilde = rng.normal(5.0, 2.0, (M, N))
    YTilde = rng.normal(5.0, 0.5, M)
    w0 = rng.uniform(0.2, 1.0, N)
    w0 /= w0.sum()
    G = np.log(w0)
    g = G + 0.3 * rng.standard_normal(N)
    theta = 2.5
    f_o, grad_o, w_o = O.logw_fdf(g, G, yTilde, YTilde, theta)
    forces = 0.01 * rng.standard_normal(M)
    ff_o, fg_o, fw_o = O.forces_fdf(forces, w0, yTilde, YTilde, theta)
    with hip.Context(yTilde, YTilde) as ctx:
        f, grad = ctx.logw_fdf(g, G, theta)
        w, _ = ctx.logw_weights(g)
        ff, fg = ctx.forces_fdf(forces, w0, theta)
        fw = ctx.forces_weights(forces, w0)
        back = ctx.read_ytilde()
    assert np.array_equal(back, yTilde)
    assert rel(f, f_o) < F_RTOL and maxrel(w, w_o) < 1e-13
    # absolute floor: the strip kernels form sum_i r_i (Y_ij - c_i) + sum_i r_i (c_i - ybar_i) -- two sums that cancel
    # to rounding where the reference's single centred sum is exactly zero (a single structure: N = 1)
    assert np.abs(grad - grad_o).max() <= G_RTOL * max(np.abs(grad_o).max(), 1e-12) + 1e-14 * np.abs(yTilde).max() * (abs(f_o) + 1)
    assert rel(ff, ff_o) < F_RTOL and maxrel(fw, fw_o) < 1e-12
    # the forces gradient is a difference of O(|yTilde| |t|) terms: absolute scale, not its own size
    assert np.abs(fg - fg_o).max() <= 1e-9 * max(np.abs(fg_o).max(), 1e-12) + 1e-13 * np.abs(yTilde).max() * (abs(ff_o) + 1)


def test_extreme_log_weights_do_not_overflow(hip):
    """exp(g) overflows for g ~ 800; the max-shifted device softmax must not (the
    reference's un-shifted _get_weights would return inf/nan here)."""
    rng = np.random.default_rng(5)
    M, N = 8, 300
    yTilde = rng.normal(5.0, 2.0, (M, N))
    YTilde = rng.normal(5.0, 0.5, M)
    g = rng.standard_normal(N) + 800.0
    G = np.zeros(N)
    with hip.Context(yTilde, YTilde) as ctx:
        f, grad = ctx.logw_fdf(g, G, 1.0)
        f2, grad2 = ctx.logw_fdf(g - 800.0, G, 1.0)
    assert np.isfinite(f) and np.all(np.isfinite(grad))
    # L is NOT invariant under a shift of g (the prior sees g - G), but the weights are:
    with hip.Context(yTilde, YTilde) as ctx:
        w1, _ = ctx.logw_weights(g)
        w2, _ = ctx.logw_weights(g - 800.0)
    assert maxrel(w1, w2) < 1e-12


# ---------------------------------------------------------------------------------------
# lock-step batch: K thetas share every matrix pass and must reproduce K single runs bitwise
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,max_batch", [("synth_logw_M37xN500.npz", 8), ("synth_logw_M37xN500.npz", 3),
                                            ("synth_logw_M129xN257.npz", 5), ("ref_data_potra_part_2_logw_M808xN10.npz", 8)])
def test_batched_theta_series_equals_single_runs_bitwise(hip, name, max_batch):
    d = load_golden(name)
    thetas = [200.0, 0.3, 50.0, 7.0, 1.0, 20.0, 2.0, 0.7, 100.0, 3.0, 0.5]      # 11 thetas: slots get refilled
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        single = [ctx.opt_lbfgs_logw(d["GInit"], d["G"], th, LBFGS_DEFAULTS) for th in thetas]
        res, w, infos = ctx.opt_lbfgs_logw_batch(thetas, d["GInit"], d["G"], LBFGS_DEFAULTS, max_batch=max_batch)
        # per-theta start vectors
        starts = np.stack([d["GInit"].ravel() + 0.01 * i for i in range(len(thetas))])
        res2, w2, infos2 = ctx.opt_lbfgs_logw_batch(thetas, starts, d["G"], LBFGS_DEFAULTS, max_batch=max_batch)
        single2 = [ctx.opt_lbfgs_logw(starts[i], d["G"], th, LBFGS_DEFAULTS) for i, th in enumerate(thetas)]
    for i, (g1, w1, i1) in enumerate(single):
        assert infos[i].lbfgs_code == i1.lbfgs_code and infos[i].iterations == i1.iterations
        assert infos[i].evaluations == i1.evaluations
        assert infos[i].fmin == i1.fmin and infos[i].chi2 == i1.chi2 and infos[i].kl == i1.kl
        assert np.array_equal(res[i], g1) and np.array_equal(w[i], w1)
    for i, (g1, w1, i1) in enumerate(single2):
        assert infos2[i].fmin == i1.fmin and infos2[i].iterations == i1.iterations
        assert np.array_equal(res2[i], g1) and np.array_equal(w2[i], w1)


def test_batched_series_with_failing_and_trivial_members(hip):
    """A batch whose members end differently: converged, stopped, max_iterations hit."""
    d = load_golden("synth_logw_M64xN2000.npz")
    thetas = [1e6, 10.0, 0.05]
    params = dict(LBFGS_DEFAULTS, max_iterations=60)
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        res, w, infos = ctx.opt_lbfgs_logw_batch(thetas, d["G"], d["G"], params, max_batch=8)
        single = [ctx.opt_lbfgs_logw(d["G"], d["G"], th, params) for th in thetas]
        bad = ctx.opt_lbfgs_logw_batch(thetas, d["G"], d["G"], dict(LBFGS_DEFAULTS, delta=-1.0))
    assert [i.lbfgs_code for i in infos] == [s[2].lbfgs_code for s in single]
    assert infos[2].lbfgs_code == -997 and infos[2].iterations == 60
    for i in range(3):
        assert infos[i].fmin == single[i][2].fmin and np.array_equal(res[i], single[i][0])
        assert abs(w[i].sum() - 1.0) < 1e-12
    assert all(i.lbfgs_code == -1015 and i.evaluations == 0 for i in bad[2])
    assert np.array_equal(bad[0][1], d["G"].ravel())


# ---------------------------------------------------------------------------------------
# direction from inner products ("Gram form"): same recursion, different rounding
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", LOGW_GOLDEN)
def test_gram_direction_matches_two_loop(hip, name):
    d = load_golden(name)
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        ctx.set_direction_mode("twoloop")
        ref_def = ctx.opt_lbfgs_logw(d["GInit"], d["G"], d["theta"], LBFGS_DEFAULTS)
        ref_tight = ctx.opt_lbfgs_logw(d["GInit"], d["G"], d["theta"], LBFGS_TIGHT)
        tl_batch = ctx.opt_lbfgs_logw_batch([30.0, 3.0, 300.0], d["GInit"], d["G"], LBFGS_DEFAULTS, max_batch=8)
        tl_single = [ctx.opt_lbfgs_logw(d["GInit"], d["G"], th, LBFGS_DEFAULTS) for th in (30.0, 3.0, 300.0)]
        ctx.set_direction_mode("gram")
        g_def = ctx.opt_lbfgs_logw(d["GInit"], d["G"], d["theta"], LBFGS_DEFAULTS)
        g_tight = ctx.opt_lbfgs_logw(d["GInit"], d["G"], d["theta"], LBFGS_TIGHT)
        again = ctx.opt_lbfgs_logw(d["GInit"], d["G"], d["theta"], LBFGS_DEFAULTS)
        thetas = [30.0, 3.0, 300.0]
        batch = ctx.opt_lbfgs_logw_batch(thetas, d["GInit"], d["G"], LBFGS_DEFAULTS, max_batch=8)
        singles = [ctx.opt_lbfgs_logw(d["GInit"], d["G"], th, LBFGS_DEFAULTS) for th in thetas]
        ctx.set_direction_mode("auto")
    for i, sgl in enumerate(tl_single):   # batched == single in the two-loop mode
        assert tl_batch[2][i].fmin == sgl[2].fmin and np.array_equal(tl_batch[0][i], sgl[0])
    # deterministic, and batched == single also in this mode
    assert again[2].fmin == g_def[2].fmin and np.array_equal(again[0], g_def[0])
    for i, sgl in enumerate(singles):
        assert batch[2][i].fmin == sgl[2].fmin and np.array_equal(batch[0][i], sgl[0])
    for (r, g, tol) in ((ref_def, g_def, 5e-5), (ref_tight, g_tight, FMIN_RTOL)):   # 5e-5: plateau stop on the flattest case
        if r[2].lbfgs_code in (0, 1, 2):
            assert g[2].lbfgs_code in (0, 1, 2)
            assert rel(g[2].fmin, r[2].fmin) < tol, (name, g[2].fmin, r[2].fmin)
            assert abs(g[2].iterations - r[2].iterations) <= max(5, r[2].iterations // 3)
    if ref_tight[2].lbfgs_code in (0, 1, 2):
        wtol = max(W_RTOL, 3.0 * float(d["lbfgs_tight_wspread"]))
        assert np.abs(g_tight[1] - ref_tight[1]).max() <= wtol * ref_tight[1].max()
    # and against the reference's own run
    if int(d["lbfgs_tight_code"]) in (0, 1, 2) and g_tight[2].lbfgs_code in (0, 1, 2):
        assert rel(g_tight[2].fmin, float(d["lbfgs_tight_fmin"])) < FMIN_RTOL


# ---------------------------------------------------------------------------------------
# forces method: theta series as one lock-step batch (BASELINE config 5 in miniature)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,max_batch", [("synth_forces_M30xN1000.npz", 8), ("synth_forces_M96xN3000.npz", 3),
                                            ("ref_data_forces_M64xN64.npz", 5)])
def test_batched_forces_series_equals_single_runs_bitwise(hip, name, max_batch):
    d = load_golden(name)
    thetas = [200.0, 0.5, 50.0, 7.0, 1.0, 20.0, 2.0, 100.0, 3.0, 10.0]
    with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
        single = [ctx.opt_lbfgs_forces(d["forces_init"], d["w0"], th, LBFGS_DEFAULTS) for th in thetas]
        res, w, infos = ctx.opt_lbfgs_forces_batch(thetas, d["forces_init"], d["w0"], LBFGS_DEFAULTS,
                                                   max_batch=max_batch)
        starts = np.stack([d["forces_init"].ravel() + 1e-4 * i for i in range(len(thetas))])
        res2, w2, infos2 = ctx.opt_lbfgs_forces_batch(thetas, starts, d["w0"], LBFGS_DEFAULTS, max_batch=max_batch)
        single2 = [ctx.opt_lbfgs_forces(starts[i], d["w0"], th, LBFGS_DEFAULTS) for i, th in enumerate(thetas)]
        bad = ctx.opt_lbfgs_forces_batch(thetas[:3], d["forces_init"], d["w0"], dict(LBFGS_DEFAULTS, delta=-1.0))
    for i, (f1, w1, i1) in enumerate(single):
        assert infos[i].lbfgs_code == i1.lbfgs_code and infos[i].iterations == i1.iterations
        assert infos[i].evaluations == i1.evaluations
        assert infos[i].fmin == i1.fmin and infos[i].chi2 == i1.chi2 and infos[i].kl == i1.kl
        assert np.array_equal(res[i], f1) and np.array_equal(w[i], w1)
        assert rel(infos[i].fmin, thetas[i] * infos[i].kl + infos[i].chi2) < 1e-12
    for i, (f1, w1, i1) in enumerate(single2):
        assert infos2[i].fmin == i1.fmin and np.array_equal(res2[i], f1) and np.array_equal(w2[i], w1)
    assert all(i.lbfgs_code == -1015 and i.evaluations == 0 for i in bad[2])
    # theta = 10 is the golden file's own theta for two of the three cases
    if abs(d["theta"] - 10.0) < 1e-12:
        k = thetas.index(10.0)
        assert rel(infos[k].fmin, float(d["lbfgs_def_fmin"])) < 5e-6


def test_large_forces_batch_vs_oracle(hip):
    """N = 2e5 x M = 128 generated in HBM: one batched forces evaluation and a short series vs the oracle."""
    from oracle import oracle_binding as O
    M, N = 128, 200000
    rng = np.random.default_rng(12345)
    YTrue = rng.uniform(1, 10, M)
    sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
    YTilde = rng.normal(YTrue, sig_exp) / sig_exp
    w0 = np.full(N, 1.0 / N)
    with hip.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=7) as ctx:
        y = ctx.read_ytilde()
        f0 = 1e-4 * rng.standard_normal(M)
        f, grad = ctx.forces_fdf(f0, w0, 10.0)
        f_o, grad_o, w_o = O.forces_fdf(f0, w0, y, YTilde, 10.0)
        assert rel(f, f_o) < 1e-12 and np.abs(grad - grad_o).max() <= 1e-9 * np.abs(grad_o).max()
        thetas = [100.0, 10.0, 1.0]
        res, w, infos = ctx.opt_lbfgs_forces_batch(thetas, np.zeros(M), w0, LBFGS_DEFAULTS)
        for th, info, wi in zip(thetas, infos, w):
            fo, fmin_o, code_o, it_o, ev_o = O.opt_lbfgs_forces(np.zeros(M), w0, y, YTilde, th)
            # (-998: either side may end on an exhausted line search at the rounding floor)
            assert info.lbfgs_code in (0, 1, -998) and code_o in (0, 1, -998)
            assert rel(info.fmin, fmin_o) < 2e-5
            assert abs(wi.sum() - 1.0) < 1e-12


def test_forces_lbfgs_beyond_the_strip_limit_vs_oracle(hip):
    """M > 1024 takes the four streaming passes (no LDS strips): a whole L-BFGS series through that
    path, batched == single bit for bit and equal to the oracle within the stopping plateau."""
    from oracle import oracle_binding as O
    rng = np.random.default_rng(77)
    M, N = 1030, 400
    yTilde = rng.normal(5.0, 2.0, (M, N))
    YTilde = rng.normal(5.0, 0.5, M)
    w0 = rng.dirichlet(np.ones(N) * 3.0)
    thetas = [1e4, 1e3]
    with hip.Context(yTilde, YTilde) as ctx:
        res, w, infos = ctx.opt_lbfgs_forces_batch(thetas, np.zeros(M), w0, LBFGS_DEFAULTS)
        for k, th in enumerate(thetas):
            fs, ws, info = ctx.opt_lbfgs_forces(np.zeros(M), w0, th, LBFGS_DEFAULTS)
            assert infos[k].fmin == info.fmin and np.array_equal(res[k], fs) and np.array_equal(w[k], ws)
            fo, fmin_o, code_o, it_o, ev_o = O.opt_lbfgs_forces(np.zeros(M), w0, yTilde, YTilde, th)
            assert info.lbfgs_code in (0, 1, -998) and code_o in (0, 1, -998)
            assert rel(info.fmin, fmin_o) < 2e-5
            assert abs(ws.sum() - 1.0) < 1e-12


# ---------------------------------------------------------------------------------------
# speculative line-search trials in idle batch slots: same bits, fewer rounds
# ---------------------------------------------------------------------------------------
_SPEC_SNIPPET = r'''
import sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np
import bioen_amd
from conftest import load_golden, LBFGS_DEFAULTS
d = load_golden(%(name)r)
thetas = [200.0, 0.3, 50.0, 7.0, 1.0, 20.0, 2.0, 0.7, 100.0, 3.0, 0.5]
out = {}
with bioen_amd.Context(d["yTilde"], d["YTilde"]) as ctx:
    for ls in (2, 3, 1, 0):
        params = dict(LBFGS_DEFAULTS, linesearch=ls)
        res, w, infos = ctx.opt_lbfgs_logw_batch(thetas, d["GInit"], d["G"], params, max_batch=%(mb)d)
        single = ctx.opt_lbfgs_logw(d["GInit"], d["G"], 0.3, params)
        out[str(ls)] = dict(res=res.tobytes().hex(), w=w.tobytes().hex(), fmin=[i.fmin for i in infos],
                            it=[i.iterations for i in infos], ev=[i.evaluations for i in infos],
                            code=[i.lbfgs_code for i in infos], single=single[0].tobytes().hex(), single_fmin=single[2].fmin)
    out["stats"] = ctx.speculation_stats()
print(json.dumps(out))
'''


@pytest.mark.parametrize("name,mb", [("synth_logw_M37xN500.npz", 3), ("synth_logw_M129xN257.npz", 8)])
def test_speculative_line_search_changes_no_bit(hip, name, mb):
    """The batch engine evaluates the steps a backtracking search may ask for next (stp/2, 2.1 stp) in idle batch
    slots, alongside the trial; a rejected trial then finds its successor already evaluated.  With the mechanism
    off (BIOEN_HIP_SPECULATE=0) every result -- optimum, weights, fmin, iteration and evaluation counts, status --
    must be the same to the last bit, for all three backtracking variants; More-Thuente never speculates."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = _SPEC_SNIPPET % dict(root=ROOT, tests=os.path.join(ROOT, "tests"), name=name, mb=mb)
    variants = {"default": {}, "nospec": {"BIOEN_HIP_SPECULATE": "0"}, "nodelivery": {"BIOEN_HIP_DELIVERY": "0"},
                "device": {"BIOEN_HIP_DEVICE_LS": "1"}, "device-shadows": {"BIOEN_HIP_DEVICE_LS": "1", "BIOEN_HIP_SHADOWS": "2", "BIOEN_HIP_SHADOW_RATE": "0.02"},
                # the sharded contexts' form of it on one GPU: shadows sweep their own (s, y) pair (no late Gram pass), from
                # the first search on, two slots kept back from a series that fills the batch
                "device-sgram": {"BIOEN_HIP_DEVICE_LS": "1", "BIOEN_HIP_SHADOWS": "2", "BIOEN_HIP_SHADOW_RATE": "0",
                                 "BIOEN_HIP_SHADOW_MINEV": "0", "BIOEN_HIP_DEV_RESERVE": "2", "BIOEN_HIP_SHADOW_GRAM": "1"},
                # the helper thread's timing decides which slots are free for shadows: without speculation AND
                # without it the schedule is the plainest one
                "plain": {"BIOEN_HIP_SPECULATE": "0", "BIOEN_HIP_DELIVERY": "0"},
                # line-search decisions taken by the host from the round's scalars (the r02 engine) instead of on the device
                "hostls": {"BIOEN_HIP_DEVICE_LS": "0"},
                "hostls-nospec": {"BIOEN_HIP_DEVICE_LS": "0", "BIOEN_HIP_SPECULATE": "0"},
                # device-resident decisions without the round queued ahead of the host
                "noqueue": {"BIOEN_HIP_QUEUE": "0", "BIOEN_HIP_SHADOW_RATE": "0", "BIOEN_HIP_SHADOWS": "2"},
                # ... with every idle slot shadowing from the first round on (the default policy waits for a problem to
                # show a rejection rate first), two rounds queued ahead
                "eager": {"BIOEN_HIP_SHADOW_RATE": "0", "BIOEN_HIP_SHADOWS": "8", "BIOEN_HIP_QUEUE": "2"}}
    runs = {}
    for tag, flags in variants.items():
        env = dict(os.environ, **flags)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (tag, p.stderr[-2000:])
        runs[tag] = json.loads(p.stdout.strip().splitlines()[-1])
    on = runs["default"]
    for tag in ("nospec", "plain", "hostls-nospec", "default", "device"):
        assert runs[tag]["stats"] == [0, 0], tag
    for tag in ("hostls", "noqueue", "eager", "device-sgram"):
        assert runs[tag]["stats"][0] > 0 and runs[tag]["stats"][1] > 0, (tag, runs[tag]["stats"])   # issued and adopted
    for tag, other in runs.items():
        for ls in ("2", "3", "1", "0"):
            assert on[ls] == other[ls], (tag, ls)


@pytest.mark.parametrize("name", ["synth_forces_M30xN1000.npz", "synth_forces_M96xN3000.npz"])
def test_forces_speculative_line_search_changes_no_bit(hip, name, monkeypatch):
    """The forces engine evaluates the steps a backtracking search may ask for next (the chain of halvings first) in batch
    slots without a problem; an adopted shadow's numbers ARE that trial's.  Off (BIOEN_HIP_SPECULATE=0) and on: optimum,
    weights, fmin, iteration and evaluation counts, status identical to the last bit -- single runs (seven free slots), a
    warm-started chain (the ala5 protocol, where searches are long) and lock-step batches with fewer problems than slots."""
    d = load_golden(name)
    w0 = d["w0"] / d["w0"].sum()
    M = d["yTilde"].shape[0]
    thetas = [1e4, 300.0, 20.0, 2.0, 0.4]
    runs = {}
    for tag in ("on", "off"):
        monkeypatch.setenv("BIOEN_HIP_SPECULATE", "1" if tag == "on" else "0")
        out = []
        with hip.Context(d["yTilde"], d["YTilde"]) as ctx:
            for ls in (2, 3, 1, 0):
                params = dict(LBFGS_DEFAULTS, linesearch=ls)
                f = np.zeros(M)
                for th in thetas:                                      # warm-started chain of single runs
                    f, w, info = ctx.opt_lbfgs_forces(f, w0, th, params)
                    out.append((f.tobytes(), w.tobytes(), info.fmin, info.iterations, info.evaluations, info.lbfgs_code))
                res, w, infos = ctx.opt_lbfgs_forces_batch(thetas, np.zeros(M), w0, params, max_batch=8)   # 5 problems, 8 slots
                out.append((res.tobytes(), w.tobytes(), [(i.fmin, i.iterations, i.evaluations, i.lbfgs_code) for i in infos]))
            stats = ctx.speculation_stats()
        runs[tag] = (out, stats)
    assert tuple(runs["off"][1]) == (0, 0)
    assert runs["on"][1][0] > 0 and runs["on"][1][1] > 0, runs["on"][1]          # issued, and some adopted
    assert runs["on"][0] == runs["off"][0]


@pytest.mark.parametrize("M,N", [(1300, 700), (2050, 400)])
def test_logw_lbfgs_over_row_panels_vs_oracle(hip, M, N, monkeypatch):
    """Matrices taller than 1024 rows: the log-weights matrix passes run the strip kernels once per panel of <= 1024 rows
    (forward: the panel's rows of the partial sums; adjoint: the column sums continue across panels).  A converged L-BFGS
    series through that path against the oracle (1e-6 / 1e-5), batched == single and device == host engine bit for bit,
    and objective/gradient against the r01 streaming kernels (BIOEN_HIP_PANELS=0) to rounding."""
    from oracle import oracle_binding as O
    rng = np.random.default_rng(M + N)
    yTilde = rng.normal(5.0, 2.0, (M, N))
    YTilde = yTilde.dot(rng.dirichlet(np.ones(N))) + 0.05 * rng.standard_normal(M)
    w0 = rng.dirichlet(np.ones(N) * 3.0)
    G = np.log(w0)
    g0 = G + 0.2 * rng.standard_normal(N)
    thetas = [30.0, 3.0]
    forces = 0.01 * rng.standard_normal(M)
    with hip.Context(yTilde, YTilde) as ctx:
        f_p, grad_p = ctx.logw_fdf(g0, G, 3.0)
        assert ctx.footprint()[0] == {"strips", "strips_colsum"}            # the panels have taken the row-major matrix's place
        ff_p, fg_p = ctx.forces_fdf(forces, w0, 3.0)                    # the forces method's four passes over the same panels
        yraw_p, _ = ctx.last_average()
        res, w, infos = ctx.opt_lbfgs_logw_batch(thetas, g0, G, LBFGS_CONV)
        monkeypatch.setenv("BIOEN_HIP_DEVICE_LS", "0")
        res_h, w_h, infos_h = ctx.opt_lbfgs_logw_batch(thetas, g0, G, LBFGS_CONV)
        monkeypatch.delenv("BIOEN_HIP_DEVICE_LS")
        for k, th in enumerate(thetas):
            gs, ws, info = ctx.opt_lbfgs_logw(g0, G, th, LBFGS_CONV)
            assert infos[k].fmin == info.fmin == infos_h[k].fmin
            assert np.array_equal(res[k], gs) and np.array_equal(w[k], ws) and np.array_equal(res[k], res_h[k])
            g_o, fmin_o, code_o, it_o, ev_o = O.opt_lbfgs_logw(g0, G, yTilde, YTilde, th, LBFGS_CONV)
            assert rel(info.fmin, fmin_o) < 1e-6
            w_o = O.logw_weights(g_o)[0]
            assert np.abs(ws - w_o).max() <= 1e-5 * w_o.max()
        assert np.array_equal(ctx.read_ytilde(), yTilde)
    monkeypatch.setenv("BIOEN_HIP_PANELS", "0")
    with hip.Context(yTilde, YTilde) as ctx:
        f_s, grad_s = ctx.logw_fdf(g0, G, 3.0)
        assert ctx.footprint()[0] == {"rowmajor"}
        ff_s, fg_s = ctx.forces_fdf(forces, w0, 3.0)
        yraw_s, _ = ctx.last_average()
    ff_o, fg_o, fw_o = O.forces_fdf(forces, w0, yTilde, YTilde, 3.0)
    assert rel(ff_p, ff_s) < 1e-13 and rel(ff_p, ff_o) < F_RTOL
    assert np.abs(fg_p - fg_s).max() <= 1e-10 * np.abs(fg_s).max() + 1e-13 * np.abs(yTilde).max() * (abs(ff_o) + 1)
    assert np.abs(fg_p - fg_o).max() <= 1e-9 * np.abs(fg_o).max() + 1e-13 * np.abs(yTilde).max() * (abs(ff_o) + 1)
    assert np.abs(yraw_p - yraw_s).max() <= 1e-12 * np.abs(yraw_s).max() and np.abs(yraw_p - yTilde.dot(fw_o)).max() < 1e-11
    f_o, grad_o, _ = O.logw_fdf(g0, G, yTilde, YTilde, 3.0)
    assert rel(f_p, f_s) < 1e-13 and rel(f_p, f_o) < F_RTOL
    assert np.abs(grad_p - grad_s).max() <= 1e-11 * np.abs(grad_s).max()
    assert np.abs(grad_p - grad_o).max() <= G_RTOL * np.abs(grad_o).max() + 1e-14 * np.abs(yTilde).max() * (abs(f_o) + 1)


def test_forces_speculation_on_random_problems_changes_no_bit(hip, monkeypatch):
    """Twenty random small problems (shapes, priors, theta ladders, all three backtracking searches, odd batch widths):
    the forces engine with and without speculative chains -- every number identical."""
    rng = np.random.default_rng(2026)
    cases = []
    for _ in range(20):
        M, N = int(rng.integers(3, 90)), int(rng.integers(40, 700))
        Y = rng.normal(4.0, 2.0, (M, N))
        YT = Y.dot(rng.dirichlet(np.ones(N))) + 0.1 * rng.standard_normal(M)
        w0 = rng.dirichlet(np.ones(N) * 2.0)
        thetas = list(10.0 ** rng.uniform(-1, 4, int(rng.integers(1, 7))))
        cases.append((Y, YT, w0, thetas, int(rng.integers(1, 4)), int(rng.integers(1, 9)), 0.02 * rng.standard_normal(M)))
    outs = {}
    for tag in ("1", "0"):
        monkeypatch.setenv("BIOEN_HIP_SPECULATE", tag)
        res_all = []
        for Y, YT, w0, thetas, ls, mb, f0 in cases:
            params = dict(LBFGS_DEFAULTS, linesearch=ls, max_iterations=300)
            with hip.Context(Y, YT) as ctx:
                res, w, infos = ctx.opt_lbfgs_forces_batch(thetas, f0, w0, params, max_batch=mb)
                res_all.append((res.tobytes(), w.tobytes(),
                                [(i.fmin, i.chi2, i.kl, i.iterations, i.evaluations, i.lbfgs_code) for i in infos]))
        outs[tag] = res_all
    assert outs["1"] == outs["0"]


def test_randomised_differential_run_against_the_restatement(hip):
    """tools/fuzz_parity.py as a test: 80 random problems over every kernel geometry (15 ... 1100 rows, ragged column
    counts), random priors, starts, thetas, all four line searches, random stop settings -- device against the CPU
    restatement: f 1e-12, grad 1e-10 / 1e-9, weights 1e-13 / 1e-12, and runs capped at 3 ... 11 iterations with equal
    status, iterations AND evaluations, fmin 1e-8.  (r04: 800 seeds of it found the run refused for a non-descent
    direction -- status -994 -- returning the trial's value instead of the accepted point's.)"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    worst, bad = fuzz.run(0, 80, min_dim=15)
    assert not bad, bad


def test_randomised_batches_equal_their_single_runs(hip):
    """tools/fuzz_batch.py as a test: 50 random series -- 1 ... 14 thetas over 1 ... 8 batch slots (slots reused, problems
    finishing at different times, shadows coming and going), shared or own starts, all line searches, random caps, both
    methods, both log-weights engines: every problem returns the bits of its single run, and finite numbers."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_batch", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_batch.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad = fuzz.run(0, 50)
    assert not bad, bad


SWITCHES = [
    {"BIOEN_HIP_LIVE": "0"},                      # no coherent host page: copies + stream synchronisation, host-driven engine
    {"BIOEN_HIP_FORCES_LIVE": "0"},
    {"BIOEN_HIP_DELIVERY": "0"},                  # results delivered synchronously
    {"BIOEN_HIP_QUEUE": "0"}, {"BIOEN_HIP_QUEUE": "2"},
    {"BIOEN_HIP_PIN_RESULTS": "0"},
    {"BIOEN_HIP_NVEC_NT": "0"}, {"BIOEN_HIP_NVEC_NT": "1"},
    {"BIOEN_HIP_STRIP_DEPTH5": "1"}, {"BIOEN_HIP_STRIP_DEPTH5": "2"},
    {"BIOEN_HIP_HOST_SHADOWS": "0", "BIOEN_HIP_DEVICE_LS": "0"}, {"BIOEN_HIP_HOST_SHADOWS": "5", "BIOEN_HIP_DEVICE_LS": "0"},
    {"BIOEN_HIP_RESERVE": "0", "BIOEN_HIP_DEVICE_LS": "0"}, {"BIOEN_HIP_RESERVE": "3", "BIOEN_HIP_DEVICE_LS": "0"},
    {"BIOEN_HIP_KEEP_ROWMAJOR": "1"},
    {"BIOEN_HIP_DEVICE_LS": "0"}, {"BIOEN_HIP_DEVICE_LS": "1"},
    {"BIOEN_HIP_SHADOWS": "2", "BIOEN_HIP_SHADOW_RATE": "0", "BIOEN_HIP_SHADOW_MINEV": "0", "BIOEN_HIP_DEVICE_LS": "1"},
]


@pytest.mark.parametrize("switch", SWITCHES, ids=lambda s: ",".join("%s=%s" % kv for kv in sorted(s.items())))
def test_fallbacks_and_ab_switches_change_no_bit(hip, switch, monkeypatch):
    """Every fallback a context takes when a resource is missing (no coherent host page: BIOEN_HIP_LIVE=0 /
    BIOEN_HIP_FORCES_LIVE=0; synchronous deliveries) and every A/B switch that is documented as bit-preserving (queue depth,
    cache policy, strip-kernel forms, shadow policies, either engine) returns the default configuration's bits: an eight-theta
    log-weights series that fills the batch, a six-theta forces series (the K > 4 strip form), weights included."""
    rng = np.random.default_rng(77)
    M, N = 160, 9000
    YTrue = rng.uniform(1, 10, M)
    y = rng.normal(YTrue[:, None], 0.5 * YTrue[:, None], (M, N)) / (0.1 * YTrue[:, None])
    YT = rng.normal(YTrue, 0.1 * YTrue) / (0.1 * YTrue)
    G, w0, f0 = np.zeros(N), np.full(N, 1.0 / N), np.zeros(M)
    params = dict(LBFGS_DEFAULTS, max_iterations=45)
    th8 = list(np.logspace(2.5, -0.5, 8))
    th6 = [300.0, 100.0, 30.0, 10.0, 3.0, 1.0]

    def series():
        with hip.Context(y, YT) as ctx:
            a = ctx.opt_lbfgs_logw_batch(th8, G, G, params, max_batch=8)
            b = ctx.opt_lbfgs_forces_batch(th6, f0, w0, params, max_batch=6)
        return (a[0].tobytes(), a[1].tobytes(), [(i.fmin, i.iterations, i.evaluations, i.lbfgs_code, i.chi2, i.kl) for i in a[2]],
                b[0].tobytes(), b[1].tobytes(), [(i.fmin, i.iterations, i.evaluations, i.lbfgs_code, i.chi2, i.kl) for i in b[2]])

    if not hasattr(test_fallbacks_and_ab_switches_change_no_bit, "_default"):
        test_fallbacks_and_ab_switches_change_no_bit._default = series()
    for k, v in switch.items():
        monkeypatch.setenv(k, v)
    assert series() == test_fallbacks_and_ab_switches_change_no_bit._default
