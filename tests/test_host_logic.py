"""CPU tests of the host side of the drop-in: the bioen.optimize-compatible Python layer
(config parsing, shape contracts, exception texts, the explicit numpy `*_base` variants and
the scipy/py minimiser route) -- everything that does not need the device."""
import numpy as np
import pytest

from conftest import LOGW_GOLDEN, FORCES_GOLDEN, load_golden
from bioen_amd import optimize
from bioen_amd.optimize import common, forces, log_weights, minimize, util
from bioen_amd.optimize.ext import c_bioen
from oracle import oracle_binding as O


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


# ---- configuration surface (util.py:95-160, bioen_optimize.yaml) ----------------------
def test_parameters_defaults_match_reference_template():
    p = minimize.Parameters("lbfgs")
    assert p["minimizer"] == "lbfgs" and p["algorithm"] == "" and p["use_c_functions"] is True
    assert p["params"] == dict(linesearch=2, max_iterations=5000, delta=1e-6, epsilon=1e-6, ftol=1e-5,
                               gtol=0.9, wolfe=0.9, past=10, max_linesearch=100)
    assert p["cache_ytilde_transposed"] == "auto" and p["n_threads"] == -1
    s = minimize.Parameters("scipy")
    assert s["algorithm"] == "fmin_bfgs" and s["use_c_functions"] is True
    assert s["params"] == dict(gtol=0.001, pgtol=0.001, epsilon=0.1, max_iterations=5000)
    g = minimize.Parameters("gsl")
    assert g["algorithm"] == "gsl_multimin_fdfminimizer_vector_bfgs2"
    assert g["params"] == dict(step_size=0.01, tol=0.001, max_iterations=5000)


def test_parameter_mod_string():
    p = minimize.Parameters("lbfgs", "lbfgs:epsilon=1e-9,general:verbose=false,lbfgs:past=0,c_functions:n_threads=4")
    assert p["params"]["epsilon"] == 1e-9 and p["params"]["past"] == 0
    assert p["verbose"] is False and p["n_threads"] == 4
    assert util.ntype("12") == 12 and util.ntype("1.5") == 1.5 and util.ntype("Yes") is True
    assert util.ntype("n") is False and util.ntype("bfgs") == "bfgs"


def test_util_relative_differences():
    assert util.compute_relative_difference_for_values(1.1, 1.0) == pytest.approx(0.1)
    assert util.compute_relative_difference_for_values(0.25, 0.0) == 0.25
    d, idx = util.compute_relative_difference_for_arrays(np.array([1.0, 2.2, 0.5]), np.array([1.0, 2.0, 0.0]))
    assert d == pytest.approx(0.1) and idx == 1
    assert util.compute_relative_difference_for_arrays(np.ones(3), np.zeros(3)) == (0.0, 0)
    assert util.library_lbfgs() is True and util.library_gsl() is True


def test_fast_openmp_flag_roundtrip():
    minimize.set_fast_openmp_flag(1)
    assert minimize.get_fast_openmp_flag() == 1
    minimize.set_fast_openmp_flag(0)
    assert minimize.get_fast_openmp_flag() == 0


def test_caching_heuristic():
    assert common.set_caching_heuristics(1024, 1000000) is True        # 8.19e9 < 8 GiB
    assert common.set_caching_heuristics(1024, 1100000) is False


# ---- shape contracts and error texts ---------------------------------------------------------
def test_shape_checks_raise_valueerror():
    n, m = 7, 3
    ok = dict(GInit=np.zeros((n, 1)), G=np.zeros((n, 1)), y=np.zeros((m, n)), yTilde=np.zeros((m, n)),
              YTilde=np.zeros((1, m)))
    log_weights.check_params_logweights(**ok)
    for key, bad in (("GInit", np.zeros(n)), ("G", np.zeros((1, n))), ("y", np.zeros((n, m))),
                     ("YTilde", np.zeros((m, 1)))):
        args = dict(ok)
        args[key] = bad
        with pytest.raises(ValueError):
            log_weights.check_params_logweights(**args)
    okf = dict(forcesInit=np.zeros((m, 1)), w0=np.zeros((n, 1)), y=np.zeros((m, n)), yTilde=np.zeros((m, n)),
               YTilde=np.zeros((1, m)))
    forces.check_params_forces(**okf)
    for key, bad in (("forcesInit", np.zeros((1, m))), ("w0", np.zeros(n)), ("YTilde", np.zeros(m))):
        args = dict(okf)
        args[key] = bad
        with pytest.raises(ValueError):
            forces.check_params_forces(**args)


def test_gsl_and_unknown_minimizer_errors():
    assert c_bioen.get_gsl_method("bfgs2") == 2
    assert c_bioen.get_gsl_method("gsl_multimin_fdfminimizer_conjugate_pr") == 1
    with pytest.raises(RuntimeError) as e:          # test_error_opt_logw.py:65-83
        c_bioen.get_gsl_method("TEST_INVALID")
    assert "return code" in str(e.value) and "-1" in str(e.value)
    d = load_golden("ref_data_16x15.npz")
    cfg = minimize.Parameters("gsl")
    cfg["verbose"] = False
    cfg["algorithm"] = "TEST_INVALID"               # rejected before anything touches the device
    with pytest.raises(RuntimeError) as e:
        c_bioen.bioen_opt_bfgs_logw(d["GInit"].ravel(), d["G"], d["yTilde"], d["YTilde"], d["theta"], cfg)
    assert "GSL return code" in str(e.value)
    cfg = minimize.Parameters("scipy")
    cfg["verbose"] = False
    cfg["use_c_functions"] = False
    cfg["algorithm"] = "simplex"
    with pytest.raises(RuntimeError) as e:
        log_weights.find_optimum(d["GInit"], d["G"], d["y"], d["yTilde"], d["YTilde"].reshape(1, -1), d["theta"], cfg)
    assert "not recognized for scipy/py" in str(e.value)
    cfg["minimizer"] = "simplex"
    with pytest.raises(RuntimeError) as e:
        log_weights.find_optimum(d["GInit"], d["G"], d["y"], d["yTilde"], d["YTilde"].reshape(1, -1), d["theta"], cfg)
    assert "not recognized" in str(e.value)


# ---- the explicit numpy variants -------------------------------------------------------------
@pytest.mark.parametrize("name", LOGW_GOLDEN)
@pytest.mark.parametrize("as_matrix", [False, True])
def test_logw_base_functions_vs_reference_values(name, as_matrix):
    d = load_golden(name)
    conv = (lambda a: np.asmatrix(a)) if as_matrix else (lambda a: np.asarray(a))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        G, yT, YT = conv(d["G"]), conv(d["yTilde"]), conv(d["YTilde"].reshape(1, -1))
        g = conv(d["GInit"].copy())
        gP = np.asarray(d["g_pert"]).ravel()
        f = log_weights.bioen_log_posterior(gP, g, G, yT, YT, d["theta"], use_c=False)
        grad = log_weights.grad_bioen_log_posterior(gP, g, G, yT, YT, d["theta"], use_c=False)
    assert rel(f, float(d["f_pert"])) < 1e-12
    assert np.abs(grad - d["grad_pert"]).max() <= 1e-10 * np.abs(d["grad_pert"]).max()
    assert np.allclose(np.asarray(g).ravel(), gP)          # writes the point into g, like the reference
    w, s = log_weights.getWeights(d["GInit"])
    assert np.allclose(w.ravel(), d["w_init"], rtol=1e-13, atol=0) and rel(s, float(d["s_init"])) < 1e-13
    assert log_weights.getWOpt(d["G"], d["GInit"].ravel()).shape == (d["G"].size, 1)


@pytest.mark.parametrize("name", FORCES_GOLDEN)
def test_forces_base_functions_vs_reference_values(name):
    d = load_golden(name)
    YT = d["YTilde"].reshape(1, -1)
    fp = d["forces_pert"].reshape(-1, 1)
    f = forces.bioen_log_posterior(fp, d["w0"], None, d["yTilde"], YT, d["theta"], use_c=False)
    grad = forces.grad_bioen_log_posterior(fp, d["w0"], None, d["yTilde"], YT, d["theta"], use_c=False)
    assert rel(f, float(d["f_pert"])) < 1e-12
    assert np.abs(grad - d["grad_pert"]).max() <= 5e-8 * np.abs(d["grad_pert"]).max()
    w = forces.get_weights_from_forces(d["w0"], d["yTilde"], d["forces_pert"])
    assert w.shape == (d["w0"].size, 1)
    assert np.allclose(w.ravel(), d["w_pert"], rtol=1e-12, atol=0)
    S, chi = forces.bioen_chi2_s_forces(fp, d["w0"], d["yTilde"], YT)
    assert S >= -1e-15 and rel(d["theta"] * S + chi, float(d["f_pert"])) < 1e-12


def test_init_helpers():
    w0 = np.full((5, 1), 0.2)
    gPrime, g, G, GInit = log_weights.init_log_weights(w0)
    assert gPrime.shape == (4,) and g.shape == (5, 1) and np.all(G == 0)
    f0 = forces.init_forces(4, 0.5)
    assert f0.shape == (4, 1) and np.all(f0 == 0.5)
    np.random.seed(0)
    YObs, YT = forces.gen_synthetic_data(3, 10, np.array([1.0, 2.0, 3.0]), np.array([0.1, 0.2, 0.3]))
    assert YObs.shape == (3,) and np.allclose(YT, YObs / np.array([0.1, 0.2, 0.3]))
    y, yT = forces.gen_sythetic_ensemble(3, 10, np.array([1.0, 2.0, 3.0]), np.array([0.1, 0.2, 0.3]),
                                         np.array([[0.5], [1.0], [1.5]]))
    assert y.shape == (3, 10) and np.allclose(yT, y / np.array([[0.1], [0.2], [0.3]]))
    assert common.getAve(np.full((10, 1), 0.1), y).shape == (3,)
    assert rel(common.chiSqrTerm(np.full((10, 1), 0.1), yT, YT.reshape(1, -1)),
               0.5 * np.sum((yT.mean(axis=1) - YT) ** 2)) < 1e-12


# ---- the scipy / pure-python minimiser route runs end-to-end on the CPU ------------------------
@pytest.mark.parametrize("algorithm", ["lbfgs", "bfgs", "cg"])
def test_find_optimum_scipy_py_route(algorithm):
    """test_find_opt_analytical_grad_logw.py:134-147 for minimizer scipy_py: fmin within 1e-1 of *.ref."""
    d = load_golden("ref_data_potra_part_2_logw_M205xN10.npz")
    cfg = minimize.Parameters("scipy")
    cfg.update(verbose=False, use_c_functions=False, algorithm=algorithm, cache_ytilde_transposed="False")
    wopt, yopt, gopt, f0, fmin = log_weights.find_optimum(d["GInit"], d["G"], d["y"], d["yTilde"],
                                                          d["YTilde"].reshape(1, -1), d["theta"], cfg)
    assert wopt.shape == (10, 1) and yopt.shape == (205,) and gopt.shape == (10,)
    assert rel(f0, float(d["f_init"])) < 1e-12
    assert rel(fmin, float(d["ref_fmin_scipy_bfgs"])) < 1e-1
    assert rel(fmin, log_weights.bioen_log_posterior_base(gopt, d["GInit"].copy(), d["G"], d["yTilde"],
                                                          d["YTilde"].reshape(1, -1), d["theta"])) < 5e-14
    assert cfg["cache_ytilde_transposed"] == "False"      # string-truthy, untouched


def test_find_optimum_forces_scipy_py_route():
    d = load_golden("ref_data_forces_M64xN64.npz")
    cfg = minimize.Parameters("scipy")
    cfg.update(verbose=False, use_c_functions=False, algorithm="bfgs")
    out = forces.find_optimum(d["forces_init"], d["w0"], d["y"], d["yTilde"], d["YTilde"].reshape(1, -1),
                              d["theta"], cfg)
    wopt, yopt, fopt, f0, fmin, chi, S = out
    assert wopt.shape == (64, 1) and yopt.shape == (64,) and np.asarray(fopt).size == 64
    assert rel(f0, float(d["f_init"])) < 1e-12
    assert rel(fmin, float(d["ref_fmin_scipy_bfgs"])) < 1e-1
    assert rel(d["theta"] * S + chi, fmin) < 1e-10
    assert cfg["cache_ytilde_transposed"] is True          # "auto" resolved and written back


def test_python_gradient_fix_is_the_c_gradient():
    """A8: for non-uniform G and theta > 0 the numpy gradient must equal the C/oracle one."""
    d = load_golden("synth_logw_M37xN500.npz")
    grad = log_weights.grad_bioen_log_posterior_base(d["g_pert"], d["GInit"].copy(), d["G"], d["yTilde"],
                                                     d["YTilde"].reshape(1, -1), d["theta"])
    assert np.abs(grad - d["grad_pert"]).max() <= 1e-10 * np.abs(d["grad_pert"]).max()
    _, go, _ = O.logw_fdf(d["g_pert"], d["G"], d["yTilde"], d["YTilde"], d["theta"])
    assert np.abs(grad - go).max() <= 1e-10 * np.abs(go).max()


def test_context_cache_fingerprint_is_content_sensitive():
    a = np.arange(12.0).reshape(3, 4)
    b = a.copy()
    assert c_bioen._fingerprint(a) == c_bioen._fingerprint(b)
    b[1, 2] += 1.0
    assert c_bioen._fingerprint(a) != c_bioen._fingerprint(b)
    # EVERY element takes part up to the full-check size: a one-element finite-difference perturbation anywhere
    rng = np.random.default_rng(3)
    big = rng.standard_normal((257, 1031))
    fp = c_bioen._fingerprint(big)
    for (i, j) in ((0, 1), (128, 517), (256, 1029), (13, 14)):
        old = big[i, j]
        big[i, j] += 1e-6
        assert c_bioen._fingerprint(big) != fp, (i, j)
        big[i, j] = old
    assert c_bioen._fingerprint(big) == fp
    # ... however small (below the rounding of any sum over the buffer) and however symmetric (r04: until then the check
    # was {sum, sum of squares}: two structures swapped in place, or a 1e-13 nudge in 265 000 numbers, went unseen)
    big[100, 500] = np.nextafter(big[100, 500], np.inf)
    assert c_bioen._fingerprint(big) != fp
    big[100, 500] = np.nextafter(big[100, 500], -np.inf)
    assert c_bioen._fingerprint(big) == fp
    big[:, [3, 900]] = big[:, [900, 3]]
    assert c_bioen._fingerprint(big) != fp
    big[:, [3, 900]] = big[:, [900, 3]]
    assert c_bioen._fingerprint(big) == fp
    big[[5, 200], :] = big[[200, 5], :]
    assert c_bioen._fingerprint(big) != fp


def test_context_cache_needs_the_same_live_object_and_the_same_content(monkeypatch):
    """a hit needs the very same host object (no address reuse) AND an unchanged content check."""
    made = []

    class FakeContext(object):
        def __init__(self, yT, YT):
            self.closed = False
            made.append(self)

        def set_target(self, YT):
            self.target = YT.copy()

        def close(self):
            self.closed = True

    monkeypatch.setattr(c_bioen._lib, "Context", FakeContext)
    c_bioen.clear_cache()
    y = np.arange(20.0).reshape(4, 5)
    YT = np.ones(4)
    c1, cached = c_bioen._context_for(y, YT)
    c2, _ = c_bioen._context_for(y, YT)
    assert cached and c1 is c2 and len(made) == 1
    c3, _ = c_bioen._context_for(y, YT * 2)                   # new targets: same context, targets replaced
    assert c3 is c1 and np.array_equal(c1.target, YT * 2)
    y[2, 3] += 1e-7                                            # in-place edit of ONE element -> miss, old context freed
    c4, _ = c_bioen._context_for(y, YT)
    assert c4 is not c1 and c1.closed and len(made) == 2
    twin = y.copy()                                            # equal content, other object -> its own context
    c5, _ = c_bioen._context_for(twin, YT)
    assert c5 is not c4 and len(made) == 3
    m = np.matrix(y)                                           # np.matrix callers (bioen.analyze) are cached too
    c6, cached6 = c_bioen._context_for(m, YT)
    c7, _ = c_bioen._context_for(m, YT)
    assert cached6 and c6 is c7
    c8, cached8 = c_bioen._context_for(y.tolist(), YT)         # no weak reference possible: never cached
    assert not cached8
    c_bioen.clear_cache()
    assert all(c.closed for c in made if c is not c8)


def test_a_held_matrix_survives_evictions_and_clear_cache(monkeypatch):
    """`with optimize.resident(A):` looping over A, B and C.  The hold owns A's context while it lasts -- calls on
    other matrices inside the block (LRU evictions, BIOEN_HIP_CACHE = 2) and clear_cache() must not close it; after the
    block it is back in the cache."""
    made = []

    class FakeContext(object):
        def __init__(self, yT, YT):
            self._h = object()
            made.append(self)

        def set_target(self, YT):
            pass

        def close(self):
            self._h = None

    monkeypatch.setattr(c_bioen._lib, "Context", FakeContext)
    c_bioen.clear_cache()
    YT = np.ones(4)
    A, B, Cm, D = (np.arange(20.0).reshape(4, 5) + i for i in range(4))
    a0, _ = c_bioen._context_for(A, YT)                       # cached before the block: the hold takes it over
    with optimize.resident(A, YT):
        assert id(A) not in c_bioen._CACHE
        for other in (B, Cm, D):                               # three more matrices through a cache of two
            c_bioen._context_for(other, YT)
            a, cached = c_bioen._context_for(A, YT)
            assert a is a0 and cached and a._h is not None
        c_bioen.clear_cache()
        a, _ = c_bioen._context_for(A, YT)
        assert a is a0 and a._h is not None                    # still alive, still the same upload
        with optimize.resident(A, YT):                         # nested
            assert c_bioen._context_for(A, YT)[0] is a0
        assert c_bioen._HELD
    assert not c_bioen._HELD
    assert c_bioen._CACHE.get(id(A)) is a0 and a0._h is not None      # handed back to the cache
    assert len(made) == 1 + 3
    big = np.zeros((4, 5))
    monkeypatch.setattr(c_bioen, "_FULL_CHECK_BYTES", 8)       # a matrix the cache refuses: the hold closes it at the end
    with optimize.resident(big, YT):
        held = c_bioen._context_for(big, YT)[0]
        c_bioen.clear_cache()
        assert held._h is not None
    assert held._h is None
    c_bioen.clear_cache()
