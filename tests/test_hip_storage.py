"""GPU tests of the reduced-byte storage EXPERIMENT (Context.set_storage; SURVEY 7: "FP32 storage of yTilde would halve bytes
but perturbs the optimum -- keep FP64 as the graded path; treat FP32/BF16-split as an experiment", 8 f4).  The default
(FP64) path is untouched by it: switching back restores the pinned bits.  The formats are held to the SAME un-widened
gate as the FP64 path -- converged runs against the reference's binary at BASELINE configs[1] size, 1e-6 on the
negative log-posterior, 1e-5 max(w) on the weights -- and the outcome is what the experiment reports: the 6-byte split
format passes, plain fp32 is recorded (not asserted to pass)."""
import numpy as np
import pytest

from conftest import LBFGS_DEFAULTS, require_reference

pytestmark = pytest.mark.gpu

CONV = dict(LBFGS_DEFAULTS, epsilon=1e-9, delta=0.0, past=0, max_iterations=200000)
_AT_OPTIMUM = (0, -998, -1000, -1001)


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def targets(M, seed=12345):
    rng = np.random.default_rng(seed)
    YTrue = rng.uniform(1, 10, M)
    sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
    return YTrue, sig_sim, sig_exp, rng.normal(YTrue, sig_exp) / sig_exp


@pytest.mark.parametrize("M,N", [(64, 3000), (205, 5000), (600, 3000), (1024, 4000)])
def test_reduced_formats_perturb_the_evaluation_by_their_rounding_only_and_f64_comes_back_bitwise(M, N):
    import bioen_amd
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    rng = np.random.default_rng(M)
    G = np.log(rng.dirichlet(np.ones(N) * 2.0))
    g = G + 0.3 * rng.standard_normal(N)
    thetas = [30.0, 3.0, 300.0]
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        f0, grad0 = ctx.logw_fdf(g, G, 5.0)
        w0 = rng.dirichlet(np.ones(N) * 2.0)
        fvec = 1e-3 * rng.standard_normal(M)
        ff0, fg0 = ctx.forces_fdf(fvec, w0, 5.0)
        r0 = ctx.opt_lbfgs_logw_batch(thetas, g, G, LBFGS_DEFAULTS)
        block0 = ctx.read_ytilde()
        for fmt, tol in (("split", 2.0 ** -30), ("fp32", 2.0 ** -21)):
            ctx.set_storage(fmt)
            f1, grad1 = ctx.logw_fdf(g, G, 5.0)
            assert rel(f1, f0) <= tol, (fmt, rel(f1, f0))
            assert np.abs(grad1 - grad0).max() <= 64 * tol * np.abs(grad0).max(), fmt
            assert f1 != f0 or fmt == "split"                      # the experiment really streams another copy
            # batched == single, bit for bit, within a format
            ra = ctx.opt_lbfgs_logw_batch(thetas, g, G, LBFGS_DEFAULTS)
            for k, th in enumerate(thetas):
                gs, ws, info = ctx.opt_lbfgs_logw(g, G, th, LBFGS_DEFAULTS)
                assert np.array_equal(gs, ra[0][k]) and info.fmin == ra[2][k].fmin
            assert np.array_equal(ctx.read_ytilde(), block0)       # the FP64 matrix stays what the caller gave
            # the forces method's strip passes stream the same copy
            ff1, fg1 = ctx.forces_fdf(fvec, w0, 5.0)
            assert rel(ff1, ff0) <= tol and np.abs(fg1 - fg0).max() <= 64 * tol * np.abs(fg0).max(), fmt
            fb, gb = ctx.forces_fdf_batch(np.stack([fvec, 0.5 * fvec, 0.0 * fvec]), w0, np.array([5.0, 50.0, 0.5]))
            assert fb[0] == ff1 and np.array_equal(gb[0], fg1)     # batched == single within the format
        ctx.set_storage("f64")
        f2, grad2 = ctx.logw_fdf(g, G, 5.0)
        r2 = ctx.opt_lbfgs_logw_batch(thetas, g, G, LBFGS_DEFAULTS)
        assert f2 == f0 and np.array_equal(grad2, grad0)
        ff2, fg2 = ctx.forces_fdf(fvec, w0, 5.0)
        assert ff2 == ff0 and np.array_equal(fg2, fg0)
        assert np.array_equal(r2[0], r0[0]) and [i.fmin for i in r2[2]] == [i.fmin for i in r0[2]]


def test_split_format_passes_the_unwidened_gate_against_the_reference_binary():
    """BASELINE configs[1] size, converged settings, the reference's own C + liblbfgs run as the yardstick -- exactly
    test_configs1_converged_against_the_reference_binary, with the matrix streamed in 6 bytes per element."""
    import bioen_amd
    from oracle import cpus
    R = require_reference()
    M, N = 256, 100000
    thetas = [316.0, 100.0, 31.6]
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    R.set_fast_openmp_flag(0)
    R.omp_set_num_threads(cpus.usable_cpus())
    G = np.zeros(N)
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        yT = np.ascontiguousarray(ctx.read_ytilde())
        refs = [R.opt_lbfgs_logw(G, G, yT, YTilde, th, CONV) for th in thetas]
        record = {}
        for fmt in ("split", "fp32"):
            ctx.set_storage(fmt)
            res, w, infos = ctx.opt_lbfgs_logw_batch(thetas, G, G, CONV)
            worst_f, worst_w = 0.0, 0.0
            for k, (g_ref, fmin_ref, code_ref) in enumerate(refs):
                assert code_ref in _AT_OPTIMUM and infos[k].lbfgs_code in _AT_OPTIMUM
                w_ref = np.asarray(R.get_weights(g_ref)[0]).ravel()
                worst_f = max(worst_f, rel(infos[k].fmin, fmin_ref))
                worst_w = max(worst_w, np.abs(w[k] - w_ref).max() / w_ref.max())
            record[fmt] = (worst_f, worst_w)
            print("storage %s: fmin within %.2e, weights within %.2e max(w) of the reference binary" % (fmt, worst_f, worst_w))
        assert record["split"][0] < 1e-6 and record["split"][1] <= 1e-5, record
        assert record["fp32"][0] < 1e-4, record                     # recorded, not gated: see DESIGN 9a


@pytest.mark.parametrize("M,N", [(512, 50000), (1024, 20000)])
def test_split_format_forces_method_passes_the_unwidened_gate_against_the_reference_binary(M, N):
    """the forces method on the 6-byte copies (k_strip / k_strip2), converged, against the reference's _opt_lbfgs_forces"""
    import bioen_amd
    from oracle import cpus
    R = require_reference()
    thetas = [316.0, 100.0, 31.6]
    YTrue, sig_sim, sig_exp, YTilde = targets(M)
    w0 = np.full(N, 1.0 / N)
    R.set_fast_openmp_flag(0)
    R.omp_set_num_threads(cpus.usable_cpus())
    with bioen_amd.Context.synthetic(M, N, YTrue, sig_sim, sig_exp, YTilde, seed=12345) as ctx:
        yT = np.ascontiguousarray(ctx.read_ytilde())
        ctx.set_storage("split")
        res, w, infos = ctx.opt_lbfgs_forces_batch(thetas, np.zeros(M), w0, CONV)
        for k, theta in enumerate(thetas):
            f_ref, fmin_ref, code_ref = R.opt_lbfgs_forces(np.zeros(M), w0, yT, YTilde, theta, CONV)
            assert code_ref in _AT_OPTIMUM and infos[k].lbfgs_code in _AT_OPTIMUM
            assert rel(infos[k].fmin, fmin_ref) < 1e-6, (theta, infos[k].fmin, fmin_ref)
            w_ref = np.asarray(R.forces_weights(f_ref, w0, yT)).ravel()
            assert np.abs(w[k] - w_ref).max() <= 1e-5 * w_ref.max(), (theta, np.abs(w[k] - w_ref).max() / w_ref.max())
