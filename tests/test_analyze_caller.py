"""north_star: "analyze and the existing examples call it unchanged".

A caller shaped like the reference's ``bioen/analyze/procedure.py:15-83`` -- ``from bioen import optimize``,
``optimize.minimize.Parameters(minimizer, parameter_mod)``, the theta loop with an inner iterations loop,
POSITIONAL arguments, ``np.matrix`` inputs, log-weights restarted from the same ``log_wopt`` for every theta
(:46,66), forces warm-started from the previous theta (:73-77), a nuisance hook between iterations (:78-83),
``optimize.common.chiSqrTerm`` / ``getAve`` afterwards -- runs against this package through the alias
INTEGRATION.md section 5 describes:  sys.modules["bioen.optimize"] = bioen_amd.optimize.

* CPU (``-m "not gpu"``): the ctypes layer ``bioen_amd.optimize.ext.c_bioen`` is replaced by fakes that answer
  from the oracle, so everything ABOVE the C ABI (argument checks, shapes, np.matrix handling, return tuples,
  cfg write-back) runs here without a GPU.
* GPU (``-m gpu``): the same caller on the real library, against the oracle's serial loop.
"""
import sys
import types
import warnings

import numpy as np
import pytest

from conftest import load_golden

# A caller of the KIND bioen/analyze/procedure.py is: only the two find_optimum calls (positional arguments, in the
# reference's order) and the forces warm start keep the reference's form -- they are the contract under test; the
# scaffolding around them is written for this test.
CALLER = '''
import numpy as np
from bioen import optimize


def as_column(values):
    return np.matrix(np.asarray(values, dtype=float).ravel()).T


def solve(method, carried, theta, params, data):
    """one optimisation of the series -> (the 5-tuple, the weights as a column matrix)"""
    if method == 'log-weights':
        out_min = optimize.log_weights.find_optimum(carried['log_wopt'], carried['log_w0'], carried['sim_init'],
                                                    carried['sim'], data.exp, theta, params)
        return out_min, out_min[0]
    out_min = optimize.forces.find_optimum(carried['forces'], carried['winit'].copy(), carried['sim_init'],
                                           carried['sim'], data.exp, theta, params)
    carried['forces'] = np.matrix(out_min[2]).T          # the next theta starts from these forces
    return out_min, np.matrix(out_min[0])


def run_series(settings, data):
    params = optimize.minimize.Parameters(settings.opt_minimizer, settings.opt_parameter_mod)
    for key, value in (('cache_ytilde_transposed', True), ('use_c_functions', True),
                       ('algorithm', settings.opt_algorithm), ('verbose', settings.opt_verbose)):
        params[key] = value
    prior = as_column(data.w0)
    carried = dict(winit=prior.copy(), log_w0=optimize.log_weights.getGs(prior),
                   log_wopt=optimize.log_weights.getGs(prior.copy()),          # every theta restarts from here
                   forces=np.matrix(np.zeros(data.exp.shape)).T, sim=data.sim, sim_init=data.sim_init)
    refit = settings.iterations > 1 or len(settings.thetas) > 1
    records = []
    for theta in settings.thetas:
        for _ in range(settings.iterations):
            out_min, weights = solve(settings.opt_method, carried, theta, params, data)
            if refit:                                     # the nuisance hook sees strictly positive weights
                positive = weights.copy()
                positive[positive == 0.0] = 1e-150
                carried['sim'], carried['sim_init'] = data.update_sim(positive, carried['sim'], carried['sim_init'])
        records.append(dict(theta=theta, wopt=weights, out_min=out_min, params=dict(params),
                            chi2=optimize.common.chiSqrTerm(weights, carried['sim'], data.exp),
                            yave=optimize.common.getAve(weights, carried['sim'])))
    return records
'''


class Obs(object):
    def __init__(self, d):
        self.sim = np.matrix(d["yTilde"])            # bioen.analyze hands np.matrix objects around
        self.sim_init = self.sim                     # generic data: y == yTilde
        self.exp = np.matrix(np.asarray(d["YTilde"]).reshape(1, -1))
        w0 = np.asarray(d["w0"], dtype=np.float64).ravel()
        self.w0 = w0 / w0.sum()
        self.updates = 0

    def update_sim(self, w, sim, sim_init):          # the nuisance hook (observables.py:191-216); generic data: no-op
        assert w.shape == (self.sim.shape[1], 1) and abs(w.sum() - 1.0) < 1e-9
        self.updates += 1
        return sim, sim_init


def options(method, thetas, iterations=1):
    return types.SimpleNamespace(opt_minimizer="lbfgs", opt_parameter_mod="", opt_algorithm="lbfgs", opt_verbose=False,
                                 opt_method=method, thetas=list(thetas), iterations=iterations)


@pytest.fixture
def aliased(monkeypatch):
    """`import bioen.optimize` resolves to bioen_amd.optimize; the caller module is compiled against it."""
    import bioen_amd.optimize as opt
    pkg = types.ModuleType("bioen")
    pkg.__path__ = []
    pkg.optimize = opt
    monkeypatch.setitem(sys.modules, "bioen", pkg)
    monkeypatch.setitem(sys.modules, "bioen.optimize", opt)
    for sub in ("log_weights", "forces", "minimize", "common", "util"):
        monkeypatch.setitem(sys.modules, "bioen.optimize." + sub, getattr(opt, sub))
    caller = types.ModuleType("caller_like_procedure")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exec(compile(CALLER, "caller_like_procedure.py", "exec"), caller.__dict__)
    return caller


def oracle_series(d, method, thetas):
    """what the reference's serial loop computes (oracle restatement of its C path)"""
    from oracle import oracle_binding as O
    from conftest import LBFGS_DEFAULTS
    w0 = np.asarray(d["w0"], dtype=np.float64).ravel()
    w0 = w0 / w0.sum()
    G = np.log(w0) - np.log(w0[-1])                                  # getGs, log_weights.py:113-127
    out, f = [], np.zeros(d["yTilde"].shape[0])
    for th in thetas:
        if method == "log-weights":
            g, fmin, code, it, ev = O.opt_lbfgs_logw(G, G, d["yTilde"], d["YTilde"], th, LBFGS_DEFAULTS)
            out.append((fmin, O.logw_weights(g)[0]))
        else:
            f, fmin, code, it, ev = O.opt_lbfgs_forces(f, w0, d["yTilde"], d["YTilde"], th, LBFGS_DEFAULTS)   # warm start
            out.append((fmin, O.forces_weights(f, w0, d["yTilde"])))
    return out


def install_oracle_fakes(monkeypatch):
    """Fakes of the ctypes layer, one per entry point find_optimum touches; same signatures, oracle numerics."""
    from oracle import oracle_binding as O
    from bioen_amd.optimize.ext import c_bioen
    calls = []

    def flat(x):
        return np.ascontiguousarray(np.asarray(x, dtype=np.float64)).ravel()

    def bioen_log_posterior_logw(gPrime, g, G, yTilde, YTilde, theta, caching=False):
        calls.append("f_logw")
        return O.logw_fdf(flat(gPrime), flat(G), np.asarray(yTilde), flat(YTilde), theta)[0]

    def bioen_opt_lbfgs_logw(g, G, yTilde, YTilde, theta, params):
        calls.append("opt_logw")
        assert isinstance(params["params"], dict) and params["minimizer"] == "lbfgs"
        res, fmin, code, it, ev = O.opt_lbfgs_logw(flat(g), flat(G), np.asarray(yTilde), flat(YTilde), theta, params["params"])
        assert code in (0, 1, 2)
        return res, fmin

    def bioen_log_posterior_forces(forces, w0, yTilde, YTilde, theta, caching=False):
        calls.append("f_forces")
        return O.forces_fdf(flat(forces), flat(w0), np.asarray(yTilde), flat(YTilde), theta)[0]

    def bioen_opt_lbfgs_forces(forces, w0, yTilde, YTilde, theta, params):
        calls.append("opt_forces")
        assert np.asarray(forces).shape == (1, np.asarray(yTilde).shape[0])      # (1, m), forces.py:372-373
        res, fmin, code, it, ev = O.opt_lbfgs_forces(flat(forces), flat(w0), np.asarray(yTilde), flat(YTilde), theta,
                                                     params["params"])
        return res.reshape(1, -1), fmin

    def chi2_and_kl_forces(forces, w0, yTilde, YTilde):
        w = O.forces_weights(flat(forces), flat(w0), np.asarray(yTilde))
        return w, O.chi_squared(w, np.asarray(yTilde), flat(YTilde))[0]

    def get_ave(w, yTilde, YTilde):
        return O.chi_squared(flat(w), np.asarray(yTilde), flat(YTilde))[1]

    class hold(object):                        # find_optimum holds the matrix for the call (one device copy): nothing to hold here
        def __init__(self, yTilde, YTilde=None):
            calls.append("hold")

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    for fn in (bioen_log_posterior_logw, bioen_opt_lbfgs_logw, bioen_log_posterior_forces, bioen_opt_lbfgs_forces,
               chi2_and_kl_forces, get_ave, hold):
        monkeypatch.setattr(c_bioen, fn.__name__, fn)
    return calls


def check_series(res, ref, d, method, thetas):
    n, m = d["yTilde"].shape[1], d["yTilde"].shape[0]
    for r, (fmin_ref, w_ref), th in zip(res, ref, thetas):
        out = r["out_min"]
        assert len(out) == (5 if method == "log-weights" else 7)
        wopt, yopt = out[0], out[1]
        assert wopt.shape == (n, 1) and np.asarray(yopt).shape == (m,)
        assert abs(out[4] - fmin_ref) <= 2e-5 * abs(fmin_ref)          # yaml defaults stop on a 1e-6 plateau
        assert out[4] <= out[3]
        assert np.abs(np.asarray(wopt).ravel() - w_ref).max() <= 2e-2 * w_ref.max()
        assert np.allclose(np.asarray(r["yave"]).ravel(), np.asarray(yopt).ravel(), rtol=1e-9, atol=1e-12)
        assert r["params"]["cache_ytilde_transposed"] is True and r["params"]["use_c_functions"] is True
        if method == "forces":
            assert np.asarray(out[2]).shape == (m,) and out[5] >= 0 and out[6] >= -1e-12      # 1-D, as the reference's
            assert abs(r["chi2"] - out[5]) <= 1e-8 * max(out[5], 1.0)  # chiSqrTerm(wopt) == returned chiSqr


@pytest.mark.parametrize("method,name", [("log-weights", "synth_logw_M37xN500.npz"), ("forces", "synth_forces_M30xN1000.npz")])
def test_procedure_shaped_caller_on_fakes_of_the_c_abi(aliased, monkeypatch, method, name):
    d = load_golden(name)
    if "w0" not in d:
        d["w0"] = np.ones(d["yTilde"].shape[1])
    calls = install_oracle_fakes(monkeypatch)
    thetas = [100.0, 10.0, 1.0]
    obs = Obs(d)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # np.matrix PendingDeprecationWarning, as under the reference
        res = aliased.run_series(options(method, thetas, iterations=2), obs)
    assert obs.updates == 6
    assert calls.count("opt_logw" if method == "log-weights" else "opt_forces") == 6
    check_series(res, oracle_series(d, method, thetas), d, method, thetas)
    import bioen.optimize as through_alias
    import bioen_amd.optimize
    assert through_alias is bioen_amd.optimize and through_alias.log_weights is bioen_amd.optimize.log_weights


@pytest.mark.gpu
@pytest.mark.parametrize("method,name", [("log-weights", "synth_logw_M37xN500.npz"), ("forces", "synth_forces_M30xN1000.npz"),
                                         ("forces", "ref_data_forces_M64xN64.npz")])
def test_procedure_shaped_caller_on_the_device(aliased, method, name):
    import bioen_amd
    assert bioen_amd.device_count() >= 1
    d = load_golden(name)
    if "w0" not in d:
        d["w0"] = np.ones(d["yTilde"].shape[1])
    thetas = [100.0, 10.0, 1.0]
    obs = Obs(d)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = aliased.run_series(options(method, thetas, iterations=1), obs)
    check_series(res, oracle_series(d, method, thetas), d, method, thetas)
    from bioen_amd.optimize.ext import c_bioen
    assert len(c_bioen._CACHE) == 1              # one upload of the np.matrix for the whole series
    c_bioen.clear_cache()
