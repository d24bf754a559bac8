import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    for k in ("theta", "kind"):
        if k in d:
            d[k] = d[k].item()
    if "y" in d and d["y"].size == 0:
        d["y"] = d["yTilde"]
    return d


def golden_files(kind):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.npz"))):
        b = os.path.basename(p)
        if b == "error_codes.npz" or b.startswith("gsl_") or b.startswith("bench_") or b.startswith("deer_"):
            continue
        z = np.load(p, allow_pickle=False)
        if z["kind"].item() == kind:
            out.append(b)
    return out


LOGW_GOLDEN = golden_files("logw")
FORCES_GOLDEN = golden_files("forces")

LBFGS_DEFAULTS = dict(linesearch=2, max_iterations=5000, delta=1e-6, epsilon=1e-6, ftol=1e-5, gtol=0.9,
                      wolfe=0.9, past=10, max_linesearch=100)
LBFGS_TIGHT = dict(LBFGS_DEFAULTS, epsilon=1e-7, delta=1e-11)
# converged: no plateau test, gradient test at 1e-9 -- a run ends AT the optimum (epsilon test, or an
# exhausted line search at the rounding floor); make_golden.py stores the reference's runs as
# lbfgs_conv_* (backtracking-Wolfe) and lbfgs_convmt_* (More-Thuente)
LBFGS_CONV = dict(LBFGS_DEFAULTS, epsilon=1e-9, delta=0.0, past=0, max_iterations=200000)
LBFGS_CONVMT = dict(LBFGS_CONV, linesearch=0)


def tall_forces_problem(M=1100, N=2000, seed=20251):
    """A forces problem with more than 1024 observables (the four passes over row panels), SURVEY 8(d)'s recipe."""
    rng = np.random.default_rng(seed)
    YTrue = rng.uniform(1, 10, M)
    sig_exp, sig_sim = 0.1 * YTrue, 0.5 * YTrue
    YTilde = rng.normal(YTrue, sig_exp) / sig_exp
    yTilde = rng.normal(YTrue[:, None], sig_sim[:, None], (M, N)) / sig_exp[:, None]
    w0 = rng.uniform(0.5, 1.5, N)
    w0 /= w0.sum()
    f0 = 1e-4 * rng.standard_normal(M)
    return dict(yTilde=yTilde, YTilde=YTilde, w0=w0, f0=f0, thetas=[1000.0, 100.0])


def require_reference():
    """-> oracle.ref_binding, the ctypes binding of the reference's own C path (oracle/_ref/libbioen_ref.so, git-ignored,
    built from /root/reference by oracle/Makefile in the build container; it travels to the GPU box with the repository).
    Where a GPU is present its absence is a FAILURE: the parity evidence against the reference binary must not turn
    into skips silently.  Only the GPU-less build container without /root/reference may skip."""
    from oracle import ref_binding
    if ref_binding.available():
        return ref_binding
    gpu = False
    try:
        import bioen_amd
        gpu = bioen_amd.device_count() > 0
    except Exception:
        gpu = False
    msg = ("oracle/_ref/libbioen_ref.so is missing (build it with `make -C oracle` where /root/reference exists; "
           "it is git-ignored but NOT gpurun-ignored and must travel with the repository)")
    if gpu:
        pytest.fail(msg + " -- on a box with a GPU this is a failure, not a skip")
    pytest.skip(msg)


@pytest.fixture(scope="session")
def have_ref():
    from oracle import ref_binding
    return ref_binding.available()
