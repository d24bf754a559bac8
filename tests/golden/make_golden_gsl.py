#!/usr/bin/env python3
"""Golden vectors for the GSL minimizers, produced by the REAL reference:
``bioen.optimize`` built from /root/reference with the vendored GSL 2.5 and liblbfgs 1.10.

Run in the BUILD container only.  The reference build is ephemeral (nothing of it is committed
or travels to the GPU box); recipe of SURVEY.md 8(c), all outputs under /tmp/oracle:

  1. cp -r /root/reference/third-party/{liblbfgs-1.10,gsl-2.5} /tmp/oracle/src/ and in each
     ``./configure --prefix=/tmp/oracle/prefix && make -j8 install``
  2. copy bioen/ setup.py setup.cfg requirements.txt MANIFEST.in README.rst to /tmp/oracle/bioen_src and
     ``BIOEN_OPENMP=1 LBFGS_HOME=/tmp/oracle/prefix GSL_HOME=/tmp/oracle/prefix python3 setup.py build_ext --inplace``
  3. ``BIOEN_REF_BUILD=/tmp/oracle/bioen_src python tests/golden/make_golden_gsl.py``

For every committed fixture (tests/golden/{ref,synth}_*.npz hold the inputs) and every GSL algorithm the
reference offers (c_bioen.pyx:176-213) it stores what the reference returns with the yaml defaults
(bioen_optimize.yaml, section gsl: step_size 0.01, tol 0.001, max_iterations 5000) and, for bfgs2,
with tol 1e-7 (`tol` is also the accuracy GSL's line search is asked for, so this setting does not
converge further: it drives the minimizers into their GSL_ENOPROG exit, status 27, another path to pin):

  <tag>_x           gopt (log-weights) / forces_opt        c_bioen.bioen_opt_bfgs_logw / _forces
  <tag>_fmin        its fmin
  <tag>_status      GSL status left in *error by _opt_bfgs_logw / _opt_bfgs_forces
                    (c_bioen_kernels_logw.c:367-509; 0, -2 = GSL_CONTINUE, 27 = GSL_ENOPROG are "success")
  <tag>_iterations  the driver's own iteration count ("Iterations :" line of its verbose output)
  <tag>_wopt        weights at the returned point (reference's _get_weights / _get_weights_from_forces)
  <tag>_restated_bitwise   [oracle, product]: did the restated minimizers (oracle/multimin_oracle.c and the
                    product's bioen_amd/csrc/multimin.hpp through bioen_hip_multimin_host), driven by the
                    reference's OWN objective functions out of the same extension module, return the same
                    status, iteration count, fmin and x -- to the last bit?  (The script stops if not.)

The Python-level call (what a BioEn user runs) and a direct call of the C driver inside the same
extension module (to read the status, which the Python layer only reports through exceptions) must
agree bit for bit; the script asserts that.  fast_openmp = 0, transposed cache on -- the setting of
the reference's reproducibility tests (test_logw_reproducibility.py:14-46).

Only data is written: tests/golden/gsl_<fixture>.npz.
"""
import ctypes as C
import glob
import os
import re
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
BUILD = os.environ.get("BIOEN_REF_BUILD", "/tmp/oracle/bioen_src")
sys.path.insert(0, BUILD)
sys.path.insert(1, ROOT)

ALGS = ["conjugate_fr", "conjugate_pr", "bfgs2", "bfgs", "steepest_descent"]
DEFAULT = dict(step_size=0.01, tol=0.001, max_iterations=5000)
STRICT = dict(step_size=0.01, tol=1e-7, max_iterations=20000)
dp = C.POINTER(C.c_double)


class params_t(C.Structure):          # c_bioen_common.h:44-60
    _fields_ = [("forces", dp), ("w0", dp), ("g", dp), ("G", dp), ("yTilde", dp), ("YTilde", dp), ("w", dp),
                ("result", dp), ("theta", C.c_double), ("yTildeT", dp), ("caching", C.c_int), ("tmp_n", dp),
                ("tmp_m", dp), ("m", C.c_int), ("n", C.c_int)]


class gsl_config_params(C.Structure):  # c_bioen_common.h:62-67
    _fields_ = [("step_size", C.c_double), ("tol", C.c_double), ("max_iterations", C.c_int), ("algorithm", C.c_int)]


class visual_params(C.Structure):      # c_bioen_common.h:89-92
    _fields_ = [("debug", C.c_size_t), ("verbose", C.c_size_t)]


def _p(a):
    return a.ctypes.data_as(dp)


class capture_fd1(object):
    """C printf of the reference goes to file descriptor 1, not sys.stdout."""
    def __enter__(self):
        sys.stdout.flush()
        self.tmp = tempfile.TemporaryFile(mode="w+b")
        self.saved = os.dup(1)
        os.dup2(self.tmp.fileno(), 1)
        return self

    def __exit__(self, *exc):
        C.CDLL(None).fflush(None)
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        self.tmp.seek(0)
        self.text = self.tmp.read().decode("latin1")
        self.tmp.close()


def main():
    warnings.filterwarnings("ignore")
    import bioen.optimize as opt                     # the reference, built per the recipe above
    from bioen.optimize.ext import c_bioen
    assert opt.util.library_gsl() and opt.util.library_lbfgs(), "reference built without GSL"
    assert os.path.realpath(c_bioen.__file__).startswith(os.path.realpath(BUILD))
    opt.minimize.set_fast_openmp_flag(0)
    L = C.CDLL(c_bioen.__file__)
    for name in ("_opt_bfgs_logw", "_opt_bfgs_forces"):
        fn = getattr(L, name)
        fn.restype = C.c_double
        fn.argtypes = [params_t, gsl_config_params, visual_params, C.POINTER(C.c_int)]
    L._get_weights.restype = C.c_double
    L._get_weights.argtypes = [dp, dp, C.c_size_t]
    L._get_weights_from_forces.argtypes = [dp, dp, dp, dp, C.c_int, dp, dp, C.c_size_t, C.c_size_t]
    L._get_weights_from_forces.restype = None

    def direct(kind, x0, aux, yTilde, YTilde, theta, alg, prm):
        m, n = yTilde.shape
        w, tmp_n, tmp_m = np.empty(n), np.empty(n), np.empty(m)
        yT = np.ascontiguousarray(yTilde.T)
        result = np.empty(n if kind == "logw" else m)
        x0 = x0.copy()
        p = params_t()
        if kind == "logw":
            p.g, p.G = _p(x0), _p(aux)
        else:
            p.forces, p.w0 = _p(x0), _p(aux)
        p.yTilde, p.YTilde, p.w, p.result = _p(yTilde), _p(YTilde), _p(w), _p(result)
        p.theta, p.yTildeT, p.caching, p.tmp_n, p.tmp_m, p.m, p.n = float(theta), _p(yT), 1, _p(tmp_n), _p(tmp_m), m, n
        cfg = gsl_config_params(prm["step_size"], prm["tol"], prm["max_iterations"], c_bioen.get_gsl_method(alg))
        err = C.c_int(0)
        with capture_fd1() as cap:
            fmin = getattr(L, "_opt_bfgs_" + kind)(p, cfg, visual_params(0, 1), C.byref(err))
        it = int(re.search(r"Iterations\s*:\s*(\d+)", cap.text).group(1))
        return result, fmin, err.value, it

    def weights(kind, x, aux, yTilde):
        m, n = yTilde.shape
        w = np.empty(n)
        if kind == "logw":
            L._get_weights(_p(x), _p(w), n)
        else:
            tmp_n = np.empty(n)
            L._get_weights_from_forces(_p(aux), _p(yTilde), _p(x), _p(w), 0, None, _p(tmp_n), m, n)
        return w

    from oracle import oracle_binding as O          # test infrastructure
    from bioen_amd import _lib as P                 # the product's minimizers (host backend, no GPU)

    def restated(kind, x0, aux, yTilde, YTilde, theta, alg, prm, ref):
        """both restatements on the reference's objective; ref = (x, fmin, status, iterations) of real GSL"""
        flags = []
        with O.RefObjective(L, kind, aux, yTilde, YTilde, theta) as obj:
            xo, fo, so, ito, _ = O.opt_gsl_refobj(L, kind, x0, aux, yTilde, YTilde, theta, dict(prm, algorithm=alg))
            xp, info = P.multimin_host(obj.fn, obj.handle, x0, alg, prm)
        for (x, f, st, it) in ((xo, fo, so, ito), (xp, info.fmin, info.lbfgs_code, info.iterations)):
            flags.append(bool(np.array_equal(x, ref[0], equal_nan=True) and np.array_equal([f], [ref[1]], equal_nan=True)
                              and st == ref[2] and it == ref[3]))
        return flags

    for path in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
        base = os.path.basename(path)
        if base.startswith("gsl_") or base == "error_codes.npz" or base.startswith("bench_"):
            continue
        z = np.load(path)
        kind = z["kind"].item()
        yTilde = np.ascontiguousarray(z["yTilde"])
        YTilde = np.ascontiguousarray(z["YTilde"]).ravel()
        theta = float(z["theta"])
        if kind == "logw":
            x0, aux = np.ascontiguousarray(z["GInit"]).ravel(), np.ascontiguousarray(z["G"]).ravel()
            pyfn = c_bioen.bioen_opt_bfgs_logw
        else:
            x0, aux = np.ascontiguousarray(z["forces_init"]).ravel(), np.ascontiguousarray(z["w0"]).ravel()
            pyfn = c_bioen.bioen_opt_bfgs_forces
        out = dict(kind=kind, fixture=base)
        for alg in ALGS:
            for tag, prm in ((alg, DEFAULT),) + ((("bfgs2_strict", STRICT),) if alg == "bfgs2" else ()):
                x, fmin, status, it = direct(kind, x0, aux, yTilde, YTilde, theta, alg, prm)
                out[tag + "_x"], out[tag + "_fmin"], out[tag + "_status"], out[tag + "_iterations"] = x, fmin, status, it
                out[tag + "_wopt"] = weights(kind, x, aux, yTilde)
                flags = restated(kind, x0, aux, yTilde, YTilde, theta, alg, prm, (x, fmin, status, it))
                assert all(flags), (base, tag, "restated minimizers [oracle, product] bitwise == GSL:", flags)
                out[tag + "_restated_bitwise"] = np.array(flags)
                # the same run through the Python layer a BioEn user calls (c_bioen.pyx:362-438, 641-716)
                cfg = dict(algorithm=alg, params=dict(prm), cache_ytilde_transposed=True, debug=False, verbose=False)
                with capture_fd1():
                    try:
                        if kind == "logw":
                            xp, fp = pyfn(x0.copy(), aux, yTilde, YTilde, theta, cfg)
                        else:
                            xp, fp = pyfn(x0.copy(), aux, yTilde, YTilde, theta, cfg)
                        ok = True
                    except RuntimeError as e:
                        ok, msg = False, str(e)
                if ok:
                    assert status in (0, -2, 27), (base, tag, status)
                    assert np.array_equal([fp], [fmin], equal_nan=True), (base, tag, fp, fmin)
                    assert np.array_equal(np.asarray(xp).ravel(), x, equal_nan=True), (base, tag)
                else:
                    assert status not in (0, -2, 27) and ("GSL return code: %d" % status) in msg, (base, tag, msg)
                print("%-44s %-16s status %3d  iterations %5d  fmin %.15g" % (base, tag, status, it, fmin))
        np.savez_compressed(os.path.join(HERE, "gsl_" + base), **out)


if __name__ == "__main__":
    main()
