"""theta-series with nuisance-parameter refits on the resident matrix.

BioEn's DEER and scattering workflows alternate, per theta and ``iterations`` times
(``bioen/analyze/procedure.py:62-83``), between

  1. a BioEn optimisation of the weights, and
  2. a 1-D refit of a nuisance parameter -- DEER modulation depth m per spin-label pair
     (``observables.py:146-171, 205-210``), scattering scaling factor c (``:174-188, 212-215``)
     -- by least squares on chi^2, after which the reference REBUILDS yTilde on the host
     (``observables.py:110-143``).

Both parameters enter affinely: yTilde(m) = 1/sigma + m (F-1)/sigma, yTilde(c) = c I/sigma.  With the
m- (c-) independent matrix resident in HBM, step 2 needs one GEMV (yraw = Y . w, on the device) and a
closed-form 1-D least-squares solution per group of rows, and step 1 sees the new parameter through
``Context.set_affine`` -- nothing is rebuilt or re-uploaded.
"""
import numpy as np


def refit_scales(yraw, YTilde, row_offset, groups):
    """argmin over s_g of sum_{i in g} (off_i + s_g * yraw_i - YTilde_i)^2, per group g.

    This is the optimum the reference's ``leastsq(moddepth_fit ...)`` / ``leastsq(coeff_fit ...)``
    converges to (chi^2 is a parabola in the parameter)."""
    yraw = np.asarray(yraw, dtype=np.float64).ravel()
    YT = np.asarray(YTilde, dtype=np.float64).ravel()
    off = np.zeros_like(YT) if row_offset is None else np.asarray(row_offset, dtype=np.float64).ravel()
    out = []
    for idx in groups:
        b = yraw[idx]
        out.append(float(b.dot(YT[idx] - off[idx]) / b.dot(b)))
    return out


def series(ctx, thetas, G, g_init, lbfgs_params, YTilde, groups=None, row_offset=None, scale0=1.0,
           iterations=10, verbose=False, accept_codes=(0, 1, 2)):
    """theta-series with per-group scale refits (DEER: scale = modulation depth of a trace and
    row_offset = 1/sigma, matrix = (F-1)/sigma; scattering: scale = c, row_offset = None,
    matrix = I/sigma).

    ctx      : bioen_amd.Context holding the parameter-independent matrix
    groups   : list of row-index arrays, one per nuisance parameter (default: one group = all rows)
    scale0   : start value(s); the parameters carry over between iterations AND thetas
               (procedure.py:82-83), the log-weights restart from g_init every time (:46,66)
    accept_codes : liblbfgs status codes taken as success (the reference: 0, 1, 2 -- c_bioen.pyx:516-520; runs with
               the plateau test off end at the rounding floor of the line search, -998 / -1000 / -1001, AT the optimum)
    Returns a list of dicts per theta: theta, w, g, fmin, chi2, S, scales, trace (per iteration)."""
    m = ctx.m
    if groups is None:
        groups = [np.arange(m)]
    groups = [np.asarray(ix, dtype=np.int64) for ix in groups]
    scales = [float(scale0)] * len(groups) if np.ndim(scale0) == 0 else [float(s) for s in scale0]
    out = []
    for theta in thetas:
        trace = []
        for it in range(int(iterations)):
            row_scale = np.ones(m)
            for s, ix in zip(scales, groups):
                row_scale[ix] = s
            ctx.set_affine(row_offset, row_scale)
            # the refit needs Y . w only (last_average: 8 m bytes); the N weights travel once per theta
            g, w, info = ctx.opt_lbfgs_logw(g_init, G, float(theta), lbfgs_params, verbose=verbose,
                                            want_weights=(it == int(iterations) - 1))
            if info.lbfgs_code not in accept_codes:
                raise RuntimeError("nuisance.series, liblbfgs return code: %d" % info.lbfgs_code)
            yraw, _ = ctx.last_average()             # raw Y . w at the optimum: already on the device, 8 m bytes back
            trace.append({"scales": list(scales), "fmin": info.fmin, "chi2": info.chi2,
                          "iterations": info.iterations, "evaluations": info.evaluations})
            scales = refit_scales(yraw, YTilde, row_offset, groups)
        out.append({"theta": float(theta), "w": w, "g": g, "fmin": info.fmin, "chi2": info.chi2, "S": -info.kl,
                    "scales": list(scales), "trace": trace})
    ctx.set_affine(None, None)
    return out
