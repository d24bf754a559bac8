/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99, scalar) of the five GSL 2.5 gradient minimizers the reference
 * can run BioEn under (c_bioen_kernels_logw.c:366-509, c_bioen_kernels_forces.c, selected by
 * gsl_config_params.algorithm, c_bioen_common.h:28-34) and of the reference's driver loop
 * around them.  It is the checker for bioen_amd/csrc/multimin.hpp; the product never links,
 * loads or calls anything in this file.
 *
 * Sources restated (all under /root/reference/third-party/gsl-2.5/):
 *   multimin/directional_minimize.c   take_step, intermediate_point, minimize   (conjugate_*, vector_bfgs)
 *   multimin/conjugate_fr.c, conjugate_pr.c, vector_bfgs.c, steepest_descent.c  (set / iterate)
 *   multimin/vector_bfgs2.c, linear_minimize.c, linear_wrapper.c                (Fletcher line search)
 *   poly/solve_quadratic.c, cblas/source_nrm2_r.h, cblas/source_dot_r.h
 *
 * Parity status: PINNED bit for bit (r02).  GSL cannot be compiled from its sources where they lie (its
 * build needs the generated config.h and the gsl/ header tree), so it is not part of oracle/_ref; but
 * tests/golden/gsl_*.npz hold 72 runs of the REAL GSL 2.5 through the reference's bioen.optimize (built once
 * in the build container under /tmp, tests/golden/make_golden_gsl.py), and driven by the reference's own
 * objective functions (oracle_opt_gsl_refobj below) this restatement reproduces every one of them -- status,
 * iteration count, fmin, x -- to the last bit (tests/test_gsl_golden.py).  Also kept: GSL's own multimin test
 * programme (multimin/test.c:56-76,106-160: Roth, Wood, Rosenbrock x2, SimpleAbs under all five minimizers).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "bioen_oracle.h"

enum { MM_SUCCESS = 0, MM_CONTINUE = -2, MM_EBADTOL = 13, MM_ENOPROG = 27 };   /* err/gsl_errno.h:40-69 */
enum { ALG_CONJUGATE_FR = 0, ALG_CONJUGATE_PR = 1, ALG_BFGS2 = 2, ALG_BFGS = 3, ALG_STEEPEST = 4 };

typedef struct {
    int n;
    double (*f)(const double* x, void* p);
    void (*df)(const double* x, void* p, double* g);
    void (*fdf)(const double* x, void* p, double* f, double* g);
    void* p;
    int nf, ng;   /* function / gradient evaluation counts */
} mm_fn;

static double ev_f(mm_fn* F, const double* x) { F->nf++; return F->f(x, F->p); }
static void ev_df(mm_fn* F, const double* x, double* g) { F->ng++; F->df(x, F->p, g); }
static void ev_fdf(mm_fn* F, const double* x, double* f, double* g) { F->nf++; F->ng++; F->fdf(x, F->p, f, g); }

/* ---- level-1 BLAS as GSL's cblas does it ---------------------------------------------- */
static double v_dot(int n, const double* x, const double* y) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += x[i] * y[i];
    return r;
}
static double v_nrm2(int n, const double* x) {   /* cblas/source_nrm2_r.h: scaled sum of squares */
    double scale = 0.0, ssq = 1.0;
    if (n <= 0) return 0.0;
    if (n == 1) return fabs(x[0]);
    for (int i = 0; i < n; ++i) {
        if (x[i] != 0.0) {
            const double ax = fabs(x[i]);
            if (scale < ax) {
                ssq = 1.0 + ssq * (scale / ax) * (scale / ax);
                scale = ax;
            } else {
                ssq += (ax / scale) * (ax / scale);
            }
        }
    }
    return scale * sqrt(ssq);
}
static void v_axpy(int n, double a, const double* x, double* y) {
    if (a == 0.0) return;   /* cblas/source_axpy_r.h */
    for (int i = 0; i < n; ++i) y[i] += a * x[i];
}
static void v_scal(int n, double a, double* x) { for (int i = 0; i < n; ++i) x[i] *= a; }
static void v_copy(int n, double* dst, const double* src) { memcpy(dst, src, (size_t)n * sizeof(double)); }
static int v_equal(int n, const double* x, const double* y) {
    for (int i = 0; i < n; ++i) if (x[i] != y[i]) return 0;
    return 1;
}
static double v_absmax(int n, const double* x) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) { const double t = fabs(x[i]); if (t > r) r = t; }
    return r;
}

/* ---- directional_minimize.c ------------------------------------------------------------ */
/* :21-30  dx = -step*lambda*p ; x1 = x + dx */
static void step_along(int n, const double* x, const double* p, double step, double lambda, double* x1, double* dx) {
    for (int i = 0; i < n; ++i) dx[i] = 0.0;
    v_axpy(n, -step * lambda, p, dx);
    v_copy(n, x1, x);
    v_axpy(n, 1.0, dx, x1);
}

/* :32-87  shrink (stepa=0, stepc) until a point with f < fa is found */
static void bracket_point(mm_fn* F, const double* x, const double* p, double lambda, double pg, double stepc,
                          double fa, double fc, double* x1, double* dx, double* grad, double* step, double* f) {
    const int n = F->n;
    for (;;) {
        const double u = fabs(pg * lambda * stepc);
        const double stepb = 0.5 * stepc * u / ((fc - fa) + u);
        step_along(n, x, p, stepb, lambda, x1, dx);
        if (v_equal(n, x, x1)) {          /* the trial point did not move */
            *step = 0.0;
            *f = fa;
            ev_df(F, x1, grad);
            return;
        }
        const double fb = ev_f(F, x1);
        if (fb >= fa && stepb > 0.0) {    /* still uphill: shrink */
            fc = fb;
            stepc = stepb;
            continue;
        }
        *step = stepb;
        *f = fb;
        ev_df(F, x1, grad);
        return;
    }
}

/* :89-248  Brent-like refinement inside (stepa, stepb, stepc), at most 10 trial points */
static void line_refine(mm_fn* F, const double* x, const double* p, double lambda, double stepa, double stepb,
                        double stepc, double fa, double fb, double fc, double tol, double* x1, double* dx1,
                        double* x2, double* dx2, double* grad, double* step, double* f, double* gnorm) {
    const int n = F->n;
    double u = stepb, v = stepa, w = stepc;
    double fu = fb, fv = fa, fw = fc;
    double old2 = fabs(w - v), old1 = fabs(v - u);
    v_copy(n, x2, x1);
    v_copy(n, dx2, dx1);
    *f = fb;
    *step = stepb;
    *gnorm = v_nrm2(n, grad);
    for (int iter = 1; iter <= 10; ++iter) {
        const double dw = w - u, dv = v - u;
        double du = 0.0, stepm;
        const double e1 = ((fv - fu) * dw * dw + (fu - fw) * dv * dv);
        const double e2 = 2.0 * ((fv - fu) * dw + (fu - fw) * dv);
        if (e2 != 0.0) du = e1 / e2;
        if (du > 0.0 && du < (stepc - stepb) && fabs(du) < 0.5 * old2) stepm = u + du;
        else if (du < 0.0 && du > (stepa - stepb) && fabs(du) < 0.5 * old2) stepm = u + du;
        else if ((stepc - stepb) > (stepb - stepa)) stepm = 0.38 * (stepc - stepb) + stepb;
        else stepm = stepb - 0.38 * (stepb - stepa);

        step_along(n, x, p, stepm, lambda, x1, dx1);
        const double fm = ev_f(F, x1);
        if (fm > fb) {
            if (fm < fv) { w = v; v = stepm; fw = fv; fv = fm; }
            else if (fm < fw) { w = stepm; fw = fm; }
            if (stepm < stepb) { stepa = stepm; fa = fm; }
            else { stepc = stepm; fc = fm; }
            continue;
        }
        /* fm <= fb (a NaN falls through both tests in GSL and ends the routine) */
        if (!(fm <= fb)) return;
        old2 = old1;
        old1 = fabs(u - stepm);
        w = v; v = u; u = stepm;
        fw = fv; fv = fu; fu = fm;
        v_copy(n, x2, x1);
        v_copy(n, dx2, dx1);
        ev_df(F, x1, grad);
        const double pg = v_dot(n, p, grad);
        const double gnorm1 = v_nrm2(n, grad);
        *f = fm;
        *step = stepm;
        *gnorm = gnorm1;
        if (fabs(pg * lambda / gnorm1) < tol) return;
        if (stepm < stepb) { stepc = stepb; fc = fb; stepb = stepm; fb = fm; }
        else { stepa = stepb; fa = fb; stepb = stepm; fb = fm; }
    }
}

/* ---- linear_minimize.c / linear_wrapper.c (vector_bfgs2) -------------------------------- */
typedef struct {
    mm_fn* F;
    int n;
    const double *x, *g, *p;
    double *x_alpha, *g_alpha;
    double f_alpha, df_alpha;
    double x_key, f_key, df_key, g_key;
} line_cache;

static void lc_moveto(line_cache* w, double alpha) {
    if (alpha == w->x_key) return;
    v_copy(w->n, w->x_alpha, w->x);
    v_axpy(w->n, alpha, w->p, w->x_alpha);
    w->x_key = alpha;
}
static double lc_f(line_cache* w, double alpha) {
    if (alpha == w->f_key) return w->f_alpha;
    lc_moveto(w, alpha);
    w->f_alpha = ev_f(w->F, w->x_alpha);
    w->f_key = alpha;
    return w->f_alpha;
}
static double lc_df(line_cache* w, double alpha) {
    if (alpha == w->df_key) return w->df_alpha;
    lc_moveto(w, alpha);
    if (alpha != w->g_key) {
        ev_df(w->F, w->x_alpha, w->g_alpha);
        w->g_key = alpha;
    }
    w->df_alpha = v_dot(w->n, w->g_alpha, w->p);
    w->df_key = alpha;
    return w->df_alpha;
}
static void lc_fdf(line_cache* w, double alpha, double* f, double* df) {
    if (alpha == w->f_key && alpha == w->df_key) { *f = w->f_alpha; *df = w->df_alpha; return; }
    if (alpha == w->f_key || alpha == w->df_key) { *f = lc_f(w, alpha); *df = lc_df(w, alpha); return; }
    lc_moveto(w, alpha);
    ev_fdf(w->F, w->x_alpha, &w->f_alpha, w->g_alpha);
    w->f_key = alpha;
    w->g_key = alpha;
    w->df_alpha = v_dot(w->n, w->g_alpha, w->p);
    w->df_key = alpha;
    *f = w->f_alpha;
    *df = w->df_alpha;
}
static void lc_restart(line_cache* w) {   /* prepare_wrapper's tail == change_direction */
    v_copy(w->n, w->x_alpha, w->x);
    w->x_key = 0.0;
    w->f_key = 0.0;
    v_copy(w->n, w->g_alpha, w->g);
    w->g_key = 0.0;
    w->df_alpha = v_dot(w->n, w->g_alpha, w->p);
    w->df_key = 0.0;
}

static int quad_roots(double a, double b, double c, double* x0, double* x1) {   /* poly/solve_quadratic.c */
    if (a == 0) {
        if (b == 0) return 0;
        *x0 = -c / b;
        return 1;
    }
    const double disc = b * b - 4 * a * c;
    if (disc > 0) {
        if (b == 0) {
            const double r = sqrt(-c / a);
            *x0 = -r;
            *x1 = r;
        } else {
            const double sgnb = (b > 0 ? 1 : -1);
            const double temp = -0.5 * (b + sgnb * sqrt(disc));
            const double r1 = temp / a, r2 = c / temp;
            if (r1 < r2) { *x0 = r1; *x1 = r2; } else { *x0 = r2; *x1 = r1; }
        }
        return 2;
    }
    if (disc == 0) {
        *x0 = -0.5 * b / a;
        *x1 = -0.5 * b / a;
        return 2;
    }
    return 0;
}

static double poly3(double c0, double c1, double c2, double c3, double z) { return c0 + z * (c1 + z * (c2 + z * c3)); }

static double min_quadratic(double f0, double fp0, double f1, double zl, double zh) {   /* linear_minimize.c:10-33 */
    const double fl = f0 + zl * (fp0 + zl * (f1 - f0 - fp0));
    const double fh = f0 + zh * (fp0 + zh * (f1 - f0 - fp0));
    const double c = 2 * (f1 - f0 - fp0);
    double zmin = zl, fmin = fl;
    if (fh < fmin) { zmin = zh; fmin = fh; }
    if (c > 0) {
        const double z = -fp0 / c;
        if (z > zl && z < zh) {
            const double f = f0 + z * (fp0 + z * (f1 - f0 - fp0));
            if (f < fmin) { zmin = z; fmin = f; }
        }
    }
    return zmin;
}

static double min_cubic(double f0, double fp0, double f1, double fp1, double zl, double zh) {   /* :45-100 */
    const double eta = 3 * (f1 - f0) - 2 * fp0 - fp1;
    const double xi = fp0 + fp1 - 2 * (f1 - f0);
    const double c0 = f0, c1 = fp0, c2 = eta, c3 = xi;
    double zmin = zl, fmin = poly3(c0, c1, c2, c3, zl);
    double z0 = 0, z1 = 0, y;
    y = poly3(c0, c1, c2, c3, zh);
    if (y < fmin) { zmin = zh; fmin = y; }
    const int nr = quad_roots(3 * c3, 2 * c2, c1, &z0, &z1);
    if (nr >= 1 && z0 > zl && z0 < zh) {
        y = poly3(c0, c1, c2, c3, z0);
        if (y < fmin) { zmin = z0; fmin = y; }
    }
    if (nr == 2 && z1 > zl && z1 < zh) {
        y = poly3(c0, c1, c2, c3, z1);
        if (y < fmin) { zmin = z1; fmin = y; }
    }
    return zmin;
}

static double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin,
                          double xmax, int order) {   /* :103-131 */
    double zmin = (xmin - a) / (b - a), zmax = (xmax - a) / (b - a), z;
    if (zmin > zmax) { const double t = zmin; zmin = zmax; zmax = t; }
    if (order > 2 && isfinite(fpb)) z = min_cubic(fa, fpa * (b - a), fb, fpb * (b - a), zmin, zmax);
    else z = min_quadratic(fa, fpa * (b - a), fb, zmin, zmax);
    return a + z * (b - a);
}

/* linear_minimize.c:136-247: Fletcher's bracketing + sectioning */
static int fletcher(line_cache* w, double rho, double sigma, double tau1, double tau2, double tau3, int order,
                    double alpha1, double* alpha_new) {
    double f0, fp0, falpha, falpha_prev, fpalpha = 0.0, fpalpha_prev, delta, alpha_next;
    double alpha = alpha1, alpha_prev = 0.0;
    double a = 0.0, b = alpha, fa, fb = 0.0, fpa, fpb = 0.0;
    const size_t bracket_iters = 100, section_iters = 100;
    size_t i = 0;
    lc_fdf(w, 0.0, &f0, &fp0);
    falpha_prev = f0;
    fpalpha_prev = fp0;
    fa = f0;
    fpa = fp0;
    while (i++ < bracket_iters) {
        falpha = lc_f(w, alpha);
        if (falpha > f0 + alpha * rho * fp0 || falpha >= falpha_prev) {
            a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
            b = alpha; fb = falpha; fpb = NAN;
            break;
        }
        fpalpha = lc_df(w, alpha);
        if (fabs(fpalpha) <= -sigma * fp0) {
            *alpha_new = alpha;
            return MM_SUCCESS;
        }
        if (fpalpha >= 0) {
            a = alpha; fa = falpha; fpa = fpalpha;
            b = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
            break;
        }
        delta = alpha - alpha_prev;
        alpha_next = interpolate(alpha_prev, falpha_prev, fpalpha_prev, alpha, falpha, fpalpha, alpha + delta,
                                 alpha + tau1 * delta, order);
        alpha_prev = alpha;
        falpha_prev = falpha;
        fpalpha_prev = fpalpha;
        alpha = alpha_next;
    }
    while (i++ < section_iters) {
        delta = b - a;
        alpha = interpolate(a, fa, fpa, b, fb, fpb, a + tau2 * delta, b - tau3 * delta, order);
        falpha = lc_f(w, alpha);
        if ((a - alpha) * fpa <= DBL_EPSILON) return MM_ENOPROG;   /* roundoff prevents progress */
        if (falpha > f0 + rho * alpha * fp0 || falpha >= fa) {
            b = alpha; fb = falpha; fpb = NAN;
        } else {
            fpalpha = lc_df(w, alpha);
            if (fabs(fpalpha) <= -sigma * fp0) {
                *alpha_new = alpha;
                return MM_SUCCESS;
            }
            if (((b - a) >= 0 && fpalpha >= 0) || ((b - a) <= 0 && fpalpha <= 0)) {
                b = a; fb = fa; fpb = fpa;
                a = alpha; fa = falpha; fpa = fpalpha;
            } else {
                a = alpha; fa = falpha; fpa = fpalpha;
            }
        }
    }
    return MM_SUCCESS;
}

/* ---- the minimizer object ------------------------------------------------------------------ */
typedef struct {
    int alg, n;
    mm_fn* F;
    double *x, *grad, *dx;   /* fdfminimizer.c: s->x, s->gradient, s->dx */
    double f;
    /* shared state */
    int iter;
    double step, max_step, tol, pnorm, g0norm;
    double *x1, *dx1, *x2, *p, *g0, *x0, *dx0, *dg0;
    /* bfgs2 */
    double delta_f, fp0, rho, sigma, tau1, tau2, tau3;
    int order;
    double *x_alpha, *g_alpha;
    line_cache lc;
    double* pool;
} mm_state;

static int mm_init(mm_state* s, int alg, mm_fn* F, const double* x0, double step_size, double tol) {
    const int n = F->n;
    memset(s, 0, sizeof *s);
    s->alg = alg;
    s->n = n;
    s->F = F;
    s->pool = (double*)calloc((size_t)n * 13, sizeof(double));
    if (!s->pool) return -1;
    double* q = s->pool;
    s->x = q; q += n; s->grad = q; q += n; s->dx = q; q += n;
    s->x1 = q; q += n; s->dx1 = q; q += n; s->x2 = q; q += n; s->p = q; q += n; s->g0 = q; q += n;
    s->x0 = q; q += n; s->dx0 = q; q += n; s->dg0 = q; q += n; s->x_alpha = q; q += n; s->g_alpha = q;
    v_copy(n, s->x, x0);
    s->iter = 0;
    s->step = step_size;
    s->max_step = step_size;
    s->tol = tol;
    ev_fdf(F, s->x, &s->f, s->grad);
    if (alg == ALG_STEEPEST) return 0;                       /* steepest_descent.c:63-77 */
    if (alg == ALG_BFGS2) {                                  /* vector_bfgs2.c:140-186 */
        s->delta_f = 0;
        v_copy(n, s->x0, s->x);
        v_copy(n, s->g0, s->grad);
        s->g0norm = v_nrm2(n, s->g0);
        v_copy(n, s->p, s->grad);
        v_scal(n, -1 / s->g0norm, s->p);
        s->pnorm = v_nrm2(n, s->p);
        s->fp0 = -s->g0norm;
        s->lc.F = F; s->lc.n = n;
        s->lc.x = s->x0; s->lc.g = s->g0; s->lc.p = s->p;
        s->lc.x_alpha = s->x_alpha; s->lc.g_alpha = s->g_alpha;
        s->lc.f_alpha = s->f;
        lc_restart(&s->lc);
        s->rho = 0.01; s->sigma = tol; s->tau1 = 9; s->tau2 = 0.05; s->tau3 = 0.5; s->order = 3;
        return 0;
    }
    /* conjugate_fr.c:93-119, conjugate_pr.c, vector_bfgs.c:141-166 */
    if (alg == ALG_BFGS) v_copy(n, s->x0, s->x);
    v_copy(n, s->p, s->grad);
    v_copy(n, s->g0, s->grad);
    s->pnorm = s->g0norm = v_nrm2(n, s->grad);
    return 0;
}

static void mm_free(mm_state* s) { free(s->pool); s->pool = NULL; }

static int iterate_steepest(mm_state* s) {   /* steepest_descent.c:99-161 */
    const int n = s->n;
    double* g1 = s->dx1;   /* state->g1 */
    const double f0 = s->f;
    double f1, step = s->step;
    int failed = 0;
    const double gnorm = v_nrm2(n, s->grad);
    if (gnorm == 0.0) {
        for (int i = 0; i < n; ++i) s->dx[i] = 0.0;
        return MM_ENOPROG;
    }
    for (;;) {
        for (int i = 0; i < n; ++i) s->dx[i] = 0.0;
        v_axpy(n, -step / gnorm, s->grad, s->dx);
        v_copy(n, s->x1, s->x);
        v_axpy(n, 1.0, s->dx, s->x1);
        if (v_equal(n, s->x, s->x1)) return MM_ENOPROG;
        ev_fdf(s->F, s->x1, &f1, g1);
        if (f1 > f0) {
            failed = 1;
            step *= s->tol;
            continue;
        }
        break;
    }
    step *= failed ? s->tol : 2.0;
    s->step = step;
    v_copy(n, s->x, s->x1);
    v_copy(n, s->grad, g1);
    s->f = f1;
    return MM_SUCCESS;
}

static int iterate_directional(mm_state* s) {   /* conjugate_fr.c:145-250, conjugate_pr.c:149-262, vector_bfgs.c:186-340 */
    const int n = s->n;
    const double fa = s->f;
    double fb, fc, g1norm;
    const double stepa = 0.0, stepc = s->step;
    double stepb;
    if (s->pnorm == 0.0 || s->g0norm == 0.0) {
        for (int i = 0; i < n; ++i) s->dx[i] = 0.0;
        return MM_ENOPROG;
    }
    const double pg = v_dot(n, s->p, s->grad);
    const double dir = (pg >= 0.0) ? +1.0 : -1.0;
    step_along(n, s->x, s->p, stepc, dir / s->pnorm, s->x1, s->dx);
    fc = ev_f(s->F, s->x1);
    if (fc < fa) {   /* plain success: double the step, no line minimisation */
        s->step = stepc * 2.0;
        s->f = fc;
        v_copy(n, s->x, s->x1);
        ev_df(s->F, s->x1, s->grad);
        return MM_SUCCESS;
    }
    bracket_point(s->F, s->x, s->p, dir / s->pnorm, pg, stepc, fa, fc, s->x1, s->dx1, s->grad, &stepb, &fb);
    if (stepb == 0.0) return MM_ENOPROG;
    line_refine(s->F, s->x, s->p, dir / s->pnorm, stepa, stepb, stepc, fa, fb, fc, s->tol, s->x1, s->dx1, s->x2,
                s->dx, s->grad, &s->step, &s->f, &g1norm);
    v_copy(n, s->x, s->x2);
    s->iter = (s->iter + 1) % n;
    if (s->iter == 0) {
        v_copy(n, s->p, s->grad);
        s->pnorm = g1norm;
    } else if (s->alg == ALG_CONJUGATE_FR) {
        const double beta = -pow(g1norm / s->g0norm, 2.0);
        v_scal(n, -beta, s->p);
        v_axpy(n, 1.0, s->grad, s->p);
        s->pnorm = v_nrm2(n, s->p);
    } else if (s->alg == ALG_CONJUGATE_PR) {
        v_axpy(n, -1.0, s->grad, s->g0);                      /* g0' = g0 - g1 */
        const double g0g1 = v_dot(n, s->g0, s->grad);         /* (g0 - g1) . g1 */
        const double beta = g0g1 / (s->g0norm * s->g0norm);
        v_scal(n, -beta, s->p);
        v_axpy(n, 1.0, s->grad, s->p);
        s->pnorm = v_nrm2(n, s->p);
    } else {   /* ALG_BFGS: p' = g1 - A dx - B dg */
        v_copy(n, s->dx0, s->x);
        v_axpy(n, -1.0, s->x0, s->dx0);
        v_copy(n, s->dg0, s->grad);
        v_axpy(n, -1.0, s->g0, s->dg0);
        const double dxg = v_dot(n, s->dx0, s->grad);
        const double dgg = v_dot(n, s->dg0, s->grad);
        const double dxdg = v_dot(n, s->dx0, s->dg0);
        const double dgnorm = v_nrm2(n, s->dg0);
        double A = 0, B = 0;
        if (dxdg != 0) {
            B = dxg / dxdg;
            A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
        }
        v_copy(n, s->p, s->grad);
        v_axpy(n, -A, s->dx0, s->p);
        v_axpy(n, -B, s->dg0, s->p);
        s->pnorm = v_nrm2(n, s->p);
    }
    if (s->alg == ALG_BFGS) {
        v_copy(n, s->g0, s->grad);
        v_copy(n, s->x0, s->x);
        s->g0norm = v_nrm2(n, s->g0);
    } else {
        s->g0norm = g1norm;
        v_copy(n, s->g0, s->grad);
    }
    return MM_SUCCESS;
}

static int iterate_bfgs2(mm_state* s) {   /* vector_bfgs2.c:208-317 */
    const int n = s->n;
    double alpha = 0.0, alpha1;
    const double f0 = s->f;
    if (s->pnorm == 0.0 || s->g0norm == 0.0 || s->fp0 == 0) {
        for (int i = 0; i < n; ++i) s->dx[i] = 0.0;
        return MM_ENOPROG;
    }
    if (s->delta_f < 0) {
        const double del = fmax(-s->delta_f, 10 * DBL_EPSILON * fabs(f0));
        alpha1 = fmin(1.0, 2.0 * del / (-s->fp0));
    } else {
        alpha1 = fabs(s->step);
    }
    const int status = fletcher(&s->lc, s->rho, s->sigma, s->tau1, s->tau2, s->tau3, s->order, alpha1, &alpha);
    if (status != MM_SUCCESS) return status;
    {   /* update_position */
        double fa_, dfa_;
        lc_fdf(&s->lc, alpha, &fa_, &dfa_);
        s->f = s->lc.f_alpha;
        v_copy(n, s->x, s->x_alpha);
        v_copy(n, s->grad, s->g_alpha);
    }
    s->delta_f = s->f - f0;
    v_copy(n, s->dx0, s->x);
    v_axpy(n, -1.0, s->x0, s->dx0);
    v_copy(n, s->dx, s->dx0);
    v_copy(n, s->dg0, s->grad);
    v_axpy(n, -1.0, s->g0, s->dg0);
    const double dxg = v_dot(n, s->dx0, s->grad);
    const double dgg = v_dot(n, s->dg0, s->grad);
    const double dxdg = v_dot(n, s->dx0, s->dg0);
    const double dgnorm = v_nrm2(n, s->dg0);
    double A = 0, B = 0;
    if (dxdg != 0) {
        B = dxg / dxdg;
        A = -(1.0 + dgnorm * dgnorm / dxdg) * B + dgg / dxdg;
    }
    v_copy(n, s->p, s->grad);
    v_axpy(n, -A, s->dx0, s->p);
    v_axpy(n, -B, s->dg0, s->p);
    v_copy(n, s->g0, s->grad);
    v_copy(n, s->x0, s->x);
    s->g0norm = v_nrm2(n, s->g0);
    s->pnorm = v_nrm2(n, s->p);
    const double pg = v_dot(n, s->p, s->grad);
    const double dir = (pg >= 0.0) ? -1.0 : +1.0;
    v_scal(n, dir / s->pnorm, s->p);
    s->pnorm = v_nrm2(n, s->p);
    s->fp0 = v_dot(n, s->p, s->g0);
    lc_restart(&s->lc);
    return MM_SUCCESS;
}

static int mm_iterate(mm_state* s) {
    if (s->alg == ALG_STEEPEST) return iterate_steepest(s);
    if (s->alg == ALG_BFGS2) return iterate_bfgs2(s);
    return iterate_directional(s);
}

/* the reference's driver loop (c_bioen_kernels_logw.c:434-464) with its max-norm stopping test
 * (c_bioen_common.c:112-138) */
static int bioen_driver(int alg, mm_fn* F, const double* x0, const oracle_gsl_config* cfg, double* x_out,
                        double* fmin, oracle_gsl_stats* st) {
    mm_state s;
    if (alg < 0 || alg > 4 || mm_init(&s, alg, F, x0, cfg->step_size, cfg->tol)) return -1;
    int iter = 0, status;
    do {
        status = mm_iterate(&s);
        if (status) break;
        status = cfg->tol < 0.0 ? MM_EBADTOL : (v_absmax(F->n, s.grad) < cfg->tol ? MM_SUCCESS : MM_CONTINUE);
        iter++;
    } while (status == MM_CONTINUE && iter < cfg->max_iterations);
    v_copy(F->n, x_out, s.x);
    *fmin = s.f;
    if (st) { st->iterations = iter; st->f_evaluations = F->nf; st->g_evaluations = F->ng; }
    mm_free(&s);
    return status;
}

/* ---- BioEn objectives ------------------------------------------------------------------------ */
typedef struct { int m, n; const double *yTilde, *YTilde, *fixed; double theta; } bioen_args;

static double logw_f(const double* x, void* p) {
    const bioen_args* a = (const bioen_args*)p;
    return oracle_logw_fdf(a->m, a->n, a->yTilde, a->YTilde, x, a->fixed, a->theta, NULL, NULL);
}
static void logw_fdf_(const double* x, void* p, double* f, double* g) {
    const bioen_args* a = (const bioen_args*)p;
    *f = oracle_logw_fdf(a->m, a->n, a->yTilde, a->YTilde, x, a->fixed, a->theta, g, NULL);
}
static void logw_df(const double* x, void* p, double* g) { double f; logw_fdf_(x, p, &f, g); }

static double forces_f(const double* x, void* p) {
    const bioen_args* a = (const bioen_args*)p;
    return oracle_forces_fdf(a->m, a->n, a->yTilde, a->YTilde, x, a->fixed, a->theta, NULL, NULL);
}
static void forces_fdf_(const double* x, void* p, double* f, double* g) {
    const bioen_args* a = (const bioen_args*)p;
    *f = oracle_forces_fdf(a->m, a->n, a->yTilde, a->YTilde, x, a->fixed, a->theta, g, NULL);
}
static void forces_df(const double* x, void* p, double* g) { double f; forces_fdf_(x, p, &f, g); }

int oracle_opt_gsl_logw(int m, int n, const double* yTilde, const double* YTilde, const double* g0,
                        const double* G, double theta, const oracle_gsl_config* cfg, double* result, double* fmin,
                        oracle_gsl_stats* stats) {
    bioen_args a = {m, n, yTilde, YTilde, G, theta};
    mm_fn F = {n, logw_f, logw_df, logw_fdf_, &a, 0, 0};
    return bioen_driver(cfg->algorithm, &F, g0, cfg, result, fmin, stats);
}

int oracle_opt_gsl_forces(int m, int n, const double* yTilde, const double* YTilde, const double* forces0,
                          const double* w0, double theta, const oracle_gsl_config* cfg, double* result,
                          double* fmin, oracle_gsl_stats* stats) {
    bioen_args a = {m, n, yTilde, YTilde, w0, theta};
    mm_fn F = {m, forces_f, forces_df, forces_fdf_, &a, 0, 0};
    return bioen_driver(cfg->algorithm, &F, forces0, cfg, result, fmin, stats);
}

/* ---- the restated minimizers on the REFERENCE's own objective functions ------------------------
 * The caller hands in the addresses of the reference's C kernels (taken from a build of the
 * reference: oracle/_ref/libbioen_ref.so, or the Cython extension of a full reference build) and
 * the glue below calls them exactly as the reference's GSL callbacks do
 * (c_bioen_kernels_logw.c:274-362: f = _get_weights + _bioen_log_posterior_logw, df = _get_weights +
 * _grad_bioen_log_posterior_logw, fdf = all three; c_bioen_kernels_forces.c:343-428 likewise).
 * With the objective identical to the last bit, any difference to a real GSL run is a difference
 * of the minimizer restatement itself -- this is how tests/golden/make_golden_gsl.py pins it. */
typedef struct {
    const oracle_ref_kernels* k;
    int m, n, caching;
    double *yTilde, *yTildeT, *YTilde, *fixed, *w, *tmp_n, *tmp_m;
    double theta;
} ref_args;

static double rlogw_f(const double* x, void* p) {
    ref_args* a = (ref_args*)p;
    const double s = a->k->get_weights((double*)x, a->w, (size_t)a->n);
    return a->k->logw_f((double*)x, a->fixed, a->yTilde, a->YTilde, a->w, NULL, a->theta, -1, NULL, a->tmp_n, a->tmp_m,
                        a->m, a->n, s);
}
static void rlogw_df(const double* x, void* p, double* g) {
    ref_args* a = (ref_args*)p;
    a->k->get_weights((double*)x, a->w, (size_t)a->n);
    a->k->logw_df((double*)x, a->fixed, a->yTilde, a->YTilde, a->w, g, a->theta, a->caching, a->yTildeT, a->tmp_n,
                  a->tmp_m, a->m, a->n, -1.0);
}
static void rlogw_fdf(const double* x, void* p, double* f, double* g) {
    ref_args* a = (ref_args*)p;
    const double s = a->k->get_weights((double*)x, a->w, (size_t)a->n);
    *f = a->k->logw_f((double*)x, a->fixed, a->yTilde, a->YTilde, a->w, NULL, a->theta, a->caching, a->yTildeT, a->tmp_n,
                      a->tmp_m, a->m, a->n, s);
    a->k->logw_df((double*)x, a->fixed, a->yTilde, a->YTilde, a->w, g, a->theta, a->caching, a->yTildeT, a->tmp_n,
                  a->tmp_m, a->m, a->n, -1.0);
}
static double rforces_f(const double* x, void* p) {
    ref_args* a = (ref_args*)p;
    a->k->forces_weights(a->fixed, a->yTilde, (double*)x, a->w, a->caching, a->yTildeT, a->tmp_n, (size_t)a->m, (size_t)a->n);
    return a->k->forces_f(a->fixed, a->yTilde, a->YTilde, a->w, NULL, a->theta, a->caching, a->yTildeT, a->tmp_n,
                          a->tmp_m, a->m, a->n);
}
static void rforces_df(const double* x, void* p, double* g) {
    ref_args* a = (ref_args*)p;
    a->k->forces_weights(a->fixed, a->yTilde, (double*)x, a->w, a->caching, a->yTildeT, a->tmp_n, (size_t)a->m, (size_t)a->n);
    a->k->forces_df(a->fixed, a->yTilde, a->YTilde, a->w, g, a->theta, a->caching, a->yTildeT, a->tmp_n, a->tmp_m,
                    a->m, a->n);
}
static void rforces_fdf(const double* x, void* p, double* f, double* g) {
    *f = rforces_f(x, p);
    ref_args* a = (ref_args*)p;
    a->k->forces_df(a->fixed, a->yTilde, a->YTilde, a->w, g, a->theta, a->caching, a->yTildeT, a->tmp_n, a->tmp_m,
                    a->m, a->n);
}

/* The same glue as a free-standing objective `int fn(void* handle, const double* x, double* f, double* grad)`
 * (grad == NULL: f alone) so that OTHER minimizer implementations -- the product's, through
 * bioen_hip_multimin_host -- can be driven by the reference's objective too. */
void* oracle_refobj_create(const oracle_ref_kernels* k, int forces, int m, int n, const double* yTilde,
                           const double* yTildeT, const double* YTilde, const double* fixed, double theta) {
    ref_args* a = (ref_args*)malloc(sizeof(ref_args) + sizeof(oracle_ref_kernels) + sizeof(int));
    if (!a) return NULL;
    oracle_ref_kernels* kc = (oracle_ref_kernels*)(a + 1);
    *kc = *k;
    *(int*)(kc + 1) = forces;
    a->k = kc; a->m = m; a->n = n; a->caching = yTildeT != NULL;
    a->yTilde = (double*)yTilde; a->yTildeT = (double*)yTildeT; a->YTilde = (double*)YTilde; a->fixed = (double*)fixed;
    a->theta = theta;
    a->w = (double*)malloc(sizeof(double) * (size_t)(2 * n + m));
    if (!a->w) { free(a); return NULL; }
    a->tmp_n = a->w + n; a->tmp_m = a->tmp_n + n;
    return a;
}
void oracle_refobj_destroy(void* h) {
    if (!h) return;
    free(((ref_args*)h)->w);
    free(h);
}
int oracle_refobj_eval(void* h, const double* x, double* f, double* grad) {
    ref_args* a = (ref_args*)h;
    const int forces = *(const int*)((const oracle_ref_kernels*)(a + 1) + 1);
    if (forces) {
        if (grad) rforces_fdf(x, a, f, grad); else *f = rforces_f(x, a);
    } else {
        if (grad) rlogw_fdf(x, a, f, grad); else *f = rlogw_f(x, a);
    }
    return 0;
}

int oracle_opt_gsl_refobj(const oracle_ref_kernels* k, int forces, int m, int n, const double* yTilde,
                          const double* yTildeT, const double* YTilde, const double* x0, const double* fixed,
                          double theta, const oracle_gsl_config* cfg, double* result, double* fmin,
                          oracle_gsl_stats* stats) {
    ref_args a;
    a.k = k; a.m = m; a.n = n; a.caching = yTildeT != NULL;
    a.yTilde = (double*)yTilde; a.yTildeT = (double*)yTildeT; a.YTilde = (double*)YTilde; a.fixed = (double*)fixed;
    a.theta = theta;
    a.w = (double*)malloc(sizeof(double) * (size_t)(2 * n + m));
    if (!a.w) return -1;
    a.tmp_n = a.w + n; a.tmp_m = a.tmp_n + n;
    mm_fn F = {forces ? m : n, forces ? rforces_f : rlogw_f, forces ? rforces_df : rlogw_df,
               forces ? rforces_fdf : rlogw_fdf, &a, 0, 0};
    const int status = bioen_driver(cfg->algorithm, &F, x0, cfg, result, fmin, stats);
    free(a.w);
    return status;
}

/* ---- GSL's own test functions (multimin/test_funcs.c) ------------------------------------------ */
static double sgn1(double v) { return v >= 0.0 ? 1.0 : -1.0; }   /* GSL_SIGN */

static void tf_eval(int kind, const double* x, double* f, double* g) {
    switch (kind) {
        case 0: {   /* Roth (test_funcs.c:140-180) */
            const double u = x[0], v = x[1];
            const double a = -13.0 + u + ((5.0 - v) * v - 2.0) * v;
            const double b = -29.0 + u + ((v + 1.0) * v - 14.0) * v;
            const double c = -2 + v * (10 - 3 * v), d = -14 + v * (2 + 3 * v);
            if (f) *f = a * a + b * b;
            if (g) { g[0] = 2 * a + 2 * b; g[1] = 2 * a * c + 2 * b * d; }
            break;
        }
        case 1: {   /* Wood (:183-240) */
            const double u1 = x[0], u2 = x[1], u3 = x[2], u4 = x[3];
            const double t1 = u1 * u1 - u2, t2 = u3 * u3 - u4;
            if (f) *f = 100 * t1 * t1 + (1 - u1) * (1 - u1) + 90 * t2 * t2 + (1 - u3) * (1 - u3)
                      + 10.1 * ((1 - u2) * (1 - u2) + (1 - u4) * (1 - u4)) + 19.8 * (1 - u2) * (1 - u4);
            if (g) {
                g[0] = 400 * u1 * t1 - 2 * (1 - u1);
                g[1] = -200 * t1 - 20.2 * (1 - u2) - 19.8 * (1 - u4);
                g[2] = 360 * u3 * t2 - 2 * (1 - u3);
                g[3] = -180 * t2 - 20.2 * (1 - u4) - 19.8 * (1 - u2);
            }
            break;
        }
        case 2: {   /* Rosenbrock, GSL's scaling (:79-133) */
            const double u = x[0], v = x[1], a = u - 1, b = u * u - v;
            if (f) *f = a * a + 10 * b * b;
            if (g) { g[0] = 2 * (u - 1) + 40 * u * b; g[1] = -20 * b; }
            break;
        }
        default: {  /* SimpleAbs (:26-77) */
            const double a = x[0] - 1, b = x[1] - 2;
            if (f) *f = fabs(a) + fabs(b);
            if (g) { g[0] = sgn1(a); g[1] = sgn1(b); }
            break;
        }
    }
}
static double tf_f(const double* x, void* p) { double f; tf_eval(*(int*)p, x, &f, NULL); return f; }
static void tf_df(const double* x, void* p, double* g) { tf_eval(*(int*)p, x, NULL, g); }
static void tf_fdf(const double* x, void* p, double* f, double* g) { tf_eval(*(int*)p, x, f, g); }

int oracle_multimin_testfn_dim(int kind) { return kind == 1 ? 4 : 2; }

/* multimin/test.c:106-160 (test_fdf): step = 0.1 |x0|, tol = 0.1, stop on |g|_2 < 1e-3, at most
 * 5000 iterations, ENOPROG ends the loop.  Returns the last status; the pass criterion is the
 * caller's (status == 0, or |f| <= 1e-5 on CONTINUE / ENOPROG). */
int oracle_selftest_multimin(int algorithm, int kind, const double* x0, double* x_out, double* fmin,
                             oracle_gsl_stats* stats) {
    int k = kind;
    const int n = oracle_multimin_testfn_dim(kind);
    mm_fn F = {n, tf_f, tf_df, tf_fdf, &k, 0, 0};
    mm_state s;
    if (algorithm < 0 || algorithm > 4 || mm_init(&s, algorithm, &F, x0, 0.1 * v_nrm2(n, x0), 0.1)) return -1;
    int iter = 0, status;
    do {
        iter++;
        status = mm_iterate(&s);
        if (status == MM_ENOPROG) break;
        status = v_nrm2(n, s.grad) < 1e-3 ? MM_SUCCESS : MM_CONTINUE;   /* gsl_multimin_test_gradient */
    } while (iter < 5000 && status == MM_CONTINUE);
    v_copy(n, x_out, s.x);
    *fmin = s.f;
    if (stats) { stats->iterations = iter; stats->f_evaluations = F.nf; stats->g_evaluations = F.ng; }
    mm_free(&s);
    return status;
}
