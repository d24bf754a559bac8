/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the BioEn optimizer hot path.
 *
 * See bioen_oracle.h for the rules (who may call this) and the parity status
 * (PINNED against oracle/_ref and the reference's *.ref known answers).
 *
 * This is a restatement, not a copy: the objective and gradient are written
 * in the closed forms of SURVEY.md section 8(a) (max-shifted softmax, one
 * forward pass, one adjoint pass over the row-major matrix, no transposed
 * copy), and the minimiser is a compact L-BFGS written from the algorithm's
 * definition.  Each function names the reference lines whose behaviour it
 * reproduces.  Everything is IEEE double.
 */
#include "bioen_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* liblbfgs status codes, /root/reference/third-party/liblbfgs-1.10/include/lbfgs.h:76-147 */
enum {
    ST_CONVERGED = 0,
    ST_STOP = 1,
    ST_ALREADY_MINIMIZED = 2,
    ERR_OUTOFMEMORY = -1022,
    ERR_INVALID_N = -1020,
    ERR_INVALID_EPSILON = -1017,
    ERR_INVALID_TESTPERIOD = -1016,
    ERR_INVALID_DELTA = -1015,
    ERR_INVALID_LINESEARCH = -1014,
    ERR_INVALID_FTOL = -1011,
    ERR_INVALID_WOLFE = -1010,
    ERR_INVALID_GTOL = -1009,
    ERR_INVALID_MAXLINESEARCH = -1007,
    ERR_OUTOFINTERVAL = -1003,
    ERR_INCORRECT_TMINMAX = -1002,
    ERR_ROUNDING_ERROR = -1001,
    ERR_MINIMUMSTEP = -1000,
    ERR_MAXIMUMSTEP = -999,
    ERR_MAXIMUMLINESEARCH = -998,
    ERR_MAXIMUMITERATION = -997,
    ERR_WIDTHTOOSMALL = -996,
    ERR_INVALIDPARAMETERS = -995,
    ERR_INCREASEGRADIENT = -994
};

/* values liblbfgs keeps at their defaults because BioEn never overrides them
 * (lbfgs.c:113-118 `_defparam`; c_bioen_kernels_logw.c:604-617) */
#define HISTORY 6
#define MIN_STEP 1e-20
#define MAX_STEP 1e20
#define XTOL 1e-16

/* ------------------------------------------------------------------ */
/* objective / gradient                                                */
/* ------------------------------------------------------------------ */

/* A1, c_bioen_kernels_logw.c:55-94.  The reference has no max-shift; shifting
 * is mathematically identical and only guards against overflow. */
double oracle_logw_weights(const double* g, double* w, size_t n) {
    double gmax = -DBL_MAX;
    for (size_t j = 0; j < n; ++j)
        if (g[j] > gmax) gmax = g[j];
    double s = 0.0;
    for (size_t j = 0; j < n; ++j) {
        w[j] = exp(g[j] - gmax);
        s += w[j];
    }
    const double inv = 1.0 / s;
    for (size_t j = 0; j < n; ++j) w[j] *= inv;
    return gmax + log(s);
}

/* forward pass: out[i] = sum_j yTilde[i,j] v[j]   (c_bioen_common.c:76-86,
 * c_bioen_kernels_forces.c:93-109) */
/* small problems run serially: a 256-thread team costs more than the loop */
#define PAR_MIN ((size_t)1 << 20)

static void matvec(const double* yTilde, const double* v, double* out, size_t m, size_t n) {
#pragma omp parallel for schedule(static) if (m * n >= PAR_MIN)
    for (size_t i = 0; i < m; ++i) {
        const double* row = yTilde + i * n;
        double acc = 0.0;
        for (size_t j = 0; j < n; ++j) acc += row[j] * v[j];
        out[i] = acc;
    }
}

/* centred forward pass: out[i] = sum_j (yTilde[i,j] - ybar[i]) v[j]
 * (c_bioen_kernels_forces.c:330-338 keeps the centring inside the sum) */
static void matvec_centred(const double* yTilde, const double* ybar, const double* v, double* out, size_t m,
                           size_t n) {
#pragma omp parallel for schedule(static) if (m * n >= PAR_MIN)
    for (size_t i = 0; i < m; ++i) {
        const double* row = yTilde + i * n;
        const double yb = ybar[i];
        double acc = 0.0;
        for (size_t j = 0; j < n; ++j) acc += (row[j] - yb) * v[j];
        out[i] = acc;
    }
}

/* adjoint pass: out[j] = sum_i yTilde[i,j] u[i], walking the row-major matrix
 * by rows (the reference walks a transposed copy instead,
 * c_bioen_kernels_logw.c:185-195; same sums, different order). */
/* ybar == NULL: plain; else the centred sum  out[j] = sum_i u[i] (yTilde[i,j] - ybar[i])
 * (c_bioen_kernels_logw.c:190) */
static void matvec_t(const double* yTilde, const double* u, const double* ybar, double* out, size_t m, size_t n) {
#pragma omp parallel if (m * n >= PAR_MIN)
    {
        const size_t chunk = 2048;
#pragma omp for schedule(static)
        for (size_t j0 = 0; j0 < n; j0 += chunk) {
            const size_t j1 = (j0 + chunk < n) ? j0 + chunk : n;
            for (size_t j = j0; j < j1; ++j) out[j] = 0.0;
            for (size_t i = 0; i < m; ++i) {
                const double ui = u[i];
                const double yb = ybar ? ybar[i] : 0.0;
                const double* row = yTilde + i * n;
                for (size_t j = j0; j < j1; ++j) out[j] += ui * (row[j] - yb);
            }
        }
    }
}

/* A4, c_bioen_common.c:70-108 */
double oracle_chi_squared(const double* w, const double* yTilde, const double* YTilde,
                          double* yave, size_t m, size_t n) {
    double* tmp = yave ? yave : (double*)malloc(m * sizeof(double));
    matvec(yTilde, w, tmp, m, n);
    double val = 0.0;
    for (size_t i = 0; i < m; ++i) {
        const double r = tmp[i] - YTilde[i];
        val += r * r;
    }
    if (!yave) free(tmp);
    return 0.5 * val;
}

/* A1 + A3 + A4 + A6 = interface_lbfgs_logw, c_bioen_kernels_logw.c:525-561.
 *
 *   L      = theta * ( sum_j w_j (g_j - G_j) - log s + log s0 ) + 0.5 |r|^2
 *   dL/dg_k = theta w_k [ (g_k - G_k) - P ] + w_k sum_i r_i (yTilde_ik - ybar_i)
 * with r = yTilde w - YTilde, ybar = yTilde w, P = sum_j w_j (g_j - G_j).
 * (c_bioen_kernels_logw.c:96-127 prior, :185-216 gradient, centred as in :190.) */
double oracle_logw_fdf(int m_, int n_, const double* yTilde, const double* YTilde,
                       const double* g, const double* G, double theta,
                       double* grad, double* w_out) {
    const size_t m = (size_t)m_, n = (size_t)n_;
    double* w = w_out ? w_out : (double*)malloc(n * sizeof(double));
    double* r = (double*)malloc(m * sizeof(double));
    double* ybar = (double*)malloc(m * sizeof(double));

    const double logs = oracle_logw_weights(g, w, n);

    /* log s0 = log sum exp(G)   (c_bioen_kernels_logw.c:29-53,122) */
    double Gmax = -DBL_MAX;
    for (size_t j = 0; j < n; ++j)
        if (G[j] > Gmax) Gmax = G[j];
    double s0 = 0.0;
    for (size_t j = 0; j < n; ++j) s0 += exp(G[j] - Gmax);
    const double logs0 = Gmax + log(s0);

    double P = 0.0;
    for (size_t j = 0; j < n; ++j) P += (g[j] - G[j]) * w[j];
    const double prior = theta * (P - logs + logs0);

    matvec(yTilde, w, ybar, m, n);
    double chi = 0.0;
    for (size_t i = 0; i < m; ++i) {
        r[i] = ybar[i] - YTilde[i];
        chi += r[i] * r[i];
    }
    const double f = prior + 0.5 * chi;

    if (grad) {
        matvec_t(yTilde, r, ybar, grad, m, n);
        for (size_t k = 0; k < n; ++k)
            grad[k] = w[k] * (theta * ((g[k] - G[k]) - P) + grad[k]);
    }
    free(ybar);
    free(r);
    if (!w_out) free(w);
    return f;
}

/* F1, c_bioen_kernels_forces.c:111-224 */
void oracle_forces_weights(int m_, int n_, const double* yTilde, const double* forces,
                           const double* w0, double* w) {
    const size_t m = (size_t)m_, n = (size_t)n_;
    matvec_t(yTilde, forces, NULL, w, m, n);
    double xmax = -DBL_MAX;
    for (size_t j = 0; j < n; ++j)
        if (w[j] > xmax) xmax = w[j];
    double s = 0.0;
    for (size_t j = 0; j < n; ++j) {
        w[j] = w0[j] * exp(w[j] - xmax);
        s += w[j];
    }
    const double inv = 1.0 / s;
    for (size_t j = 0; j < n; ++j) w[j] = inv * w[j];
}

/* F1 + F2 + F3 = interface_lbfgs_forces, c_bioen_kernels_forces.c:43-76.
 *   L        = theta sum_j w_j log(w_j / w0_j) + 0.5 |r|^2      (:227-277)
 *   dL/df_i  = sum_j (yTilde_ij - ybar_i) t_j,
 *   t_j      = w_j [ theta (1 + log(w_j/w0_j)) + (yTilde^T r)_j ]  (:280-340)
 * Entries with w_j < DBL_MIN or w0_j < DBL_MIN contribute log-term 0 (:250-254,323-325). */
double oracle_forces_fdf(int m_, int n_, const double* yTilde, const double* YTilde,
                         const double* forces, const double* w0, double theta,
                         double* grad, double* w_out) {
    const size_t m = (size_t)m_, n = (size_t)n_;
    double* w = w_out ? w_out : (double*)malloc(n * sizeof(double));
    double* t = (double*)malloc(n * sizeof(double));
    double* r = (double*)malloc(m * sizeof(double));
    double* ybar = (double*)malloc(m * sizeof(double));

    oracle_forces_weights(m_, n_, yTilde, forces, w0, w);
    matvec(yTilde, w, ybar, m, n);
    double chi = 0.0;
    for (size_t i = 0; i < m; ++i) {
        r[i] = ybar[i] - YTilde[i];
        chi += r[i] * r[i];
    }
    double kl = 0.0;
    for (size_t j = 0; j < n; ++j)
        if (w[j] >= DBL_MIN && w0[j] >= DBL_MIN) kl += (log(w[j]) - log(w0[j])) * w[j];
    const double f = theta * kl + 0.5 * chi;

    if (grad) {
        matvec_t(yTilde, r, NULL, t, m, n);
        for (size_t j = 0; j < n; ++j) {
            double d = 1.0;
            if (w[j] >= DBL_MIN && w0[j] >= DBL_MIN) d += log(w[j]) - log(w0[j]);
            t[j] = (d * theta + t[j]) * w[j];
        }
        matvec_centred(yTilde, ybar, t, grad, m, n);
    }
    free(ybar);
    free(r);
    free(t);
    if (!w_out) free(w);
    return f;
}

/* ------------------------------------------------------------------ */
/* L-BFGS (liblbfgs 1.10 semantics)                                    */
/* ------------------------------------------------------------------ */

typedef double (*eval_fn)(void* ctx, const double* x, double* grad);

static double dot(const double* a, const double* b, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

typedef struct {
    int n;
    eval_fn eval;
    void* ctx;
    const oracle_lbfgs_config* cfg;
    int evaluations;
} problem_t;

static double evaluate(problem_t* p, const double* x, double* g) {
    p->evaluations++;
    return p->eval(p->ctx, x, g);
}

/* lbfgs.c:645-734 -- backtracking with Armijo (1), Wolfe (2) or strong Wolfe (3) exit. */
static int search_backtracking(problem_t* p, double* x, double* f, double* g, const double* d,
                               double* stp, const double* xp) {
    const oracle_lbfgs_config* c = p->cfg;
    const int n = p->n;
    if (*stp <= 0.0) return ERR_INVALIDPARAMETERS;
    const double dginit = dot(g, d, n);
    if (!(dginit <= 0.0)) return ERR_INCREASEGRADIENT;   /* NaN too: what the reference's -ffast-math build of lbfgs.c:671 / :845 does (a direction out of a degenerate pair, y.s = 0, beyond the rounding floor) */
    const double finit = *f;
    const double dgtest = c->ftol * dginit;
    for (int count = 1;; ++count) {
        for (int i = 0; i < n; ++i) x[i] = xp[i] + (*stp) * d[i];
        *f = evaluate(p, x, g);
        double factor;
        if (*f > finit + (*stp) * dgtest) {
            factor = 0.5;
        } else {
            if (c->linesearch == 1) return count;
            const double dg = dot(g, d, n);
            if (dg < c->wolfe * dginit) {
                factor = 2.1;
            } else {
                if (c->linesearch == 2) return count;
                if (dg > -c->wolfe * dginit)
                    factor = 0.5;
                else
                    return count;
            }
        }
        if (*stp < MIN_STEP) return ERR_MINIMUMSTEP;
        if (*stp > MAX_STEP) return ERR_MAXIMUMSTEP;
        if (c->max_linesearch <= count) return ERR_MAXIMUMLINESEARCH;
        *stp *= factor;
    }
}

/* minimiser of the cubic through (u,fu,du),(v,fv,dv); lbfgs.c:996-1011 */
static double cubic_min(double u, double fu, double du, double v, double fv, double dv) {
    const double d = v - u;
    const double th = (fu - fv) * 3.0 / d + du + dv;
    double s = fabs(th);
    if (fabs(du) > s) s = fabs(du);
    if (fabs(dv) > s) s = fabs(dv);
    const double a = th / s;
    double gam = s * sqrt(a * a - (du / s) * (dv / s));
    if (v < u) gam = -gam;
    const double p = gam - du + th;
    const double q = gam - du + gam + dv;
    return u + (p / q) * d;
}

/* safeguarded variant used when the derivative magnitude shrinks; lbfgs.c:1024-1045 */
static double cubic_min_bounded(double u, double fu, double du, double v, double fv, double dv,
                                double lo, double hi) {
    const double d = v - u;
    const double th = (fu - fv) * 3.0 / d + du + dv;
    double s = fabs(th);
    if (fabs(du) > s) s = fabs(du);
    if (fabs(dv) > s) s = fabs(dv);
    const double a = th / s;
    double rad = a * a - (du / s) * (dv / s);
    if (rad < 0.0) rad = 0.0;
    double gam = s * sqrt(rad);
    if (u < v) gam = -gam;
    const double p = gam - dv + th;
    const double q = gam - dv + gam + du;
    const double r = p / q;
    if (r < 0.0 && gam != 0.0) return v - r * d;
    return (a < 0.0) ? hi : lo;
}

/* quadratic through (u,fu,du),(v,fv); lbfgs.c:1056-1058 */
static double quad_min(double u, double fu, double du, double v, double fv) {
    const double a = v - u;
    return u + du / ((fu - fv) / a + du) / 2.0 * a;
}

/* secant on the derivatives; lbfgs.c:1068-1070 */
static double secant_min(double u, double du, double v, double dv) {
    const double a = u - v;
    return v + dv / (dv - du) * a;
}

/* More-Thuente trial step + interval update, lbfgs.c:1125-1296.
 * (x,fx,dx): best step so far; (y,fy,dy): other end; (t,ft,dt): trial. */
static int trial_interval(double* x, double* fx, double* dx, double* y, double* fy, double* dy,
                          double* t, double ft, double dt, double tmin, double tmax, int* brackt) {
    const int opposite = (dt * (*dx / fabs(*dx)) < 0.0);
    int bound;
    double newt, mc, mq;

    if (*brackt) {
        const double lo = (*x < *y) ? *x : *y, hi = (*x < *y) ? *y : *x;
        if (*t <= lo || hi <= *t) return ERR_OUTOFINTERVAL;
        if (0.0 <= *dx * (*t - *x)) return ERR_INCREASEGRADIENT;
        if (tmax < tmin) return ERR_INCORRECT_TMINMAX;
    }

    if (*fx < ft) { /* higher value: minimum bracketed */
        *brackt = 1;
        bound = 1;
        mc = cubic_min(*x, *fx, *dx, *t, ft, dt);
        mq = quad_min(*x, *fx, *dx, *t, ft);
        newt = (fabs(mc - *x) < fabs(mq - *x)) ? mc : mc + 0.5 * (mq - mc);
    } else if (opposite) { /* lower value, derivative changed sign: bracketed */
        *brackt = 1;
        bound = 0;
        mc = cubic_min(*x, *fx, *dx, *t, ft, dt);
        mq = secant_min(*x, *dx, *t, dt);
        newt = (fabs(mc - *t) > fabs(mq - *t)) ? mc : mq;
    } else if (fabs(dt) < fabs(*dx)) { /* lower value, same sign, derivative shrinks */
        bound = 1;
        mc = cubic_min_bounded(*x, *fx, *dx, *t, ft, dt, tmin, tmax);
        mq = secant_min(*x, *dx, *t, dt);
        if (*brackt)
            newt = (fabs(*t - mc) < fabs(*t - mq)) ? mc : mq;
        else
            newt = (fabs(*t - mc) > fabs(*t - mq)) ? mc : mq;
    } else { /* lower value, same sign, derivative does not shrink */
        bound = 0;
        if (*brackt)
            newt = cubic_min(*t, ft, dt, *y, *fy, *dy);
        else
            newt = (*x < *t) ? tmax : tmin;
    }

    if (*fx < ft) {
        *y = *t; *fy = ft; *dy = dt;
    } else {
        if (opposite) { *y = *x; *fy = *fx; *dy = *dx; }
        *x = *t; *fx = ft; *dx = dt;
    }

    if (tmax < newt) newt = tmax;
    if (newt < tmin) newt = tmin;
    if (*brackt && bound) {
        mq = *x + 0.66 * (*y - *x);
        if (*x < *y) { if (mq < newt) newt = mq; }
        else         { if (newt < mq) newt = mq; }
    }
    *t = newt;
    return 0;
}

/* lbfgs.c:812-976 */
static int search_morethuente(problem_t* p, double* x, double* f, double* g, const double* d,
                              double* stp, const double* xp) {
    const oracle_lbfgs_config* c = p->cfg;
    const int n = p->n;
    if (*stp <= 0.0) return ERR_INVALIDPARAMETERS;
    const double dginit = dot(g, d, n);
    if (!(dginit <= 0.0)) return ERR_INCREASEGRADIENT;   /* NaN too: what the reference's -ffast-math build of lbfgs.c:671 / :845 does (a direction out of a degenerate pair, y.s = 0, beyond the rounding floor) */

    int brackt = 0, stage1 = 1, uinfo = 0, count = 0;
    const double finit = *f, dgtest = c->ftol * dginit;
    double width = MAX_STEP - MIN_STEP, prev_width = 2.0 * width;
    double stx = 0.0, fx = finit, dgx = dginit;
    double sty = 0.0, fy = finit, dgy = dginit;

    for (;;) {
        double stmin, stmax;
        if (brackt) {
            stmin = (stx < sty) ? stx : sty;
            stmax = (stx < sty) ? sty : stx;
        } else {
            stmin = stx;
            stmax = *stp + 4.0 * (*stp - stx);
        }
        if (*stp < MIN_STEP) *stp = MIN_STEP;
        if (MAX_STEP < *stp) *stp = MAX_STEP;
        if ((brackt && ((*stp <= stmin || stmax <= *stp) || c->max_linesearch <= count + 1 || uinfo != 0)) ||
            (brackt && (stmax - stmin <= XTOL * stmax)))
            *stp = stx;

        for (int i = 0; i < n; ++i) x[i] = xp[i] + (*stp) * d[i];
        *f = evaluate(p, x, g);
        double dg = dot(g, d, n);
        const double ftest1 = finit + (*stp) * dgtest;
        ++count;

        if (brackt && ((*stp <= stmin || stmax <= *stp) || uinfo != 0)) return ERR_ROUNDING_ERROR;
        if (*stp == MAX_STEP && *f <= ftest1 && dg <= dgtest) return ERR_MAXIMUMSTEP;
        if (*stp == MIN_STEP && (ftest1 < *f || dgtest <= dg)) return ERR_MINIMUMSTEP;
        if (brackt && (stmax - stmin) <= XTOL * stmax) return ERR_WIDTHTOOSMALL;
        if (c->max_linesearch <= count) return ERR_MAXIMUMLINESEARCH;
        if (*f <= ftest1 && fabs(dg) <= c->gtol * (-dginit)) return count;

        const double mintol = (c->ftol < c->gtol) ? c->ftol : c->gtol;
        if (stage1 && *f <= ftest1 && mintol * dginit <= dg) stage1 = 0;

        if (stage1 && ftest1 < *f && *f <= fx) {
            /* work on the modified function psi(t) = f(t) - t*dgtest */
            double fm = *f - (*stp) * dgtest, dgm = dg - dgtest;
            double fxm = fx - stx * dgtest, dgxm = dgx - dgtest;
            double fym = fy - sty * dgtest, dgym = dgy - dgtest;
            uinfo = trial_interval(&stx, &fxm, &dgxm, &sty, &fym, &dgym, stp, fm, dgm, stmin, stmax, &brackt);
            fx = fxm + stx * dgtest;
            fy = fym + sty * dgtest;
            dgx = dgxm + dgtest;
            dgy = dgym + dgtest;
        } else {
            uinfo = trial_interval(&stx, &fx, &dgx, &sty, &fy, &dgy, stp, *f, dg, stmin, stmax, &brackt);
        }

        if (brackt) {
            if (0.66 * prev_width <= fabs(sty - stx)) *stp = stx + 0.5 * (sty - stx);
            prev_width = width;
            width = fabs(sty - stx);
        }
    }
}

/* lbfgs.c:245-641 (orthant-wise branch omitted: BioEn never sets orthantwise_c) */
static int lbfgs_minimize(problem_t* p, double* x, double* fx_out, oracle_lbfgs_stats* stats) {
    const oracle_lbfgs_config* c = p->cfg;
    const int n = p->n;
    int iterations = 0;
    int status;

    /* parameter checks in liblbfgs' order, lbfgs.c:285-331 */
    if (n <= 0) return ERR_INVALID_N;
    if (c->epsilon < 0.0) return ERR_INVALID_EPSILON;
    if (c->past < 0) return ERR_INVALID_TESTPERIOD;
    if (c->delta < 0.0) return ERR_INVALID_DELTA;
    if (c->ftol < 0.0) return ERR_INVALID_FTOL;
    if (c->linesearch == 2 || c->linesearch == 3)
        if (c->wolfe <= c->ftol || 1.0 <= c->wolfe) return ERR_INVALID_WOLFE;
    if (c->gtol < 0.0) return ERR_INVALID_GTOL;
    if (c->max_linesearch <= 0) return ERR_INVALID_MAXLINESEARCH;
    if (c->linesearch < 0 || c->linesearch > 3) return ERR_INVALID_LINESEARCH;

    double* buf = (double*)malloc(sizeof(double) * (size_t)n * (4 + 2 * HISTORY));
    if (!buf) return ERR_OUTOFMEMORY;
    double *xp = buf, *g = buf + n, *gp = buf + 2 * (size_t)n, *d = buf + 3 * (size_t)n;
    double* S[HISTORY];
    double* Y[HISTORY];
    double ys_hist[HISTORY], alpha[HISTORY];
    for (int i = 0; i < HISTORY; ++i) {
        S[i] = buf + (4 + 2 * (size_t)i) * n;
        Y[i] = buf + (5 + 2 * (size_t)i) * n;
        ys_hist[i] = alpha[i] = 0.0;
    }
    double* pf = (c->past > 0) ? (double*)malloc(sizeof(double) * (size_t)c->past) : NULL;

    double fx = evaluate(p, x, g);
    if (pf) pf[0] = fx;
    for (int i = 0; i < n; ++i) d[i] = -g[i];

    double xnorm = sqrt(dot(x, x, n)), gnorm = sqrt(dot(g, g, n));
    if (xnorm < 1.0) xnorm = 1.0;
    if (!(gnorm / xnorm > c->epsilon)) {      /* NaN ends the run: what the reference's -ffast-math build of lbfgs.c:447 does (tools/nan_probe.py) */
        status = ST_ALREADY_MINIMIZED;
        goto done;
    }
    double step = 1.0 / sqrt(dot(d, d, n));
    int k = 1, end = 0;

    for (;;) {
        memcpy(xp, x, sizeof(double) * (size_t)n);
        memcpy(gp, g, sizeof(double) * (size_t)n);

        int ls = (c->linesearch == 0) ? search_morethuente(p, x, &fx, g, d, &step, xp)
                                      : search_backtracking(p, x, &fx, g, d, &step, xp);
        if (ls < 0) {
            memcpy(x, xp, sizeof(double) * (size_t)n);
            memcpy(g, gp, sizeof(double) * (size_t)n);
            status = ls;
            goto done;
        }
        xnorm = sqrt(dot(x, x, n));
        gnorm = sqrt(dot(g, g, n));
        ++iterations; /* the progress callback, c_bioen_kernels_logw.c:565-576 */

        if (xnorm < 1.0) xnorm = 1.0;
        if (!(gnorm / xnorm > c->epsilon)) { status = ST_CONVERGED; break; }

        if (pf) {
            if (c->past <= k) {
                const double rate = (pf[k % c->past] - fx) / fx;
                if (rate < c->delta) { status = ST_STOP; break; }
            }
            pf[k % c->past] = fx;
        }
        if (c->max_iterations != 0 && c->max_iterations < k + 1) { status = ERR_MAXIMUMITERATION; break; }

        double* s = S[end];
        double* y = Y[end];
        for (int i = 0; i < n; ++i) { s[i] = x[i] - xp[i]; y[i] = g[i] - gp[i]; }
        const double ys = dot(y, s, n), yy = dot(y, y, n);
        ys_hist[end] = ys;

        const int bound = (HISTORY <= k) ? HISTORY : k;
        ++k;
        end = (end + 1) % HISTORY;

        for (int i = 0; i < n; ++i) d[i] = -g[i];
        int j = end;
        for (int b = 0; b < bound; ++b) {
            j = (j + HISTORY - 1) % HISTORY;
            alpha[j] = dot(S[j], d, n) / ys_hist[j];
            for (int i = 0; i < n; ++i) d[i] -= alpha[j] * Y[j][i];
        }
        const double scale = ys / yy;
        for (int i = 0; i < n; ++i) d[i] *= scale;
        for (int b = 0; b < bound; ++b) {
            const double beta = dot(Y[j], d, n) / ys_hist[j];
            const double coef = alpha[j] - beta;
            for (int i = 0; i < n; ++i) d[i] += coef * S[j][i];
            j = (j + 1) % HISTORY;
        }
        step = 1.0;
    }

done:
    *fx_out = fx;
    if (stats) {
        stats->iterations = iterations;
        stats->evaluations = p->evaluations;
    }
    free(pf);
    free(buf);
    return status;
}

/* ------------------------------------------------------------------ */
/* drivers: _opt_lbfgs_logw (c_bioen_kernels_logw.c:581-669) and         */
/*          _opt_lbfgs_forces (c_bioen_kernels_forces.c:574-662)         */
/* ------------------------------------------------------------------ */

typedef struct {
    int m, n;
    const double *yTilde, *YTilde, *fixed; /* fixed = G (logw) or w0 (forces) */
    double theta;
} model_t;

static double eval_logw(void* ctx, const double* x, double* grad) {
    const model_t* q = (const model_t*)ctx;
    return oracle_logw_fdf(q->m, q->n, q->yTilde, q->YTilde, x, q->fixed, q->theta, grad, NULL);
}

static double eval_forces(void* ctx, const double* x, double* grad) {
    const model_t* q = (const model_t*)ctx;
    return oracle_forces_fdf(q->m, q->n, q->yTilde, q->YTilde, x, q->fixed, q->theta, grad, NULL);
}

int oracle_opt_lbfgs_logw(int m, int n, const double* yTilde, const double* YTilde,
                          const double* g0, const double* G, double theta,
                          const oracle_lbfgs_config* cfg, double* result, double* fmin,
                          oracle_lbfgs_stats* stats) {
    model_t q = {m, n, yTilde, YTilde, G, theta};
    problem_t p = {n, eval_logw, &q, cfg, 0};
    memcpy(result, g0, sizeof(double) * (size_t)n);
    *fmin = 0.0;
    if (stats) stats->iterations = stats->evaluations = 0;
    return lbfgs_minimize(&p, result, fmin, stats);
}

int oracle_opt_lbfgs_forces(int m, int n, const double* yTilde, const double* YTilde,
                            const double* forces0, const double* w0, double theta,
                            const oracle_lbfgs_config* cfg, double* result, double* fmin,
                            oracle_lbfgs_stats* stats) {
    model_t q = {m, n, yTilde, YTilde, w0, theta};
    problem_t p = {m, eval_forces, &q, cfg, 0};
    memcpy(result, forces0, sizeof(double) * (size_t)m);
    *fmin = 0.0;
    if (stats) stats->iterations = stats->evaluations = 0;
    return lbfgs_minimize(&p, result, fmin, stats);
}

/* ------------------------------------------------------------------ */
/* analytic self-test objectives                                       */
/* ------------------------------------------------------------------ */
typedef struct { int kind, n; } selftest_t;

static double eval_selftest(void* ctx, const double* x, double* g) {
    const selftest_t* q = (const selftest_t*)ctx;
    const int n = q->n;
    double f = 0.0;
    for (int i = 0; i < n; ++i) g[i] = 0.0;
    if (q->kind == 0) {
        for (int i = 0; i + 1 < n; i += 2) {
            const double t1 = 1.0 - x[i];
            const double t2 = 10.0 * (x[i + 1] - x[i] * x[i]);
            g[i + 1] = 20.0 * t2;
            g[i] = -2.0 * (x[i] * g[i + 1] + t1);
            f += t1 * t1 + t2 * t2;
        }
        if (n & 1) {
            f += x[n - 1] * x[n - 1];
            g[n - 1] = 2.0 * x[n - 1];
        }
    } else {
        for (int i = 0; i < n; ++i) {
            const double c = pow(10.0, 4.0 * i / (n > 1 ? n - 1 : 1) - 2.0);
            const double d = x[i] - 1.0;
            f += c * d * d + 0.01 * d * d * d * d;
            g[i] = 2.0 * c * d + 0.04 * d * d * d;
        }
    }
    return f;
}

int oracle_selftest_lbfgs(int kind, int n, const double* x0, const oracle_lbfgs_config* cfg,
                          double* x_out, double* fmin, oracle_lbfgs_stats* stats) {
    selftest_t q = {kind, n};
    problem_t p = {n, eval_selftest, &q, cfg, 0};
    memcpy(x_out, x0, sizeof(double) * (size_t)n);
    *fmin = 0.0;
    if (stats) stats->iterations = stats->evaluations = 0;
    return lbfgs_minimize(&p, x_out, fmin, stats);
}
