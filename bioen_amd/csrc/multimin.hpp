// The five gradient minimizers the reference can run BioEn under through GSL
// (c_bioen_common.h:28-34; driver loop c_bioen_kernels_logw.c:366-509), written against an
// abstract vector backend so that the variables can stay where the objective lives:
//   * DeviceVectors (api.hip): N-vectors resident in HBM, level-1 algebra as HIP kernels, the
//     log-weights objective evaluated in place -- nothing but scalars crosses PCIe;
//   * HostVectors (below): the M-vector problems (forces method, analytic self tests).
//
// Algorithms followed (GSL 2.5, /root/reference/third-party/gsl-2.5/multimin/):
//   conjugate_fr.c, conjugate_pr.c, vector_bfgs.c  + directional_minimize.c
//   vector_bfgs2.c + linear_minimize.c (Fletcher) + linear_wrapper.c (the alpha cache)
//   steepest_descent.c
//
// Backend concept (handles are small integers):
//   void copy(dst, src); void zero(v); void axpy(a, x, y); void scal(a, x);
//   void step(x, p, coef, x1, dx);              dx = coef p ; x1 = x + dx
//   double dot(x, y); double nrm2(x); bool equal(x, y); double absmax(x);
//   void dots(k, xs, ys, out);                  k <= 4 inner products in one pass
//   double eval_f(x); void eval_df(x, g); void eval_fdf(x, &f, g);
//   bool failed();                              a device error: unwind
#pragma once

// host arithmetic of the minimizers: no fused multiply-add contraction, so that the scalar logic
// reproduces an ANSI build of GSL (and the oracle's restatement) operation for operation
#pragma clang fp contract(off)

#include <cfloat>
#include <cmath>
#include <cstddef>
#include <functional>
#include <vector>

namespace bioen {
namespace multimin {

enum Status : int { SUCCESS = 0, CONTINUE = -2, EBADTOL = 13, ENOPROG = 27, EBACKEND = -1 };   // gsl_errno.h
enum Algorithm : int { CONJUGATE_FR = 0, CONJUGATE_PR = 1, VECTOR_BFGS2 = 2, VECTOR_BFGS = 3, STEEPEST_DESCENT = 4 };

const char* status_string(int code);        // gsl_strerror's text for the codes above
const char* algorithm_name(int algorithm);

// work vectors of a minimizer
enum Slot : int { V_X = 0, V_GRAD, V_DX, V_X1, V_DX1, V_X2, V_P, V_G0, V_X0, V_DX0, V_DG0, V_XA, V_GA, V_COUNT };

// ---- one-dimensional helpers of Fletcher's line search (linear_minimize.c:10-131) -------------
int quadratic_roots(double a, double b, double c, double* r0, double* r1);   // poly/solve_quadratic.c
double interpolate(double a, double fa, double fpa, double b, double fb, double fpb, double xmin, double xmax,
                   int order);

template <class B>
class Minimizer {
public:
    Minimizer(B& backend, int algorithm) : b(backend), alg(algorithm) {}

    double f = 0.0;         // value at V_X
    int n_f = 0, n_g = 0;   // evaluation counts (an fdf counts in both)

    // gsl_multimin_fdfminimizer_set: V_X already holds the start point
    void set(double step_size, double tolerance) {
        iter = 0;
        step = step_size;
        tol = tolerance;
        b.zero(V_DX);
        fdf(V_X, &f, V_GRAD);
        if (alg == STEEPEST_DESCENT) return;
        if (alg == VECTOR_BFGS2) {                  // vector_bfgs2.c:140-186
            delta_f = 0.0;
            b.copy(V_X0, V_X);
            b.copy(V_G0, V_GRAD);
            g0norm = b.nrm2(V_G0);
            b.copy(V_P, V_GRAD);
            b.scal(-1.0 / g0norm, V_P);
            pnorm = b.nrm2(V_P);
            fp0 = -g0norm;
            f_alpha = f;
            restart_line();
            return;
        }
        if (alg == VECTOR_BFGS) b.copy(V_X0, V_X);  // vector_bfgs.c:155
        b.copy(V_P, V_GRAD);                        // the gradient is the first direction
        b.copy(V_G0, V_GRAD);
        pnorm = g0norm = b.nrm2(V_GRAD);
    }

    // gsl_multimin_fdfminimizer_iterate
    int iterate() {
        int st;
        if (alg == STEEPEST_DESCENT) st = iterate_steepest();
        else if (alg == VECTOR_BFGS2) st = iterate_bfgs2();
        else st = iterate_directional();
        return b.failed() ? (int)EBACKEND : st;
    }

private:
    B& b;
    const int alg;
    int iter = 0;
    double step = 0.0, tol = 0.0, pnorm = 0.0, g0norm = 0.0;
    // bfgs2
    double delta_f = 0.0, fp0 = 0.0;
    double f_alpha = 0.0, df_alpha = 0.0;
    double key_x = 0.0, key_f = 0.0, key_df = 0.0, key_g = 0.0;   // linear_wrapper.c's cache keys

    double fv(int x) { ++n_f; return b.eval_f(x); }
    void dfv(int x, int g) { ++n_g; b.eval_df(x, g); }
    void fdf(int x, double* fo, int g) { ++n_f; ++n_g; b.eval_fdf(x, fo, g); }

    // ---- steepest_descent.c:99-161 ----
    int iterate_steepest() {
        const double f0 = f;
        double f1 = 0.0, st = step;
        bool failed = false;
        const double gnorm = b.nrm2(V_GRAD);
        if (gnorm == 0.0) {
            b.zero(V_DX);
            return ENOPROG;
        }
        for (;;) {
            b.step(V_X, V_GRAD, -st / gnorm, V_X1, V_DX);
            if (b.equal(V_X, V_X1)) return ENOPROG;
            fdf(V_X1, &f1, V_DX1);              // V_DX1 plays g1
            if (b.failed()) return EBACKEND;
            if (f1 > f0) {                      // uphill: shrink and retry
                failed = true;
                st *= tol;
                continue;
            }
            break;
        }
        st *= failed ? tol : 2.0;
        step = st;
        b.copy(V_X, V_X1);
        b.copy(V_GRAD, V_DX1);
        f = f1;
        return SUCCESS;
    }

    // ---- directional_minimize.c:32-87 ----
    void bracket(double lambda, double pg, double stepc, double fa, double fc, double* stepb_out, double* fb_out) {
        for (;;) {
            const double u = std::fabs(pg * lambda * stepc);
            const double stepb = 0.5 * stepc * u / ((fc - fa) + u);
            b.step(V_X, V_P, -stepb * lambda, V_X1, V_DX1);
            if (b.equal(V_X, V_X1)) {
                *stepb_out = 0.0;
                *fb_out = fa;
                dfv(V_X1, V_GRAD);
                return;
            }
            const double fb = fv(V_X1);
            if (b.failed()) { *stepb_out = 0.0; *fb_out = fa; return; }
            if (fb >= fa && stepb > 0.0) {
                fc = fb;
                stepc = stepb;
                continue;
            }
            *stepb_out = stepb;
            *fb_out = fb;
            dfv(V_X1, V_GRAD);
            return;
        }
    }

    // ---- directional_minimize.c:89-248: parabolic / golden-section refinement, <= 10 trials ----
    void refine(double lambda, double stepa, double stepb, double stepc, double fa, double fb, double fc,
                double* gnorm_out) {
        double u = stepb, v = stepa, w = stepc;
        double fu = fb, fvv = fa, fw = fc;
        double old2 = std::fabs(w - v), old1 = std::fabs(v - u);
        b.copy(V_X2, V_X1);
        b.copy(V_DX, V_DX1);
        f = fb;
        step = stepb;
        *gnorm_out = b.nrm2(V_GRAD);
        for (int trial = 1; trial <= 10; ++trial) {
            const double dw = w - u, dv = v - u;
            double du = 0.0, stepm;
            const double e1 = ((fvv - fu) * dw * dw + (fu - fw) * dv * dv);
            const double e2 = 2.0 * ((fvv - fu) * dw + (fu - fw) * dv);
            if (e2 != 0.0) du = e1 / e2;
            if (du > 0.0 && du < (stepc - stepb) && std::fabs(du) < 0.5 * old2) stepm = u + du;
            else if (du < 0.0 && du > (stepa - stepb) && std::fabs(du) < 0.5 * old2) stepm = u + du;
            else if ((stepc - stepb) > (stepb - stepa)) stepm = 0.38 * (stepc - stepb) + stepb;
            else stepm = stepb - 0.38 * (stepb - stepa);

            b.step(V_X, V_P, -stepm * lambda, V_X1, V_DX1);
            const double fm = fv(V_X1);
            if (b.failed()) return;
            if (fm > fb) {
                if (fm < fvv) { w = v; v = stepm; fw = fvv; fvv = fm; }
                else if (fm < fw) { w = stepm; fw = fm; }
                if (stepm < stepb) { stepa = stepm; fa = fm; }
                else { stepc = stepm; fc = fm; }
                continue;
            }
            if (!(fm <= fb)) return;            // NaN: GSL falls off the end of the routine
            old2 = old1;
            old1 = std::fabs(u - stepm);
            w = v; v = u; u = stepm;
            fw = fvv; fvv = fu; fu = fm;
            b.copy(V_X2, V_X1);
            b.copy(V_DX, V_DX1);
            dfv(V_X1, V_GRAD);
            const double pg = b.dot(V_P, V_GRAD);
            const double gnorm1 = b.nrm2(V_GRAD);
            f = fm;
            step = stepm;
            *gnorm_out = gnorm1;
            if (std::fabs(pg * lambda / gnorm1) < tol) return;
            if (stepm < stepb) { stepc = stepb; fc = fb; stepb = stepm; fb = fm; }
            else { stepa = stepb; fa = fb; stepb = stepm; fb = fm; }
        }
    }

    // ---- conjugate_fr.c:145-250, conjugate_pr.c:149-262, vector_bfgs.c:186-340 ----
    int iterate_directional() {
        const double fa = f, stepc = step;
        if (pnorm == 0.0 || g0norm == 0.0) {
            b.zero(V_DX);
            return ENOPROG;
        }
        const double pg = b.dot(V_P, V_GRAD);
        const double lambda = ((pg >= 0.0) ? +1.0 : -1.0) / pnorm;
        b.step(V_X, V_P, -stepc * lambda, V_X1, V_DX);
        const double fc = fv(V_X1);
        if (b.failed()) return EBACKEND;
        if (fc < fa) {                          // downhill already: take it and double the step
            step = stepc * 2.0;
            f = fc;
            b.copy(V_X, V_X1);
            dfv(V_X1, V_GRAD);
            return SUCCESS;
        }
        double stepb, fb, g1norm = 0.0;
        bracket(lambda, pg, stepc, fa, fc, &stepb, &fb);
        if (b.failed()) return EBACKEND;
        if (stepb == 0.0) return ENOPROG;
        refine(lambda, 0.0, stepb, stepc, fa, fb, fc, &g1norm);
        if (b.failed()) return EBACKEND;
        b.copy(V_X, V_X2);

        iter = (iter + 1) % b.size();
        if (iter == 0) {                        // periodic restart along the gradient
            b.copy(V_P, V_GRAD);
            pnorm = g1norm;
        } else if (alg == CONJUGATE_FR) {
            const double beta = -std::pow(g1norm / g0norm, 2.0);
            b.scal(-beta, V_P);
            b.axpy(1.0, V_GRAD, V_P);
            pnorm = b.nrm2(V_P);
        } else if (alg == CONJUGATE_PR) {
            b.axpy(-1.0, V_GRAD, V_G0);                     // g0 - g1
            const double beta = b.dot(V_G0, V_GRAD) / (g0norm * g0norm);
            b.scal(-beta, V_P);
            b.axpy(1.0, V_GRAD, V_P);
            pnorm = b.nrm2(V_P);
        } else {
            bfgs_direction();
            pnorm = b.nrm2(V_P);
        }
        if (alg == VECTOR_BFGS) {
            b.copy(V_G0, V_GRAD);
            b.copy(V_X0, V_X);
            g0norm = b.nrm2(V_G0);
        } else {
            g0norm = g1norm;
            b.copy(V_G0, V_GRAD);
        }
        return SUCCESS;
    }

    // p' = g1 - A dx - B dg,  B = dx.g / dx.dg,  A = -(1 + dg.dg/dx.dg) B + dg.g/dx.dg   (vector_bfgs.c:292-327)
    void bfgs_direction() {
        b.copy(V_DX0, V_X);
        b.axpy(-1.0, V_X0, V_DX0);
        b.copy(V_DG0, V_GRAD);
        b.axpy(-1.0, V_G0, V_DG0);
        const int xs[3] = {V_DX0, V_DG0, V_DX0}, ys[3] = {V_GRAD, V_GRAD, V_DG0};
        double o[3];
        b.dots(3, xs, ys, o);
        const double dxg = o[0], dgg = o[1], dxdg = o[2], dgnorm = b.nrm2(V_DG0);
        double A = 0.0, Bc = 0.0;
        if (dxdg != 0) {
            Bc = dxg / dxdg;
            A = -(1.0 + dgnorm * dgnorm / dxdg) * Bc + dgg / dxdg;
        }
        b.copy(V_P, V_GRAD);
        b.axpy(-A, V_DX0, V_P);
        b.axpy(-Bc, V_DG0, V_P);
    }

    // ---- linear_wrapper.c: f and f' along x0 + alpha p with the four cache keys ----
    void move_to(double alpha) {
        if (alpha == key_x) return;
        b.copy(V_XA, V_X0);
        b.axpy(alpha, V_P, V_XA);
        key_x = alpha;
    }
    double line_f(double alpha) {
        if (alpha == key_f) return f_alpha;
        move_to(alpha);
        f_alpha = fv(V_XA);
        key_f = alpha;
        return f_alpha;
    }
    double line_df(double alpha) {
        if (alpha == key_df) return df_alpha;
        move_to(alpha);
        if (alpha != key_g) {
            dfv(V_XA, V_GA);
            key_g = alpha;
        }
        df_alpha = b.dot(V_GA, V_P);
        key_df = alpha;
        return df_alpha;
    }
    void line_fdf(double alpha, double* fo, double* dfo) {
        if (alpha == key_f && alpha == key_df) { *fo = f_alpha; *dfo = df_alpha; return; }
        if (alpha == key_f || alpha == key_df) { *fo = line_f(alpha); *dfo = line_df(alpha); return; }
        move_to(alpha);
        fdf(V_XA, &f_alpha, V_GA);
        key_f = alpha;
        key_g = alpha;
        df_alpha = b.dot(V_GA, V_P);
        key_df = alpha;
        *fo = f_alpha;
        *dfo = df_alpha;
    }
    void restart_line() {                       // prepare_wrapper / change_direction
        b.copy(V_XA, V_X0);
        key_x = 0.0;
        key_f = 0.0;
        b.copy(V_GA, V_G0);
        key_g = 0.0;
        df_alpha = b.dot(V_GA, V_P);
        key_df = 0.0;
    }

    // ---- linear_minimize.c:136-247 ----
    int fletcher(double alpha1, double* alpha_new) {
        const double rho = 0.01, sigma = tol, tau1 = 9, tau2 = 0.05, tau3 = 0.5;   // vector_bfgs2.c:177-182
        const int order = 3;
        double f0, fp0_, falpha, falpha_prev, fpalpha, fpalpha_prev, delta, alpha_next;
        double alpha = alpha1, alpha_prev = 0.0;
        double a = 0.0, bb = alpha, fa, fb = 0.0, fpa, fpb = 0.0;
        const size_t bracket_iters = 100, section_iters = 100;
        size_t i = 0;
        line_fdf(0.0, &f0, &fp0_);
        falpha_prev = f0;
        fpalpha_prev = fp0_;
        fa = f0;
        fpa = fp0_;
        while (i++ < bracket_iters) {
            falpha = line_f(alpha);
            if (b.failed()) return EBACKEND;
            if (falpha > f0 + alpha * rho * fp0_ || falpha >= falpha_prev) {
                a = alpha_prev; fa = falpha_prev; fpa = fpalpha_prev;
                bb = alpha; fb = falpha; fpb = NAN;
                break;
            }
            fpalpha = line_df(alpha);
            if (b.failed()) return EBACKEND;
            if (std::fabs(fpalpha) <= -sigma * fp0_) {
                *alpha_new = alpha;
                return SUCCESS;
            }
            if (fpalpha >= 0) {
                a = alpha; fa = falpha; fpa = fpalpha;
                bb = alpha_prev; fb = falpha_prev; fpb = fpalpha_prev;
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
            delta = bb - a;
            alpha = interpolate(a, fa, fpa, bb, fb, fpb, a + tau2 * delta, bb - tau3 * delta, order);
            falpha = line_f(alpha);
            if (b.failed()) return EBACKEND;
            if ((a - alpha) * fpa <= DBL_EPSILON) return ENOPROG;
            if (falpha > f0 + rho * alpha * fp0_ || falpha >= fa) {
                bb = alpha; fb = falpha; fpb = NAN;
            } else {
                fpalpha = line_df(alpha);
                if (b.failed()) return EBACKEND;
                if (std::fabs(fpalpha) <= -sigma * fp0_) {
                    *alpha_new = alpha;
                    return SUCCESS;
                }
                if (((bb - a) >= 0 && fpalpha >= 0) || ((bb - a) <= 0 && fpalpha <= 0)) {
                    bb = a; fb = fa; fpb = fpa;
                    a = alpha; fa = falpha; fpa = fpalpha;
                } else {
                    a = alpha; fa = falpha; fpa = fpalpha;
                }
            }
        }
        return SUCCESS;
    }

    // ---- vector_bfgs2.c:208-317 ----
    int iterate_bfgs2() {
        double alpha = 0.0, alpha1;
        const double f0 = f;
        if (pnorm == 0.0 || g0norm == 0.0 || fp0 == 0) {
            b.zero(V_DX);
            return ENOPROG;
        }
        if (delta_f < 0) {
            const double del = std::fmax(-delta_f, 10 * DBL_EPSILON * std::fabs(f0));
            alpha1 = std::fmin(1.0, 2.0 * del / (-fp0));
        } else {
            alpha1 = std::fabs(step);
        }
        const int st = fletcher(alpha1, &alpha);
        if (st != SUCCESS) return st;
        double fo, dfo;
        line_fdf(alpha, &fo, &dfo);                 // update_position: make sure all is cached
        if (b.failed()) return EBACKEND;
        f = f_alpha;
        b.copy(V_X, V_XA);
        b.copy(V_GRAD, V_GA);
        delta_f = f - f0;
        bfgs_direction();
        b.copy(V_DX, V_DX0);
        b.copy(V_G0, V_GRAD);
        b.copy(V_X0, V_X);
        g0norm = b.nrm2(V_G0);
        pnorm = b.nrm2(V_P);
        const double dir = (b.dot(V_P, V_GRAD) >= 0.0) ? -1.0 : +1.0;
        b.scal(dir / pnorm, V_P);
        pnorm = b.nrm2(V_P);
        fp0 = b.dot(V_P, V_G0);
        restart_line();
        return SUCCESS;
    }
};

struct Config {             // == the reference's gsl_config_params
    double step_size;
    double tol;
    int max_iterations;
    int algorithm;
};

struct Outcome {
    int status = SUCCESS;
    int iterations = 0;
    int f_evaluations = 0, g_evaluations = 0;
    double fmin = 0.0;
};

// The reference's driver (c_bioen_kernels_logw.c:434-464): iterate, stop when the max-norm of the
// gradient drops below tol (c_bioen_common.c:112-138) or the budget is used (status CONTINUE).
// The start point must be in V_X; the result is left there.
template <class B>
Outcome run(B& backend, const Config& cfg) {
    Outcome out;
    Minimizer<B> s(backend, cfg.algorithm);
    s.set(cfg.step_size, cfg.tol);
    if (backend.failed()) { out.status = EBACKEND; return out; }
    int iter = 0, status;
    do {
        status = s.iterate();
        if (status) break;
        status = cfg.tol < 0.0 ? (int)EBADTOL : (backend.absmax(V_GRAD) < cfg.tol ? (int)SUCCESS : (int)CONTINUE);
        if (backend.failed()) { status = EBACKEND; break; }
        ++iter;
    } while (status == CONTINUE && iter < cfg.max_iterations);
    out.status = status;
    out.iterations = iter;
    out.f_evaluations = s.n_f;
    out.g_evaluations = s.n_g;
    out.fmin = s.f;
    return out;
}

// GSL's own test protocol (multimin/test.c:106-160): step 0.1 |x0|, tol 0.1, |g|_2 < 1e-3, <= 5000
template <class B>
Outcome run_gsl_test(B& backend, int algorithm) {
    Outcome out;
    Minimizer<B> s(backend, algorithm);
    s.set(0.1 * backend.nrm2(V_X), 0.1);
    int iter = 0, status;
    do {
        ++iter;
        status = s.iterate();
        if (status == ENOPROG || status == EBACKEND) break;
        status = backend.nrm2(V_GRAD) < 1e-3 ? (int)SUCCESS : (int)CONTINUE;
    } while (iter < 5000 && status == CONTINUE);
    out.status = status;
    out.iterations = iter;
    out.f_evaluations = s.n_f;
    out.g_evaluations = s.n_g;
    out.fmin = s.f;
    return out;
}

// ---- host-resident backend -------------------------------------------------------------------
class HostVectors {
public:
    using Fdf = std::function<int(const double* x, double* f, double* grad)>;   // grad may be NULL; rc != 0: error
    HostVectors(int n, Fdf fn) : n_(n), fn_(std::move(fn)), v_(V_COUNT, std::vector<double>((size_t)n, 0.0)) {}
    int size() const { return n_; }
    double* data(int v) { return v_[v].data(); }
    bool failed() const { return rc_ != 0; }
    int error() const { return rc_; }

    void copy(int dst, int src) { v_[dst] = v_[src]; }
    void zero(int v) { std::fill(v_[v].begin(), v_[v].end(), 0.0); }
    void axpy(double a, int x, int y) {
        if (a == 0.0) return;
        const double* xs = v_[x].data();
        double* ys = v_[y].data();
        for (int i = 0; i < n_; ++i) ys[i] += a * xs[i];
    }
    void scal(double a, int x) { for (double& e : v_[x]) e *= a; }
    void step(int x, int p, double coef, int x1, int dx) {
        for (int i = 0; i < n_; ++i) {
            const double d = coef * v_[p][i];
            v_[dx][i] = d;
            v_[x1][i] = v_[x][i] + d;
        }
    }
    double dot(int x, int y) const {
        double s = 0.0;
        for (int i = 0; i < n_; ++i) s += v_[x][i] * v_[y][i];
        return s;
    }
    void dots(int k, const int* xs, const int* ys, double* out) const { for (int q = 0; q < k; ++q) out[q] = dot(xs[q], ys[q]); }
    double nrm2(int x) const {                  // cblas_dnrm2's scaled sum of squares (source_nrm2_r.h)
        if (n_ == 1) return std::fabs(v_[x][0]);
        double scale = 0.0, ssq = 1.0;
        for (double e : v_[x]) {
            if (e == 0.0) continue;
            const double ax = std::fabs(e);
            if (scale < ax) {
                ssq = 1.0 + ssq * (scale / ax) * (scale / ax);
                scale = ax;
            } else {
                ssq += (ax / scale) * (ax / scale);
            }
        }
        return scale * std::sqrt(ssq);
    }
    bool equal(int x, int y) const { return v_[x] == v_[y]; }
    double absmax(int x) const {
        double r = 0.0;
        for (double e : v_[x]) r = std::fmax(r, std::fabs(e));
        return r;
    }
    double eval_f(int x) {
        double f = 0.0;
        if (!rc_) rc_ = fn_(v_[x].data(), &f, nullptr);
        return f;
    }
    void eval_df(int x, int g) {
        double f;
        if (!rc_) rc_ = fn_(v_[x].data(), &f, v_[g].data());
    }
    void eval_fdf(int x, double* f, int g) {
        *f = 0.0;
        if (!rc_) rc_ = fn_(v_[x].data(), f, v_[g].data());
    }

private:
    int n_;
    Fdf fn_;
    std::vector<std::vector<double>> v_;
    int rc_ = 0;
};

// analytic objectives of GSL's multimin test programme (test_funcs.c): 0 Roth, 1 Wood,
// 2 Rosenbrock (GSL's scaling), 3 SimpleAbs
int test_function_dim(int kind);
void test_function(int kind, const double* x, double* f, double* grad);

}  // namespace multimin
}  // namespace bioen
