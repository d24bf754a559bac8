/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99 + OpenMP) of the BioEn log-weights / forces hot
 * path and of the liblbfgs driver loop that the reference runs it under.
 * It is the checker for the HIP product in bioen_amd/csrc; the product never
 * links, loads or calls anything declared here.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Parity status: PINNED -- validated here against oracle/_ref/libbioen_ref.so
 * (the reference's own C sources compiled by oracle/Makefile) and against the
 * reference's known answers test/optimize/data/\*.ref (tests/test_oracle.py).
 */
#ifndef BIOEN_ORACLE_H
#define BIOEN_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same fields, same order as the reference's lbfgs_config_params
 * (/root/reference/bioen/optimize/ext/c_bioen_common.h:69-79) */
typedef struct oracle_lbfgs_config {
    int linesearch;
    int max_iterations;
    double delta;
    double epsilon;
    double ftol;
    double gtol;
    double wolfe;
    int past;
    int max_linesearch;
} oracle_lbfgs_config;

/* counters the reference only prints (c_bioen_kernels_logw.c:515,569) */
typedef struct oracle_lbfgs_stats {
    int iterations;   /* progress-callback count == accepted line searches */
    int evaluations;  /* fdf calls */
} oracle_lbfgs_stats;

/* A1: w = softmax(g); returns log(sum_j exp(g_j))  (max-shifted). */
double oracle_logw_weights(const double* g, double* w, size_t n);

/* A4: 0.5 * || yTilde w - YTilde ||^2 ; optionally returns yTilde.w in yave[m]. */
double oracle_chi_squared(const double* w, const double* yTilde, const double* YTilde,
                          double* yave, size_t m, size_t n);

/* A1+A3+A4+A6 fused (what interface_lbfgs_logw evaluates). grad/w may be NULL. */
double oracle_logw_fdf(int m, int n, const double* yTilde, const double* YTilde,
                       const double* g, const double* G, double theta,
                       double* grad, double* w);

/* F1: weights from forces. */
void oracle_forces_weights(int m, int n, const double* yTilde, const double* forces,
                           const double* w0, double* w);

/* F1+F2+F3 fused (what interface_lbfgs_forces evaluates). grad/w may be NULL. */
double oracle_forces_fdf(int m, int n, const double* yTilde, const double* YTilde,
                         const double* forces, const double* w0, double theta,
                         double* grad, double* w);

/* A11/A12/A13/A14: L-BFGS drivers. Return the liblbfgs status code
 * (0,1,2 success; negative = error, same numbering as lbfgs.h:76-147). */
int oracle_opt_lbfgs_logw(int m, int n, const double* yTilde, const double* YTilde,
                          const double* g0, const double* G, double theta,
                          const oracle_lbfgs_config* cfg, double* result, double* fmin,
                          oracle_lbfgs_stats* stats);

int oracle_opt_lbfgs_forces(int m, int n, const double* yTilde, const double* YTilde,
                            const double* forces0, const double* w0, double theta,
                            const oracle_lbfgs_config* cfg, double* result, double* fmin,
                            oracle_lbfgs_stats* stats);

/* The oracle's L-BFGS on built-in analytic objectives (kind 0: extended Rosenbrock,
 * kind 1: ill-conditioned quadratic + quartic) -- the counterpart of
 * bioen_hip_selftest_lbfgs, used by the CPU tests to pin the product's driver. */
int oracle_selftest_lbfgs(int kind, int n, const double* x0, const oracle_lbfgs_config* cfg,
                          double* x_out, double* fmin, oracle_lbfgs_stats* stats);

/* ---- GSL multimin drivers (oracle/multimin_oracle.c) --------------------------------------
 * same fields, same order as the reference's gsl_config_params (c_bioen_common.h:62-67);
 * algorithm numbering of c_bioen_common.h:28-34 (0 conjugate_fr, 1 conjugate_pr, 2 vector_bfgs2,
 * 3 vector_bfgs, 4 steepest_descent). */
typedef struct oracle_gsl_config {
    double step_size;
    double tol;
    int max_iterations;
    int algorithm;
} oracle_gsl_config;

typedef struct oracle_gsl_stats {
    int iterations;
    int f_evaluations;
    int g_evaluations;
} oracle_gsl_stats;

/* The reference's _opt_bfgs_logw / _opt_bfgs_forces (c_bioen_kernels_logw.c:366-509): return the
 * GSL status (0 success, -2 GSL_CONTINUE = iteration budget used, 27 GSL_ENOPROG, 13 GSL_EBADTOL). */
int oracle_opt_gsl_logw(int m, int n, const double* yTilde, const double* YTilde, const double* g0,
                        const double* G, double theta, const oracle_gsl_config* cfg, double* result, double* fmin,
                        oracle_gsl_stats* stats);
int oracle_opt_gsl_forces(int m, int n, const double* yTilde, const double* YTilde, const double* forces0,
                          const double* w0, double theta, const oracle_gsl_config* cfg, double* result,
                          double* fmin, oracle_gsl_stats* stats);

/* The restated minimizers driven by the REFERENCE's own kernels (addresses taken from a build of the
 * reference; prototypes of c_bioen_kernels_logw.h:8-50, c_bioen_kernels_forces.h:10-56).  forces = 0:
 * x0 = log-weights (n), fixed = G; forces = 1: x0 = forces (m), fixed = w0.  yTildeT = NULL switches
 * the reference's transposed cache off. */
typedef struct oracle_ref_kernels {
    double (*get_weights)(double* g, double* w, size_t n);
    double (*logw_f)(double* g, double* G, double* yTilde, double* YTilde, double* w, double* t1, double theta,
                     int caching, double* yTildeT, double* tmp_n, double* tmp_m, int m, int n, double weights_sum);
    void (*logw_df)(double* g, double* G, double* yTilde, double* YTilde, double* w, double* gradient, double theta,
                    int caching, double* yTildeT, double* tmp_n, double* tmp_m, int m, int n, double weights_sum);
    void (*forces_weights)(double* w0, double* yTilde, double* forces, double* w, int caching, double* yTildeT,
                           double* tmp_n, size_t m, size_t n);
    double (*forces_f)(double* w0, double* yTilde, double* YTilde, double* w, double* t1, double theta, int caching,
                       double* yTildeT, double* tmp_n, double* tmp_m, int m, int n);
    void (*forces_df)(double* w0, double* yTilde, double* YTilde, double* w, double* gradient, double theta,
                      int caching, double* yTildeT, double* tmp_n, double* tmp_m, int m, int n);
} oracle_ref_kernels;
int oracle_opt_gsl_refobj(const oracle_ref_kernels* k, int forces, int m, int n, const double* yTilde,
                          const double* yTildeT, const double* YTilde, const double* x0, const double* fixed,
                          double theta, const oracle_gsl_config* cfg, double* result, double* fmin,
                          oracle_gsl_stats* stats);

void* oracle_refobj_create(const oracle_ref_kernels* k, int forces, int m, int n, const double* yTilde,
                           const double* yTildeT, const double* YTilde, const double* fixed, double theta);
void oracle_refobj_destroy(void* handle);
int oracle_refobj_eval(void* handle, const double* x, double* f, double* grad);   /* grad == NULL: f alone */

/* GSL's own multimin test programme (multimin/test.c, test_funcs.c): kind 0 Roth, 1 Wood,
 * 2 Rosenbrock, 3 SimpleAbs, run exactly as test_fdf does. */
int oracle_multimin_testfn_dim(int kind);
int oracle_selftest_multimin(int algorithm, int kind, const double* x0, double* x_out, double* fmin,
                             oracle_gsl_stats* stats);

#ifdef __cplusplus
}
#endif
#endif
