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

#ifdef __cplusplus
}
#endif
#endif
