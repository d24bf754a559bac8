/* The C ABI of include/bioen_hip.h used from plain C99 -- no Python, no C++: what a maintainer of another host language
 * binds.  Replaces, call for call, what bioen/optimize/ext/c_bioen.pyx does around the reference's C entry points
 * (c_bioen.pyx:441-520 -> _opt_lbfgs_logw, c_bioen_kernels_logw.c:581-669; :719-792 -> _opt_lbfgs_forces).
 *
 *   gcc -std=c99 -Wall -Wextra -pedantic -I include examples/c_abi_demo.c -L bioen_amd -lbioen_hip \
 *       -Wl,-rpath,$PWD/bioen_amd -Wl,-rpath,/opt/rocm/lib -lm -o build/c_abi_demo && build/c_abi_demo
 *
 * Exit status: 0 = ran and the checks held, 77 = no HIP device visible (the library has no CPU path), 1 = a check failed. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "bioen_hip.h"

#define CHECK(call)                                                                                     \
    do {                                                                                                \
        int rc_ = (call);                                                                               \
        if (rc_ != BIOEN_HIP_OK) {                                                                      \
            fprintf(stderr, "%s -> %s: %s\n", #call, bioen_hip_strerror(rc_), bioen_hip_last_error()); \
            return 1;                                                                                   \
        }                                                                                               \
    } while (0)

static double unit(unsigned long long* s) {     /* splitmix64 -> (0, 1) */
    unsigned long long z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return ((double)(z >> 11) + 0.5) / 9007199254740992.0;
}

int main(void) {
    int ndev = 0;
    printf("%s\n", bioen_hip_version());
    if (bioen_hip_device_count(&ndev) != BIOEN_HIP_OK || ndev < 1) {
        printf("no HIP device visible: nothing to run (libbioen_hip has no CPU path)\n");
        return 77;
    }
    enum { M = 24, N = 1500 };
    static double yTilde[M * N], YTilde[M], g0[N], G[N], gopt[N], w[N], grad[N], f0[M], fopt[M], w0[N];
    unsigned long long seed = 12345;
    for (int i = 0; i < M; ++i) {
        const double ytrue = 1.0 + 9.0 * unit(&seed);
        YTilde[i] = 10.0 + 0.3 * (unit(&seed) - 0.5);
        for (int j = 0; j < N; ++j) yTilde[i * N + j] = (ytrue + 0.5 * ytrue * 3.4 * (unit(&seed) - 0.5)) / (0.1 * ytrue);
    }
    for (int j = 0; j < N; ++j) { G[j] = 0.0; g0[j] = 0.0; w0[j] = 1.0 / N; }
    for (int i = 0; i < M; ++i) f0[i] = 0.0;

    bioen_hip_ctx* ctx = NULL;
    CHECK(bioen_hip_ctx_create(M, N, yTilde, YTilde, 0, &ctx));          /* yTilde goes to HBM once */

    const double theta = 10.0;
    double f = 0.0;
    CHECK(bioen_hip_logw_fdf(ctx, g0, G, theta, &f, grad));               /* interface_lbfgs_logw */
    /* uniform weights: f = 0.5 |yTilde w - YTilde|^2, checked here on the host */
    double chi = 0.0;
    for (int i = 0; i < M; ++i) {
        double yb = 0.0;
        for (int j = 0; j < N; ++j) yb += yTilde[i * N + j] / N;
        chi += 0.5 * (yb - YTilde[i]) * (yb - YTilde[i]);
    }
    printf("f(g0) = %.12g (host closed form %.12g)\n", f, chi);
    if (fabs(f - chi) > 1e-10 * chi) return 1;

    bioen_lbfgs_config cfg = {2, 5000, 1e-6, 1e-6, 1e-5, 0.9, 0.9, 10, 100};   /* bioen_optimize.yaml:33-46 */
    bioen_visual_params vis = {0, 0};
    bioen_opt_result info;
    CHECK(bioen_hip_opt_lbfgs_logw(ctx, g0, G, theta, &cfg, &vis, gopt, w, &info));   /* _opt_lbfgs_logw */
    double sw = 0.0;
    for (int j = 0; j < N; ++j) sw += w[j];
    printf("log-weights: fmin = %.10g after %d iterations / %d evaluations, liblbfgs status %d (%s), sum w = %.15g\n",
           info.fmin, info.iterations, info.evaluations, info.lbfgs_code, bioen_hip_lbfgs_strerror(info.lbfgs_code), sw);
    if (!(info.lbfgs_code >= 0 && info.lbfgs_code <= 2) || !(info.fmin < f) || fabs(sw - 1.0) > 1e-12) return 1;
    if (fabs(info.fmin - (theta * info.kl + info.chi2)) > 1e-10 * fabs(info.fmin)) return 1;

    bioen_opt_result finfo;
    CHECK(bioen_hip_opt_lbfgs_forces(ctx, f0, w0, theta, &cfg, &vis, fopt, w, &finfo));  /* _opt_lbfgs_forces */
    printf("forces:      fmin = %.10g after %d iterations, status %d\n", finfo.fmin, finfo.iterations, finfo.lbfgs_code);
    /* both methods minimise the same posterior (uniform prior): the minima agree to the stopping tolerance */
    if (fabs(finfo.fmin - info.fmin) > 1e-4 * fabs(info.fmin)) return 1;

    CHECK(bioen_hip_ctx_destroy(ctx));
    printf("ok\n");
    return 0;
}
