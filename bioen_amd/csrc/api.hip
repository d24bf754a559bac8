// extern "C" entry points of libbioen_hip.so (declared in include/bioen_hip.h) and the
// two L-BFGS backends that sit on the kernels of kernels.hip.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "ctx.hpp"
#include "kernels.hpp"
#include "lbfgs.hpp"

static_assert(bioen::kHistory == bioen::kLbfgsM, "history length");

namespace bioen {

static thread_local std::string g_last_error;
static int g_fast_openmp_flag = 0;

void set_last_error(const std::string& s) { g_last_error = s; }

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_last_error = buf;
    return BIOEN_HIP_EHIP;
}

static int fail(int code, const char* msg) {
    g_last_error = msg;
    return code;
}

// ---------------------------------------------------------------------------------
// allocation helpers
// ---------------------------------------------------------------------------------
static int dalloc_zero(double** p, size_t count, hipStream_t s) {
    *p = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(double));
    if (e != hipSuccess) {
        hip_fail(e, "hipMalloc", __FILE__, __LINE__);
        return BIOEN_HIP_ENOMEM;
    }
    BIOEN_HIP_CHECK(hipMemsetAsync(*p, 0, count * sizeof(double), s));
    return 0;
}

static void choose_fwd_tiling(bioen_hip_ctx* c) {
    const int total_steps = (int)(c->ld / 128);
    const int row_blocks = c->mp / kRowAlign;
    int want_tiles = (6144 + row_blocks - 1) / row_blocks;
    want_tiles = std::max(1, std::min(want_tiles, total_steps));
    int spt = (total_steps + want_tiles - 1) / want_tiles;
    if (spt & 1) ++spt;   // two 1-KiB steps in flight per row
    c->fwd_steps = spt;
    c->fwd_ctiles = (total_steps + spt - 1) / spt;
}

static int ctx_alloc(int m, int n, int device, bioen_hip_ctx** out) {
    if (!out) return fail(BIOEN_HIP_EINVAL, "ctx pointer is NULL");
    *out = nullptr;
    if (m <= 0 || n <= 0) return fail(BIOEN_HIP_EINVAL, "m and n must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(BIOEN_HIP_ENODEV, "no HIP device visible (libbioen_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(BIOEN_HIP_EINVAL, "device index out of range");
    BIOEN_HIP_CHECK(hipSetDevice(device));

    bioen_hip_ctx* c = new (std::nothrow) bioen_hip_ctx();
    if (!c) return fail(BIOEN_HIP_ENOMEM, "host allocation failed");
    c->device = device;
    c->m = m;
    c->n = n;
    c->mp = (int)round_up((size_t)m, kRowAlign);
    c->ld = round_up((size_t)n, kColAlign);
    choose_fwd_tiling(c);
    // stream yTilde with non-temporal loads once it no longer fits the 256 MiB Infinity Cache
    c->nontemporal = (size_t)c->mp * c->ld * sizeof(double) > (size_t)192 * 1024 * 1024;

    int rc = 0;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__);
    }
#define TRY(x) if ((rc = (x)) != 0) { bioen_hip_ctx_destroy(c); return rc; }
    TRY(dalloc_zero(&c->Y, (size_t)c->mp * c->ld, c->stream));
    TRY(dalloc_zero(&c->YT, c->mp, c->stream));
    TRY(dalloc_zero(&c->ybar, c->mp, c->stream));
    TRY(dalloc_zero(&c->r, c->mp, c->stream));
    TRY(dalloc_zero(&c->um, c->mp, c->stream));
    TRY(dalloc_zero(&c->gm, c->mp, c->stream));
    double** nvecs[] = {&c->x, &c->xp, &c->g, &c->gp, &c->d, &c->w, &c->fixed, &c->a, &c->t};
    for (double** p : nvecs) TRY(dalloc_zero(p, c->ld, c->stream));
    TRY(dalloc_zero(&c->fwd_partial, (size_t)c->mp * c->fwd_ctiles, c->stream));
    TRY(dalloc_zero(&c->part, (size_t)P_COUNT * kMaxPartials, c->stream));
    TRY(dalloc_zero(&c->scal, 64, c->stream));
#undef TRY
    e = hipHostMalloc(reinterpret_cast<void**>(&c->host_scal), 64 * sizeof(double), hipHostMallocDefault);
    if (e != hipSuccess) {
        bioen_hip_ctx_destroy(c);
        return hip_fail(e, "hipHostMalloc", __FILE__, __LINE__);
    }
    *out = c;
    return 0;
}

static int ensure_history(bioen_hip_ctx* c) {
    if (c->history_allocated) return 0;
    for (int i = 0; i < kHistory; ++i) {
        int rc = dalloc_zero(&c->S[i], c->ld, c->stream);
        if (rc) return rc;
        rc = dalloc_zero(&c->Yh[i], c->ld, c->stream);
        if (rc) return rc;
    }
    c->history_allocated = true;
    return 0;
}

static int upload_n(bioen_hip_ctx* c, double* dst, const double* src) {
    BIOEN_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)c->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return 0;
}

static int read_scalars(bioen_hip_ctx* c) {
    BIOEN_HIP_CHECK(hipMemcpyAsync(c->host_scal, c->scal, S_COUNT * sizeof(double), hipMemcpyDeviceToHost,
                                   c->stream));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

static int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch", __FILE__, __LINE__);
    return 0;
}

// ---------------------------------------------------------------------------------
// evaluation pipelines (all asynchronous on c->stream)
// ---------------------------------------------------------------------------------
// log-weights: x must hold the point and P_MAX its block maxima (launch_trial does both).
static void enqueue_logw_eval(bioen_hip_ctx* c, double theta, bool with_grad) {
    launch_logw_exp(c);                 // A1 first half + prior partials
    launch_logw_norm(c);                // A1 second half -> w, log s, P
    launch_fwd_partial(c, c->w);        // A4: ybar = yTilde . w            [matrix pass 1]
    launch_fwd_rows_residual(c);        //     r, chi^2, ybar . r
    launch_logw_scalars(c, theta);      // A5: f
    if (with_grad) {
        launch_adj(c, c->r, c->a, true); // A6: a_k = sum_i r_i (yTilde_ik - ybar_i) [matrix pass 2]
        launch_logw_grad(c, theta);     //     gradient epilogue + g.d, g.g, x.x
        launch_finish_eval(c);
    }
}

// forces: um holds the forces
static void enqueue_forces_weights(bioen_hip_ctx* c) {
    launch_adj(c, c->um, c->a);         // F1: x_j = sum_i f_i yTilde_ij     [matrix pass 1]
    launch_max(c, c->a);
    launch_forces_exp(c, c->a);
    launch_forces_norm(c);              // w ; KL partials
}

static void enqueue_forces_eval(bioen_hip_ctx* c, double theta, bool with_grad) {
    enqueue_forces_weights(c);
    launch_fwd_partial(c, c->w);        // F2: ybar                         [matrix pass 2]
    launch_fwd_rows_residual(c);
    launch_forces_scalars(c, theta);    //     f = theta KL + 0.5 chi^2
    if (with_grad) {
        launch_adj(c, c->r, c->a);      // F3: b = yTilde^T r                [matrix pass 3]
        launch_forces_t(c, theta);      //     t_j, sum t
        launch_fwd_partial(c, c->t, true); //  gm_i = sum_j (yTilde_ij - ybar_i) t_j [matrix pass 4]
        launch_fwd_rows_forces_grad(c);
    }
}

// ---------------------------------------------------------------------------------
// L-BFGS backends
// ---------------------------------------------------------------------------------
struct DeviceLogwBackend {
    bioen_hip_ctx* c;
    double theta;
    bool result_is_trial = false;
    int rc = 0;   // first HIP failure, if any

    int fetch(TrialResult* t) {
        int e = read_scalars(c);
        if (e && !rc) rc = e;
        const double* h = c->host_scal;
        t->f = h[S_F];
        t->dg = h[S_DG];
        t->gg = h[S_GG];
        t->xx = h[S_XX];
        t->dginit = h[S_DGINIT];
        return e;
    }

    void direction(int end_after, int bound, bool finalize_pair, int newest) {
        // lbfgs.c:571-598 as 1 + 2*bound fused launches
        RecurArgs a{};
        a.mode = 0;
        a.hist = newest;
        a.finalize_sy = finalize_pair ? 1 : 0;
        if (bound == 0) {
            a.vdot = c->gp;
            a.out_slot = P_DGINIT;
            launch_recur(c, a);
            return;
        }
        int j = end_after;
        int order[kHistory];
        for (int b = 0; b < bound; ++b) {
            j = (j + kHistory - 1) % kHistory;
            order[b] = j;   // newest -> oldest
        }
        a.vdot = c->S[order[0]];
        a.out_slot = P_REC;
        launch_recur(c, a);
        for (int b = 0; b < bound; ++b) {   // first loop
            RecurArgs s{};
            s.mode = 1;
            s.hist = order[b];
            s.vaxpy = c->Yh[order[b]];
            const bool last = (b == bound - 1);
            s.scale = last ? 1 : 0;
            s.vdot = last ? c->Yh[order[b]] : c->S[order[b + 1]];
            s.out_slot = P_REC;
            launch_recur(c, s);
        }
        for (int b = bound - 1; b >= 0; --b) {   // second loop, oldest -> newest
            RecurArgs s{};
            s.mode = 2;
            s.hist = order[b];
            s.vaxpy = c->S[order[b]];
            const bool last = (b == 0);
            s.vdot = last ? c->gp : c->Yh[order[b - 1]];
            s.out_slot = last ? P_DGINIT : P_REC;
            launch_recur(c, s);
        }
    }

    void initial(double* f, double* gg, double* xx) {
        // x0 is in xp; d is zero.  trial(0) evaluates there.
        launch_trial(c, 0.0);
        enqueue_logw_eval(c, theta, true);
        TrialResult t{};
        fetch(&t);
        *f = t.f;
        *gg = t.gg;
        *xx = t.xx;
        std::swap(c->g, c->gp);   // gradient at the accepted point
        direction(0, 0, false, 0);
    }

    void trial(double stp, TrialResult* t) {
        launch_trial(c, stp);
        enqueue_logw_eval(c, theta, true);
        fetch(t);
    }

    void accept(int end, int bound) {
        launch_update_sy(c, c->S[end], c->Yh[end]);
        std::swap(c->x, c->xp);
        std::swap(c->g, c->gp);
        direction((end + 1) % kHistory, bound, true, end);
    }

    void revert() { result_is_trial = false; }
    void keep_trial() { result_is_trial = true; }
};

// forces: M variables live on the host, evaluations on the device
struct HostForcesBackend {
    bioen_hip_ctx* c;
    double theta;
    int m;
    std::vector<double> x, xp, g, gp, d;
    std::vector<double> S[kHistory], Y[kHistory];
    double ys[kHistory] = {}, alpha[kHistory] = {};
    bool result_is_trial = false;
    int rc = 0;

    HostForcesBackend(bioen_hip_ctx* ctx, double th, const double* x0)
        : c(ctx), theta(th), m(ctx->m), x(m), xp(x0, x0 + m), g(m), gp(m), d(m) {
        for (int i = 0; i < kHistory; ++i) {
            S[i].assign(m, 0.0);
            Y[i].assign(m, 0.0);
        }
    }

    static double dot(const std::vector<double>& a, const std::vector<double>& b) {
        double s = 0.0;
        for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
        return s;
    }

    double evaluate(const std::vector<double>& at, std::vector<double>& grad) {
        hipError_t e = hipMemcpyAsync(c->um, at.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice,
                                      c->stream);
        if (e != hipSuccess && !rc) rc = hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__);
        enqueue_forces_eval(c, theta, true);
        e = hipMemcpyAsync(grad.data(), c->gm, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess && !rc) rc = hip_fail(e, "hipMemcpyAsync", __FILE__, __LINE__);
        int r2 = read_scalars(c);
        if (r2 && !rc) rc = r2;
        return c->host_scal[S_F];
    }

    void initial(double* f, double* gg, double* xx) {
        *f = evaluate(xp, gp);
        *gg = dot(gp, gp);
        *xx = dot(xp, xp);
        for (int i = 0; i < m; ++i) d[i] = -gp[i];
    }

    void trial(double stp, TrialResult* t) {
        for (int i = 0; i < m; ++i) x[i] = xp[i] + stp * d[i];
        t->f = evaluate(x, g);
        t->dg = dot(g, d);
        t->gg = dot(g, g);
        t->xx = dot(x, x);
        t->dginit = dot(gp, d);
    }

    void accept(int end, int bound) {
        std::vector<double>& s = S[end];
        std::vector<double>& y = Y[end];
        for (int i = 0; i < m; ++i) {
            s[i] = x[i] - xp[i];
            y[i] = g[i] - gp[i];
        }
        const double ys_new = dot(y, s), yy = dot(y, y);
        ys[end] = ys_new;
        x.swap(xp);
        g.swap(gp);
        for (int i = 0; i < m; ++i) d[i] = -gp[i];
        int j = (end + 1) % kHistory;
        for (int b = 0; b < bound; ++b) {
            j = (j + kHistory - 1) % kHistory;
            alpha[j] = dot(S[j], d) / ys[j];
            for (int i = 0; i < m; ++i) d[i] -= alpha[j] * Y[j][i];
        }
        const double sc = ys_new / yy;
        for (int i = 0; i < m; ++i) d[i] *= sc;
        for (int b = 0; b < bound; ++b) {
            const double beta = dot(Y[j], d) / ys[j];
            const double coef = alpha[j] - beta;
            for (int i = 0; i < m; ++i) d[i] += coef * S[j][i];
            j = (j + 1) % kHistory;
        }
    }

    void revert() { result_is_trial = false; }
    void keep_trial() { result_is_trial = true; }
};

// analytic objectives of bioen_hip_selftest_lbfgs (host only, test hook)
static double selftest_objective(int kind, int n, const double* x, double* g) {
    double f = 0.0;
    for (int i = 0; i < n; ++i) g[i] = 0.0;
    if (kind == 0) {   // extended Rosenbrock over consecutive pairs
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
    } else {           // sum_i c_i (x_i - 1)^2 + 0.01 (x_i - 1)^4, c_i spread over 4 decades
        for (int i = 0; i < n; ++i) {
            const double c = std::pow(10.0, 4.0 * i / (n > 1 ? n - 1 : 1) - 2.0);
            const double d = x[i] - 1.0;
            f += c * d * d + 0.01 * d * d * d * d;
            g[i] = 2.0 * c * d + 0.04 * d * d * d;
        }
    }
    return f;
}

struct HostSelftestBackend {
    int kind, n;
    std::vector<double> x, xp, g, gp, d;
    std::vector<double> S[kHistory], Y[kHistory];
    double ys[kHistory] = {}, alpha[kHistory] = {};
    bool result_is_trial = false;

    HostSelftestBackend(int k, int nn, const double* x0) : kind(k), n(nn), x(nn), xp(x0, x0 + nn), g(nn), gp(nn), d(nn) {
        for (int i = 0; i < kHistory; ++i) {
            S[i].assign(nn, 0.0);
            Y[i].assign(nn, 0.0);
        }
    }
    static double dot(const std::vector<double>& a, const std::vector<double>& b) {
        double s = 0.0;
        for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
        return s;
    }
    void initial(double* f, double* gg, double* xx) {
        *f = selftest_objective(kind, n, xp.data(), gp.data());
        *gg = dot(gp, gp);
        *xx = dot(xp, xp);
        for (int i = 0; i < n; ++i) d[i] = -gp[i];
    }
    void trial(double stp, TrialResult* t) {
        for (int i = 0; i < n; ++i) x[i] = xp[i] + stp * d[i];
        t->f = selftest_objective(kind, n, x.data(), g.data());
        t->dg = dot(g, d);
        t->gg = dot(g, g);
        t->xx = dot(x, x);
        t->dginit = dot(gp, d);
    }
    void accept(int end, int bound) {
        for (int i = 0; i < n; ++i) {
            S[end][i] = x[i] - xp[i];
            Y[end][i] = g[i] - gp[i];
        }
        const double ys_new = dot(Y[end], S[end]), yy = dot(Y[end], Y[end]);
        ys[end] = ys_new;
        x.swap(xp);
        g.swap(gp);
        for (int i = 0; i < n; ++i) d[i] = -gp[i];
        int j = (end + 1) % kHistory;
        for (int b = 0; b < bound; ++b) {
            j = (j + kHistory - 1) % kHistory;
            alpha[j] = dot(S[j], d) / ys[j];
            for (int i = 0; i < n; ++i) d[i] -= alpha[j] * Y[j][i];
        }
        const double sc = ys_new / yy;
        for (int i = 0; i < n; ++i) d[i] *= sc;
        for (int b = 0; b < bound; ++b) {
            const double beta = dot(Y[j], d) / ys[j];
            const double coef = alpha[j] - beta;
            for (int i = 0; i < n; ++i) d[i] += coef * S[j][i];
            j = (j + 1) % kHistory;
        }
    }
    void revert() { result_is_trial = false; }
    void keep_trial() { result_is_trial = true; }
};

static void print_config(const bioen_lbfgs_config& p) {
    // same table the reference prints when verbose (c_bioen_kernels_logw.c:620-634)
    std::printf("\t=========================\n");
    std::printf("\tdevice-resident L-BFGS   : gfx950 / HIP\n");
    std::printf("\tlinesearch               : %d\n", p.linesearch);
    std::printf("\tmax_iterations           : %d\n", p.max_iterations);
    std::printf("\tdelta                    : %lf\n", p.delta);
    std::printf("\tepsilon                  : %lf\n", p.epsilon);
    std::printf("\tftol                     : %lf\n", p.ftol);
    std::printf("\tgtol                     : %lf\n", p.gtol);
    std::printf("\twolfe                    : %lf\n", p.wolfe);
    std::printf("\tpast                     : %d\n", p.past);
    std::printf("\tmax_linesearch           : %d\n", p.max_linesearch);
    std::printf("\t=========================\n");
}

static void print_summary(const bioen_hip_ctx* c, const bioen_opt_result& r) {
    std::printf("\t%s\n", lbfgs_code_string(r.lbfgs_code));
    std::printf("\tConfig: m=%d and n=%d\n", c->m, c->n);
    std::printf("\tCurrent function value  = %.6lf\n", r.fmin);
    std::printf("\tIterations              : %d\n", r.iterations);
    std::printf("\tEvaluations             : %d\n", r.evaluations);
    std::printf("\tTime(s) of L-BFGS       : %.12lf\n", r.seconds);
    std::printf("\tTime(s) per iter        : %.12lf\n", r.iterations ? r.seconds / r.iterations : 0.0);
    std::fflush(stdout);
}

static void resolve_timers(bioen_hip_ctx* c) {
    KernelTimer& t = c->timer;
    for (auto& p : t.pending) {
        hipEventSynchronize(p.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            t.total_ms[p.which] += ms;
            t.launches[p.which] += 1;
        }
        t.pool.push_back(p);
    }
    t.pending.clear();
}

}  // namespace bioen

using namespace bioen;

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" {

const char* bioen_hip_version(void) { return "bioen_hip 0.1 (gfx950)"; }

int bioen_hip_device_count(int* count) {
    if (!count) return fail(BIOEN_HIP_EINVAL, "count is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return 0;
}

const char* bioen_hip_strerror(int code) {
    switch (code) {
        case BIOEN_HIP_OK: return "success";
        case BIOEN_HIP_EINVAL: return "invalid argument";
        case BIOEN_HIP_ENODEV: return "no HIP device available";
        case BIOEN_HIP_EHIP: return "HIP runtime error";
        case BIOEN_HIP_ENOMEM: return "out of memory";
        case BIOEN_HIP_ERCCL: return "RCCL error";
        case BIOEN_HIP_ESTATE: return "invalid state";
        default: return "unknown bioen_hip error";
    }
}

const char* bioen_hip_last_error(void) { return g_last_error.c_str(); }
const char* bioen_hip_lbfgs_strerror(int code) { return lbfgs_code_string(code); }
void bioen_hip_set_fast_openmp_flag(int flag) { g_fast_openmp_flag = flag; }
int bioen_hip_get_fast_openmp_flag(void) { return g_fast_openmp_flag; }

int bioen_hip_ctx_create(int m, int n, const double* yTilde, const double* YTilde, int device,
                         bioen_hip_ctx** ctx) {
    if (!yTilde || !YTilde) return fail(BIOEN_HIP_EINVAL, "yTilde / YTilde is NULL");
    bioen_hip_ctx* c = nullptr;
    int rc = ctx_alloc(m, n, device, &c);
    if (rc) return rc;
    hipError_t e = hipMemcpy2DAsync(c->Y, c->ld * sizeof(double), yTilde, (size_t)n * sizeof(double),
                                    (size_t)n * sizeof(double), (size_t)m, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(c->YT, YTilde, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        bioen_hip_ctx_destroy(c);
        return hip_fail(e, "upload of yTilde", __FILE__, __LINE__);
    }
    *ctx = c;
    return 0;
}

int bioen_hip_ctx_create_synthetic(int m, int n, const double* YTrue, const double* sig_sim,
                                   const double* sig_exp, const double* YTilde, unsigned long long seed,
                                   int device, bioen_hip_ctx** ctx) {
    if (!YTrue || !sig_sim || !sig_exp || !YTilde) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    bioen_hip_ctx* c = nullptr;
    int rc = ctx_alloc(m, n, device, &c);
    if (rc) return rc;
    // stage the three M-vectors in ybar / r / um (all mp long), then generate in place
    hipError_t e = hipMemcpyAsync(c->ybar, YTrue, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(c->r, sig_sim, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(c->um, sig_exp, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(c->YT, YTilde, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_generate(c, c->ybar, c->r, c->um, seed);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemsetAsync(c->ybar, 0, c->mp * sizeof(double), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->r, 0, c->mp * sizeof(double), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->um, 0, c->mp * sizeof(double), c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        bioen_hip_ctx_destroy(c);
        return hip_fail(e, "synthetic generation", __FILE__, __LINE__);
    }
    *ctx = c;
    return 0;
}

int bioen_hip_ctx_destroy(bioen_hip_ctx* c) {
    if (!c) return 0;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    bioen_hip_comm_destroy(c);
    resolve_timers(c);
    for (auto& p : c->timer.pool) {
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    double* bufs[] = {c->Y, c->YT, c->ybar, c->r, c->um, c->gm, c->x, c->xp, c->g, c->gp, c->d, c->w,
                      c->fixed, c->a, c->t, c->fwd_partial, c->part, c->scal};
    for (double* p : bufs)
        if (p) hipFree(p);
    for (int i = 0; i < kHistory; ++i) {
        if (c->S[i]) hipFree(c->S[i]);
        if (c->Yh[i]) hipFree(c->Yh[i]);
    }
    if (c->host_scal) hipHostFree(c->host_scal);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int bioen_hip_ctx_shape(const bioen_hip_ctx* c, int* m, int* n) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    if (m) *m = c->m;
    if (n) *n = c->n;
    return 0;
}

int bioen_hip_ctx_read_ytilde(bioen_hip_ctx* c, int row0, int rows, int col0, int cols, double* out) {
    if (!c || !out) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    if (row0 < 0 || col0 < 0 || rows <= 0 || cols <= 0 || row0 + rows > c->m || col0 + cols > c->n)
        return fail(BIOEN_HIP_EINVAL, "block out of range");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    BIOEN_HIP_CHECK(hipMemcpy2DAsync(out, (size_t)cols * sizeof(double), c->Y + (size_t)row0 * c->ld + col0,
                                     c->ld * sizeof(double), (size_t)cols * sizeof(double), (size_t)rows,
                                     hipMemcpyDeviceToHost, c->stream));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int bioen_hip_ctx_set_ytilde_target(bioen_hip_ctx* c, const double* YTilde) {
    if (!c || !YTilde) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    BIOEN_HIP_CHECK(hipMemcpyAsync(c->YT, YTilde, (size_t)c->m * sizeof(double), hipMemcpyHostToDevice, c->stream));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int bioen_hip_synchronize(bioen_hip_ctx* c) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// ---- log-weights ----------------------------------------------------------------------
int bioen_hip_logw_weights(bioen_hip_ctx* c, const double* g, double* w, double* log_s) {
    if (!c || !g) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    int rc = upload_n(c, c->xp, g);
    if (rc) return rc;
    BIOEN_HIP_CHECK(hipMemsetAsync(c->fixed, 0, c->ld * sizeof(double), c->stream));
    launch_trial(c, 0.0);
    launch_logw_exp(c);
    launch_logw_norm(c);
    if ((rc = check_launch())) return rc;
    if (w) BIOEN_HIP_CHECK(hipMemcpyAsync(w, c->w, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if ((rc = read_scalars(c))) return rc;
    if (log_s) *log_s = c->host_scal[S_LOGS];
    return 0;
}

int bioen_hip_logw_fdf(bioen_hip_ctx* c, const double* g, const double* G, double theta, double* f,
                       double* grad) {
    if (!c || !g || !G) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    int rc;
    if ((rc = upload_n(c, c->xp, g))) return rc;
    if ((rc = upload_n(c, c->fixed, G))) return rc;
    launch_logw_logs0(c);
    launch_trial(c, 0.0);
    enqueue_logw_eval(c, theta, grad != nullptr);
    if ((rc = check_launch())) return rc;
    if (grad)
        BIOEN_HIP_CHECK(hipMemcpyAsync(grad, c->g, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if ((rc = read_scalars(c))) return rc;
    if (f) *f = c->host_scal[S_F];
    return 0;
}

int bioen_hip_opt_lbfgs_logw(bioen_hip_ctx* c, const double* g0, const double* G, double theta,
                             const bioen_lbfgs_config* config, const bioen_visual_params* visual,
                             double* result, double* w_opt, bioen_opt_result* info) {
    if (!c || !g0 || !G || !config || !result || !info) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    const bool verbose = visual && visual->verbose;
    std::memset(info, 0, sizeof *info);
    int rc;
    if ((rc = ensure_history(c))) return rc;
    if ((rc = upload_n(c, c->xp, g0))) return rc;
    if ((rc = upload_n(c, c->fixed, G))) return rc;
    BIOEN_HIP_CHECK(hipMemsetAsync(c->d, 0, c->ld * sizeof(double), c->stream));
    BIOEN_HIP_CHECK(hipMemsetAsync(bioen::part(c, P_DGINIT), 0, kMaxPartials * sizeof(double), c->stream));
    launch_logw_logs0(c);
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));

    if (verbose) {
        std::printf("L-BFGS minimizer\n");
        print_config(*config);
    }
    DeviceLogwBackend B{c, theta};
    const auto t0 = std::chrono::steady_clock::now();
    double fx = 0.0;
    info->lbfgs_code = lbfgs_run(B, c->n, *config, &fx, &info->iterations, &info->evaluations);
    hipStreamSynchronize(c->stream);
    info->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    info->fmin = fx;
    if (B.rc) return B.rc;
    if ((rc = check_launch())) return rc;

    const double* res = B.result_is_trial ? c->x : c->xp;
    if (info->evaluations > 0) {
        if (!B.result_is_trial) {
            // re-establish w, chi^2, KL at the accepted point (one forward pass, outside the timing)
            BIOEN_HIP_CHECK(hipMemcpyAsync(c->x, c->xp, c->ld * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            launch_max(c, c->x);
            enqueue_logw_eval(c, theta, false);
            res = c->x;
        }
        if ((rc = read_scalars(c))) return rc;
        info->chi2 = 0.5 * c->host_scal[S_CHI];
        info->kl = c->host_scal[S_P] - c->host_scal[S_LOGS] + c->host_scal[S_LOGS0];
        if (w_opt)
            BIOEN_HIP_CHECK(hipMemcpyAsync(w_opt, c->w, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost,
                                           c->stream));
    } else {
        res = c->xp;
    }
    BIOEN_HIP_CHECK(hipMemcpyAsync(result, res, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (verbose) print_summary(c, *info);
    return 0;
}

// ---- forces ---------------------------------------------------------------------------
static int upload_forces_inputs(bioen_hip_ctx* c, const double* forces, const double* w0) {
    BIOEN_HIP_CHECK(hipMemcpyAsync(c->um, forces, (size_t)c->m * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return upload_n(c, c->fixed, w0);
}

int bioen_hip_forces_weights(bioen_hip_ctx* c, const double* forces, const double* w0, double* w) {
    if (!c || !forces || !w0 || !w) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    int rc;
    if ((rc = upload_forces_inputs(c, forces, w0))) return rc;
    enqueue_forces_weights(c);
    if ((rc = check_launch())) return rc;
    BIOEN_HIP_CHECK(hipMemcpyAsync(w, c->w, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int bioen_hip_forces_fdf(bioen_hip_ctx* c, const double* forces, const double* w0, double theta, double* f,
                         double* grad) {
    if (!c || !forces || !w0) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    int rc;
    if ((rc = upload_forces_inputs(c, forces, w0))) return rc;
    enqueue_forces_eval(c, theta, grad != nullptr);
    if ((rc = check_launch())) return rc;
    if (grad)
        BIOEN_HIP_CHECK(hipMemcpyAsync(grad, c->gm, (size_t)c->m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if ((rc = read_scalars(c))) return rc;
    if (f) *f = c->host_scal[S_F];
    return 0;
}

int bioen_hip_opt_lbfgs_forces(bioen_hip_ctx* c, const double* forces0, const double* w0, double theta,
                               const bioen_lbfgs_config* config, const bioen_visual_params* visual,
                               double* result, double* w_opt, bioen_opt_result* info) {
    if (!c || !forces0 || !w0 || !config || !result || !info) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    const bool verbose = visual && visual->verbose;
    std::memset(info, 0, sizeof *info);
    int rc;
    if ((rc = upload_n(c, c->fixed, w0))) return rc;
    if (verbose) {
        std::printf("L-BFGS minimizer\n");
        print_config(*config);
    }
    HostForcesBackend B(c, theta, forces0);
    const auto t0 = std::chrono::steady_clock::now();
    double fx = 0.0;
    info->lbfgs_code = lbfgs_run(B, c->m, *config, &fx, &info->iterations, &info->evaluations);
    hipStreamSynchronize(c->stream);
    info->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    info->fmin = fx;
    if (B.rc) return B.rc;
    if ((rc = check_launch())) return rc;

    const std::vector<double>& res = B.result_is_trial ? B.x : B.xp;
    std::memcpy(result, res.data(), (size_t)c->m * sizeof(double));
    if (info->evaluations > 0) {
        // weights, chi^2 and KL at the returned forces (forces.py:535-548 recomputes them too)
        BIOEN_HIP_CHECK(hipMemcpyAsync(c->um, result, (size_t)c->m * sizeof(double), hipMemcpyHostToDevice, c->stream));
        enqueue_forces_eval(c, theta, false);
        if ((rc = check_launch())) return rc;
        if (w_opt)
            BIOEN_HIP_CHECK(hipMemcpyAsync(w_opt, c->w, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost,
                                           c->stream));
        if ((rc = read_scalars(c))) return rc;
        info->chi2 = 0.5 * c->host_scal[S_CHI];
        info->kl = c->host_scal[S_KL];
    }
    if (verbose) print_summary(c, *info);
    return 0;
}

// ---- shared ---------------------------------------------------------------------------
int bioen_hip_chi_squared(bioen_hip_ctx* c, const double* w, double* yave, double* chi2) {
    if (!c || !w) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    int rc;
    if ((rc = upload_n(c, c->w, w))) return rc;
    launch_fwd_partial(c, c->w);
    launch_fwd_rows_residual(c);
    launch_forces_scalars(c, 0.0);   // S_CHI (the KL partials it also sums are irrelevant here)
    if ((rc = check_launch())) return rc;
    if (yave)
        BIOEN_HIP_CHECK(hipMemcpyAsync(yave, c->ybar, (size_t)c->m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if ((rc = read_scalars(c))) return rc;
    if (chi2) *chi2 = 0.5 * c->host_scal[S_CHI];
    return 0;
}

// ---- host self-test -----------------------------------------------------------------------
int bioen_hip_selftest_lbfgs(int kind, int n, const double* x0, const bioen_lbfgs_config* config, double* x_out,
                             bioen_opt_result* info) {
    if (!x0 || !config || !x_out || !info || n <= 0 || kind < 0 || kind > 1)
        return fail(BIOEN_HIP_EINVAL, "bad argument");
    std::memset(info, 0, sizeof *info);
    HostSelftestBackend B(kind, n, x0);
    double fx = 0.0;
    info->lbfgs_code = lbfgs_run(B, n, *config, &fx, &info->iterations, &info->evaluations);
    info->fmin = fx;
    const std::vector<double>& res = B.result_is_trial ? B.x : B.xp;
    std::memcpy(x_out, res.data(), (size_t)n * sizeof(double));
    return 0;
}

// ---- measurement ------------------------------------------------------------------------
int bioen_hip_kernel_stats(bioen_hip_ctx* c, int which, double* total_ms, long long* launches) {
    if (!c || which < 0 || which > 1) return fail(BIOEN_HIP_EINVAL, "bad argument");
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    resolve_timers(c);
    if (total_ms) *total_ms = c->timer.total_ms[which];
    if (launches) *launches = c->timer.launches[which];
    return 0;
}

int bioen_hip_kernel_stats_reset(bioen_hip_ctx* c) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    resolve_timers(c);
    c->timer.total_ms[0] = c->timer.total_ms[1] = 0.0;
    c->timer.launches[0] = c->timer.launches[1] = 0;
    return 0;
}

int bioen_hip_kernel_stats_enable(bioen_hip_ctx* c, int enable) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    c->timer.enabled = enable != 0;
    return 0;
}

// ---- RCCL (resolved lazily so single-GPU use never needs librccl) ---------------------------
namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.h) return 0;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
    }
    for (const char* nm : names) {
        if (h) break;
        h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) return fail(BIOEN_HIP_ERCCL, "librccl.so not found");
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(h, "ncclAllGather"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy)
        return fail(BIOEN_HIP_ERCCL, "librccl.so lacks a required symbol");
    g_rccl.h = h;
    return 0;
}

int rccl_fail(ncclResult_t r, const char* what) {
    std::string s = std::string(what) + " failed: " +
                    (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "(no error string)");
    set_last_error(s);
    return BIOEN_HIP_ERCCL;
}
}  // namespace

int bioen_hip_comm_unique_id(unsigned char id[128]) {
    if (!id) return fail(BIOEN_HIP_EINVAL, "id is NULL");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId u;
    ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
    std::memcpy(id, u.internal, 128);
    return 0;
}

int bioen_hip_comm_init(bioen_hip_ctx* c, const unsigned char id[128], int rank, int nranks) {
    if (!c || !id || nranks <= 0 || rank < 0 || rank >= nranks) return fail(BIOEN_HIP_EINVAL, "bad argument");
    if (c->comm) return fail(BIOEN_HIP_ESTATE, "communicator already initialised");
    int rc = load_rccl();
    if (rc) return rc;
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, 128);
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, u, rank);
    if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitRank");
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_nranks = nranks;
    return 0;
}

int bioen_hip_comm_allgather(bioen_hip_ctx* c, const double* send, size_t count, double* recv) {
    if (!c || !send || !recv || count == 0) return fail(BIOEN_HIP_EINVAL, "bad argument");
    if (!c->comm) return fail(BIOEN_HIP_ESTATE, "communicator not initialised");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    const size_t need = count * (size_t)(c->comm_nranks + 1);
    if (c->comm_buf_count < need) {
        if (c->comm_buf) hipFree(c->comm_buf);
        c->comm_buf = nullptr;
        c->comm_buf_count = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->comm_buf), need * sizeof(double));
        if (e != hipSuccess) {
            hip_fail(e, "hipMalloc", __FILE__, __LINE__);
            return BIOEN_HIP_ENOMEM;
        }
        c->comm_buf_count = need;
    }
    double* dsend = c->comm_buf;
    double* drecv = c->comm_buf + count;
    BIOEN_HIP_CHECK(hipMemcpyAsync(dsend, send, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    ncclResult_t r = g_rccl.AllGather(dsend, drecv, count, ncclDouble, static_cast<ncclComm_t>(c->comm), c->stream);
    if (r != ncclSuccess) return rccl_fail(r, "ncclAllGather");
    BIOEN_HIP_CHECK(hipMemcpyAsync(recv, drecv, count * c->comm_nranks * sizeof(double), hipMemcpyDeviceToHost,
                                   c->stream));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int bioen_hip_comm_destroy(bioen_hip_ctx* c) {
    if (!c) return 0;
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
    if (c->comm_buf) hipFree(c->comm_buf);
    c->comm_buf = nullptr;
    c->comm_buf_count = 0;
    return 0;
}

}  // extern "C"
