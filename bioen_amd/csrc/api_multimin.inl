// GSL-style minimizers on the device objective (part of api.hip's translation unit).
//   log-weights: the N variables, the gradient and all of the minimizer's work vectors stay in HBM
//                (DeviceVectors); per evaluation only f crosses PCIe, per inner product 8 bytes.
//   forces:      the M variables live on the host (HostVectors), the objective on the device.
// Reference: _opt_bfgs_logw / _opt_bfgs_forces (c_bioen_kernels_logw.c:366-509,
// c_bioen_kernels_forces.c), which hand BioEn's f / df / fdf to gsl_multimin_fdfminimizer_*.

namespace bioen {

class DeviceVectors {
public:
    DeviceVectors(bioen_hip_ctx* ctx, double theta_) : c(ctx), theta(theta_) {
        keep(alloc_slot(c, 0, true));          // the L-BFGS history buffers double as work vectors
        ProblemSlot& s = c->slot[0];
        if (rc) return;
        double* pool[multimin::V_COUNT] = {s.xa, s.xb, s.ga, s.gb, s.d, s.S[0], s.S[1], s.S[2], s.S[3], s.S[4],
                                           s.S[5], s.Yh[0], s.Yh[1]};
        for (int i = 0; i < multimin::V_COUNT; ++i) {
            v[i] = pool[i];
            note(hipMemsetAsync(v[i], 0, c->ld * sizeof(double), c->stream));
        }
        const int one[1] = {0};
        base = make_round(c, one, 1, nullptr, &theta);
    }
    int size() const { return (int)std::min<long long>(c->n_global, 0x7fffffff); }      // the problem's dimension (all ranks')
    double* data(int h) { return v[h]; }
    bool failed() const { return rc != 0; }
    int error() const { return rc; }

    void copy(int dst, int src) {
        note(hipMemcpyAsync(v[dst], v[src], c->ld * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        touch(dst);
    }
    void zero(int h) {
        note(hipMemsetAsync(v[h], 0, c->ld * sizeof(double), c->stream));
        touch(h);
    }
    void axpy(double a, int x, int y) {
        if (a == 0.0) return;
        launch_vaxpy(c, a, v[x], v[y]);
        touch(y);
    }
    void scal(double a, int x) {
        launch_vscal(c, a, v[x]);
        touch(x);
    }
    void step(int x, int p, double coef, int x1, int dx) {
        launch_vstep(c, v[x], v[p], coef, v[x1], v[dx]);
        touch(x1);
        touch(dx);
    }
    void dots(int k, const int* xs, const int* ys, double* out) { reduce(k, 0, xs, ys, out); }
    double dot(int x, int y) {
        double o;
        reduce(1, 0, &x, &y, &o);
        return o;
    }
    double nrm2(int x) { return std::sqrt(dot(x, x)); }
    bool equal(int x, int y) {
        const int xs[2] = {x, x}, ys[2] = {y, y};
        double o[2];
        reduce(2, 1, xs, ys, o);
        return o[0] == 0.0;
    }
    double absmax(int x) {
        const int xs[2] = {x, x}, ys[2] = {x, x};
        double o[2];
        reduce(2, 1, xs, ys, o);
        return o[1];
    }

    // f only: the forward matrix pass; remembers for which vector content it is valid
    double eval_f(int x) {
        if (rc) return 0.0;
        forward(x);
        return fetch_f();
    }
    // gradient: the adjoint pass alone when the forward state of this very point is still in place
    // (GSL's line search asks f(alpha) first and f'(alpha) only if the point is acceptable)
    void eval_df(int x, int g) {
        if (rc) return;
        Round r = round_for(x, g);
        if (!(fwd_vec == x && fwd_version == version[x])) {
            forward(x);
            ++saved_none;
        } else {
            ++saved_forward;
        }
        keep(enqueue_logw_adjoint(c, r));
        keep(check_launch());
        touch(g);
    }
    void eval_fdf(int x, double* f, int g) {
        *f = 0.0;
        if (rc) return;
        eval_df(x, g);
        *f = fetch_f();
    }
    // leave w (and chi^2, S) of `x` in slot 0 / host_scal
    int finalize(int x) {
        if (rc) return rc;
        forward(x);
        fetch_f();
        launch_scale_w(c, round_for(x, -1));   // e -> w
        keep(check_launch());
        fwd_vec = -1;                          // slot.w no longer holds e
        return rc;
    }
    long long saved_forward = 0, saved_none = 0;

private:
    bioen_hip_ctx* c;
    double theta;
    double* v[multimin::V_COUNT];
    long long version[multimin::V_COUNT] = {};
    int fwd_vec = -1;
    long long fwd_version = -1;
    Round base;
    int rc = 0;

    void note(hipError_t e) {
        if (e != hipSuccess && !rc) rc = hip_fail(e, "multimin device op", __FILE__, __LINE__);
    }
    void keep(int r) {
        if (r && !rc) rc = r;
    }
    void touch(int h) { ++version[h]; }
    Round round_for(int x, int g) const {
        Round r = base;
        r.x[0] = v[x];
        r.g[0] = g >= 0 ? v[g] : base.g[0];
        r.d[0] = v[x];          // k_logw_grad's g.d by-product is not used here; any resident vector does
        return r;
    }
    void forward(int x) {
        const Round r = round_for(x, -1);
        launch_max(c, r);
        keep(enqueue_logw_eval(c, r, false));
        keep(check_launch());
        fwd_vec = x;
        fwd_version = version[x];
    }
    double fetch_f() {
        keep(read_scalars(c));
        return c->host_scal[S_F];
    }
    void reduce(int k, int mode, const int* xs, const int* ys, double* out) {
        for (int q = 0; q < k; ++q) out[q] = 0.0;
        if (rc) return;
        VDotArgs a{};
        a.k = k;
        a.mode = mode;
        for (int q = 0; q < k; ++q) {
            a.x[q] = v[xs[q]];
            a.y[q] = v[ys[q]];
        }
        ProblemSlot& s = c->slot[0];
        launch_vdots_part(c, a);
        keep(exchange(c, X_GRAD, 4 * (size_t)vec_grid(c)));      // sharded contexts: every rank sums the same segments' partials
        launch_vdots_finish(c, a, s.scal + S_SPARE0 + 0);
        // S_SPARE0, S_SPARE1, S_YSH.. are contiguous: 4 doubles are free while no L-BFGS history is live
        double* host = c->host_scal + kScalStride;   // second slot's mirror: untouched by read_scalars(c, 1)
        note(hipMemcpyAsync(host, s.scal + S_SPARE0, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        note(hipStreamSynchronize(c->stream));
        for (int q = 0; q < k; ++q) out[q] = host[q];
    }
};

static void print_gsl_config(const bioen_gsl_config& p, bool verbose) {
    if (!verbose) return;
    std::printf("\t=========================\n");
    std::printf("\tGSL-style minimizer      : %s\n", multimin::algorithm_name(p.algorithm));
    std::printf("\ttol                      : %f\n", p.tol);
    std::printf("\tstep_size                : %f\n", p.step_size);
    std::printf("\tmax_iteration            : %d\n", p.max_iterations);
    std::printf("\t=========================\n");
}

static void print_gsl_summary(const bioen_hip_ctx* c, const bioen_opt_result& r, bool verbose) {
    if (!verbose) return;
    std::printf("Optimization terminated, status %d (%s)\n", r.lbfgs_code, multimin::status_string(r.lbfgs_code));
    std::printf("\tConfig: m=%d and n=%lld\n", c->m, c->n_global);
    std::printf("\tCurrent function value  = %.6lf\n", r.fmin);
    std::printf("\tIterations              : %d\n", r.iterations);
    std::printf("\tEvaluations             : %d\n", r.evaluations);
    std::printf("\tMinimization time [s]   : %.12lf\n", r.seconds);
}

}  // namespace bioen

extern "C" {

const char* bioen_hip_gsl_strerror(int code) { return multimin::status_string(code); }

int bioen_hip_opt_gsl_logw(bioen_hip_ctx* c, const double* g0, const double* G, double theta,
                           const bioen_gsl_config* config, const bioen_visual_params* visual, double* result,
                           double* w_opt, bioen_opt_result* info) {
    if (!c || !g0 || !G || !config || !result || !info) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    if (config->algorithm < 0 || config->algorithm > 4) return fail(BIOEN_HIP_EINVAL, "unknown GSL algorithm id");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    const bool verbose = visual && visual->verbose;
    print_gsl_config(*config, verbose);
    std::memset(info, 0, sizeof *info);
    const auto t0 = std::chrono::steady_clock::now();
    int rc;
    DeviceVectors B(c, theta);
    if (B.failed()) return B.error();
    if ((rc = upload_n(c, B.data(multimin::V_X), g0))) return rc;
    if ((rc = upload_n(c, c->fixed, G))) return rc;
    {
        const int one[1] = {0};
        if (int e0 = enqueue_logs0(c, make_round(c, one, 1, nullptr, &theta))) return e0;
    }
    const multimin::Config cfg{config->step_size, config->tol, config->max_iterations, config->algorithm};
    const multimin::Outcome out = multimin::run(B, cfg);
    if (B.failed()) return B.error();
    if ((rc = B.finalize(multimin::V_X))) return rc;        // w, chi^2, S at the result
    if ((rc = download_n(c, result, B.data(multimin::V_X)))) return rc;
    if (w_opt && (rc = download_n(c, w_opt, c->slot[0].w))) return rc;
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));      // pageable destinations, work queued by finalize()
    if ((rc = check_launch())) return rc;
    const double* h = c->host_scal;
    info->fmin = out.fmin;
    info->chi2 = 0.5 * h[S_CHI];
    info->kl = h[S_P] - h[S_LOGS] + h[S_LOGS0];
    info->lbfgs_code = out.status;
    info->iterations = out.iterations;
    info->evaluations = out.f_evaluations + out.g_evaluations;
    info->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    print_gsl_summary(c, *info, verbose);
    return 0;
}

int bioen_hip_opt_gsl_forces(bioen_hip_ctx* c, const double* forces0, const double* w0, double theta,
                             const bioen_gsl_config* config, const bioen_visual_params* visual, double* result,
                             double* w_opt, bioen_opt_result* info) {
    if (!c || !forces0 || !w0 || !config || !result || !info) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    if (config->algorithm < 0 || config->algorithm > 4) return fail(BIOEN_HIP_EINVAL, "unknown GSL algorithm id");
    int rc = forces_guard(c);
    if (rc) return rc;
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    const bool verbose = visual && visual->verbose;
    print_gsl_config(*config, verbose);
    std::memset(info, 0, sizeof *info);
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = upload_n(c, c->fixed, w0))) return rc;
    bioen_lbfgs_config unused{};
    ForcesBatchEngine eng(c, unused, false);
    const int one[1] = {0};
    const int m = c->m;
    multimin::HostVectors B(m, [&](const double* x, double* f, double* grad) -> int {
        const double* pt[1] = {x};
        eng.evaluate(one, 1, pt, &theta, grad != nullptr);
        if (eng.rc) return eng.rc;
        if (grad) std::memcpy(grad, eng.gm_h, (size_t)m * sizeof(double));
        *f = c->host_scal[S_F];
        return 0;
    });
    std::memcpy(B.data(multimin::V_X), forces0, (size_t)m * sizeof(double));
    const multimin::Config cfg{config->step_size, config->tol, config->max_iterations, config->algorithm};
    const multimin::Outcome out = multimin::run(B, cfg);
    if (B.failed()) return B.error();
    {   // w, chi^2, S at the result
        const double* pt[1] = {B.data(multimin::V_X)};
        eng.evaluate(one, 1, pt, &theta, false);
        if (eng.rc) return eng.rc;
    }
    std::memcpy(result, B.data(multimin::V_X), (size_t)m * sizeof(double));
    if (w_opt && (rc = download_n(c, w_opt, c->slot[0].w))) return rc;
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));      // pageable destination
    if ((rc = check_launch())) return rc;
    const double* h = c->host_scal;
    info->fmin = out.fmin;
    info->chi2 = 0.5 * h[S_CHI];
    info->kl = h[S_KL];
    info->lbfgs_code = out.status;
    info->iterations = out.iterations;
    info->evaluations = out.f_evaluations + out.g_evaluations;
    info->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    print_gsl_summary(c, *info, verbose);
    return 0;
}

// GSL's multimin test programme on the host backend (no GPU needed): kind 0 Roth, 1 Wood,
// 2 Rosenbrock, 3 SimpleAbs; protocol of multimin/test.c:106-160.
int bioen_hip_selftest_multimin(int algorithm, int kind, const double* x0, double* x_out, bioen_opt_result* info) {
    if (!x0 || !x_out || !info || algorithm < 0 || algorithm > 4 || kind < 0 || kind > 3)
        return fail(BIOEN_HIP_EINVAL, "bad argument");
    std::memset(info, 0, sizeof *info);
    const int n = multimin::test_function_dim(kind);
    multimin::HostVectors B(n, [kind](const double* x, double* f, double* grad) -> int {
        multimin::test_function(kind, x, f, grad);
        return 0;
    });
    std::memcpy(B.data(multimin::V_X), x0, (size_t)n * sizeof(double));
    const multimin::Outcome out = multimin::run_gsl_test(B, algorithm);
    std::memcpy(x_out, B.data(multimin::V_X), (size_t)n * sizeof(double));
    info->fmin = out.fmin;
    info->lbfgs_code = out.status;
    info->iterations = out.iterations;
    info->evaluations = out.f_evaluations + out.g_evaluations;
    info->reserved = out.g_evaluations;
    return 0;
}

int bioen_hip_multimin_host(int n, bioen_host_objective objective, void* user, const double* x0,
                            const bioen_gsl_config* config, double* x_out, bioen_opt_result* info) {
    if (n <= 0 || !objective || !x0 || !config || !x_out || !info) return fail(BIOEN_HIP_EINVAL, "bad argument");
    if (config->algorithm < 0 || config->algorithm > 4) return fail(BIOEN_HIP_EINVAL, "unknown GSL algorithm id");
    std::memset(info, 0, sizeof *info);
    const auto t0 = std::chrono::steady_clock::now();
    multimin::HostVectors B(n, [&](const double* x, double* f, double* grad) -> int { return objective(user, x, f, grad); });
    std::memcpy(B.data(multimin::V_X), x0, (size_t)n * sizeof(double));
    const multimin::Config cfg{config->step_size, config->tol, config->max_iterations, config->algorithm};
    const multimin::Outcome out = multimin::run(B, cfg);
    if (B.failed()) return fail(BIOEN_HIP_ESTATE, "the caller's objective reported an error");
    std::memcpy(x_out, B.data(multimin::V_X), (size_t)n * sizeof(double));
    info->fmin = out.fmin;
    info->lbfgs_code = out.status;
    info->iterations = out.iterations;
    info->evaluations = out.f_evaluations + out.g_evaluations;
    info->reserved = out.g_evaluations;
    info->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

}  // extern "C"
