// Lock-step batch engine of the forces method (part of api.hip's translation unit: uses its static helpers).

// ---------------------------------------------------------------------------------
// forces method: the M variables of each problem live on the host (a few KB), the K problems of
// a round share the matrix passes of the evaluation (two strip passes for M <= 1024, else four).
// ---------------------------------------------------------------------------------
struct ForcesProblem {
    int id = -1;
    double theta = 0.0;
    LbfgsMachine* machine = nullptr;
    bool initial = true;
    std::vector<double> x, xp, g, gp, d;
    std::vector<double> S[kHistory], Y[kHistory];
    double ys[kHistory] = {}, alpha[kHistory] = {};
    std::chrono::steady_clock::time_point t0;

    static double dot(const std::vector<double>& a, const std::vector<double>& b) {
        double s = 0.0;
        for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
        return s;
    }
    void start(int m, const double* x0) {
        x.assign(m, 0.0);
        xp.assign(x0, x0 + m);
        g.assign(m, 0.0);
        gp.assign(m, 0.0);
        d.assign(m, 0.0);
        for (int i = 0; i < kHistory; ++i) {
            S[i].assign(m, 0.0);
            Y[i].assign(m, 0.0);
        }
        initial = true;
    }
    // lbfgs.c:543-598 on host vectors
    void accept(int end, int bound) {
        const int m = (int)x.size();
        for (int i = 0; i < m; ++i) {
            S[end][i] = x[i] - xp[i];
            Y[end][i] = g[i] - gp[i];
        }
        const double ys_new = dot(Y[end], S[end]), yy = dot(Y[end], Y[end]);
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
};

struct ForcesBatchEngine {
    bioen_hip_ctx* c;
    const bioen_lbfgs_config& cfg;
    bool verbose;
    int rc = 0;
    double *um_h = nullptr, *gm_h = nullptr;     // pinned staging (pageable memory would make every copy a blocking one)

    ForcesBatchEngine(bioen_hip_ctx* ctx, const bioen_lbfgs_config& config, bool verb)
        : c(ctx), cfg(config), verbose(verb) {
        const size_t cnt = (size_t)c->mp * kMaxBatch;
        if (!c->host_m)
            note(hipHostMalloc(reinterpret_cast<void**>(&c->host_m), 2 * cnt * sizeof(double), hipHostMallocDefault),
                 "hipHostMalloc");
        um_h = c->host_m;
        gm_h = c->host_m ? c->host_m + cnt : nullptr;
    }

    void note(int e) { if (e && !rc) rc = e; }
    void note(hipError_t e, const char* what) { if (e != hipSuccess && !rc) rc = hip_fail(e, what, __FILE__, __LINE__); }

    // evaluate the K points pts[a] (each m long); f -> host_scal, gradients -> gm_h (compact)
    void evaluate(const int* slots, int k, const double* const* pts, const double* thetas, bool with_grad) {
        const int m = c->m;
        if (rc) return;
        std::fill(um_h, um_h + (size_t)c->mp * k, 0.0);
        for (int a = 0; a < k; ++a)
            for (int i = 0; i < m; ++i) um_h[(size_t)i * k + a] = pts[a][i];
        note(hipMemcpyAsync(c->um, um_h, (size_t)c->mp * k * sizeof(double), hipMemcpyHostToDevice, c->stream),
             "forces H2D");
        const ForcesRound fr = make_forces_round(c, slots, k, thetas);
        const Round r = make_round(c, slots, k, nullptr, thetas);
        note(enqueue_forces_eval(c, fr, r, with_grad));
        if (with_grad)
            note(hipMemcpyAsync(gm_h, c->gm, (size_t)c->mp * k * sizeof(double), hipMemcpyDeviceToHost, c->stream),
                 "gradient D2H");
        note(read_scalars(c, kMaxBatch));
        note(check_launch());
    }

    int run(int ntheta, const double* thetas, const double* f0, size_t f0_stride, const double* w0_host, int max_batch,
            double* results, double* w_opt, bioen_opt_result* infos) {
        const int m = c->m;
        for (int i = 0; i < ntheta; ++i) std::memset(&infos[i], 0, sizeof(bioen_opt_result));
        LbfgsMachine probe(m, cfg);
        const int bad = probe.validate();
        if (bad != 0) {
            for (int i = 0; i < ntheta; ++i) {
                infos[i].lbfgs_code = bad;
                std::memcpy(results + (size_t)i * m, f0 + (size_t)i * f0_stride, (size_t)m * sizeof(double));
            }
            return 0;
        }
        const int kb = std::max(1, std::min(std::min(max_batch, kMaxBatch), ntheta));
        for (int s = 0; s < kb; ++s) note(alloc_slot(c, s, false));
        if (rc) return rc;
        note(upload_n(c, c->fixed, w0_host));

        std::vector<LbfgsMachine> machines;
        machines.reserve(ntheta);
        for (int i = 0; i < ntheta; ++i) machines.emplace_back(m, cfg);
        std::vector<ForcesProblem> probs(kb);
        bool occupied[kMaxBatch] = {};
        int next = 0, active = 0;

        auto start_problem = [&](int s) {
            ForcesProblem& p = probs[s];
            p.id = next;
            p.theta = thetas[next];
            p.machine = &machines[next];
            p.start(m, f0 + (size_t)next * f0_stride);
            p.t0 = std::chrono::steady_clock::now();
            occupied[s] = true;
            ++active;
            ++next;
        };
        auto finish_problem = [&](int s, int code, bool keep_trial) {
            ForcesProblem& p = probs[s];
            bioen_opt_result& info = infos[p.id];
            info.lbfgs_code = code;
            info.iterations = p.machine->iterations();
            info.evaluations = p.machine->evaluations();
            info.fmin = p.machine->fx();
            const std::vector<double>& res = keep_trial ? p.x : p.xp;
            std::memcpy(results + (size_t)p.id * m, res.data(), (size_t)m * sizeof(double));
            // weights, chi^2 and KL at the returned forces (forces.py:535-548 recomputes them too)
            const int one[1] = {s};
            const double* pt[1] = {res.data()};
            evaluate(one, 1, pt, &p.theta, false);
            const double* h = c->host_scal + (size_t)s * kScalStride;
            info.chi2 = 0.5 * h[S_CHI];
            info.kl = h[S_KL];
            if (w_opt) {
                note(download_n(c, w_opt + (size_t)p.id * c->n_global, c->slot[s].w));   // gathers the ranks' blocks
                note(hipStreamSynchronize(c->stream), "sync");
            }
            info.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - p.t0).count();
            if (verbose) {
                std::printf("\ttheta = %g\n", p.theta);
                print_summary(c, info);
            }
            occupied[s] = false;
            --active;
        };

        for (int s = 0; s < kb && next < ntheta; ++s) start_problem(s);
        while (active > 0 && !rc) {
            int list[kMaxBatch];
            double th[kMaxBatch];
            const double* pts[kMaxBatch];
            int k = 0;
            for (int s = 0; s < kb; ++s) {
                if (!occupied[s]) continue;
                ForcesProblem& p = probs[s];
                if (p.initial) {
                    pts[k] = p.xp.data();
                } else {
                    const double stp = p.machine->trial_step();
                    for (int i = 0; i < m; ++i) p.x[i] = p.xp[i] + stp * p.d[i];
                    pts[k] = p.x.data();
                }
                list[k] = s;
                th[k] = p.theta;
                ++k;
            }
            evaluate(list, k, pts, th, true);
            if (rc) break;
            for (int a = 0; a < k; ++a) {
                const int s = list[a];
                ForcesProblem& p = probs[s];
                const double f = c->host_scal[(size_t)s * kScalStride + S_F];
                std::vector<double>& grad = p.initial ? p.gp : p.g;
                for (int i = 0; i < m; ++i) grad[i] = gm_h[(size_t)i * k + a];
                LbfgsMachine::Action act;
                if (p.initial) {
                    act = p.machine->on_initial(f, ForcesProblem::dot(p.gp, p.gp), ForcesProblem::dot(p.xp, p.xp));
                    if (act.kind != LbfgsMachine::DONE) {
                        for (int i = 0; i < m; ++i) p.d[i] = -p.gp[i];
                        p.initial = false;
                    }
                } else {
                    TrialResult t{f, ForcesProblem::dot(p.g, p.d), ForcesProblem::dot(p.g, p.g),
                                  ForcesProblem::dot(p.x, p.x), ForcesProblem::dot(p.gp, p.d)};
                    act = p.machine->on_trial(t);
                    if (act.kind == LbfgsMachine::ACCEPT) p.accept(act.end, act.bound);
                }
                if (act.kind == LbfgsMachine::DONE) {
                    finish_problem(s, act.code, act.keep_trial && !p.initial);
                    if (next < ntheta && !rc) start_problem(s);
                }
            }
        }
        note(hipStreamSynchronize(c->stream), "sync");
        return rc;
    }
};

