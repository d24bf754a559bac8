// Lock-step batch engine of the log-weights method (part of api.hip's translation unit: uses its static helpers).

// ---------------------------------------------------------------------------------
// lock-step batch engine for the log-weights method: up to kMaxBatch thetas advance one
// evaluation per round and share both matrix passes of that round.
// ---------------------------------------------------------------------------------
struct BatchProblem {
    int id = -1;                 // index into the caller's theta list
    double theta = 0.0;
    LbfgsMachine* machine = nullptr;
    bool initial = true;         // next evaluation is the one at the start point
    bool need_direction = false; // build d before the next trial
    bool accept = false;         //   ... after committing the (s, y) pair
    int end = 0, bound = 0;
    std::chrono::steady_clock::time_point t0;
};

struct LogwBatchEngine {
    bioen_hip_ctx* c;
    const bioen_lbfgs_config& cfg;
    bool verbose;
    int rc = 0;

    LogwBatchEngine(bioen_hip_ctx* ctx, const bioen_lbfgs_config& config, bool verb)
        : c(ctx), cfg(config), verbose(verb) {}

    void note(int e) { if (e && !rc) rc = e; }
    void note(hipError_t e, const char* what) { if (e != hipSuccess && !rc) rc = hip_fail(e, what, __FILE__, __LINE__); }

    bool use_gram() const {
        return c->direction_mode != 1;   // auto = Gram form
    }

    // Gram form (see kernels.hpp: GramArgs): 3 launches and one exchange for all accepting problems
    void directions_gram(BatchProblem* slots, const std::vector<int>& list) {
        GramArgs ga{};
        ga.n = (int)list.size();
        for (int a = 0; a < ga.n; ++a) {
            BatchProblem& p = slots[list[a]];
            ProblemSlot& sl = c->slot[list[a]];
            std::swap(sl.x, sl.xp);           // the trial point becomes the accepted point
            std::swap(sl.g, sl.gp);
            ga.xnew[a] = sl.xp; ga.xold[a] = sl.x; ga.gnew[a] = sl.gp; ga.gold[a] = sl.g;
            for (int i = 0; i < kHistory; ++i) {
                ga.S[a][i] = sl.S[i];
                ga.Y[a][i] = sl.Yh[i];
            }
            ga.d[a] = sl.d; ga.gram[a] = sl.gram; ga.scal[a] = sl.scal;
            ga.end[a] = p.end;
            ga.bound[a] = p.bound;
        }
        launch_gram(c, ga);
        if (c->world > 1) {                       // ship 39 totals per problem, not 39 x blocks partials
            launch_gram_rank_reduce(c, ga.n);
            note(exchange(c, X_GRAMR, (size_t)kGramDots * ga.n));
        }
        launch_gram_solve(c, ga);
        launch_combine(c, ga);
        for (int s : list) {
            slots[s].need_direction = false;
            slots[s].accept = false;
        }
    }

    // d = -H gp for the problems in `all` (lbfgs.c:571-598).  Gram mode: see directions_gram.
    // Two-loop mode: 1 + 2*bound fused launches per problem, issued together; a problem with a
    // shorter history starts later, so that all of them finish in the same launch (one X_DGI
    // exchange for everybody).
    void directions(BatchProblem* slots, const std::vector<int>& all) {
        if (all.empty()) return;
        std::vector<int> list = all;
        if (use_gram()) {     // first directions (d = -g) keep the plain path; the rest go through the Gram form
            std::vector<int> first, rest;
            for (int s : all) (slots[s].accept ? rest : first).push_back(s);
            if (!rest.empty()) directions_gram(slots, rest);
            if (first.empty()) return;
            list = first;
        }
        const int k = (int)list.size();
        const size_t g = (size_t)vec_grid(c);
        // commit the new pairs first
        PairArgs pa{};
        int np = 0;
        for (int a = 0; a < k; ++a) {
            const int s = list[a];
            BatchProblem& p = slots[s];
            if (!p.accept) continue;
            ProblemSlot& sl = c->slot[s];
            pa.x[np] = sl.x; pa.xp[np] = sl.xp; pa.g[np] = sl.g; pa.gp[np] = sl.gp;
            pa.s[np] = sl.S[p.end]; pa.y[np] = sl.Yh[p.end]; pa.xpos[np] = a;
            ++np;
        }
        if (np) {
            pa.n = np;
            launch_update_sy(c, pa, k);
            note(exchange(c, X_SY, 2 * k * g));
        }
        int order[kMaxBatch][kHistory];
        int maxb = 0;
        for (int a = 0; a < k; ++a) {
            BatchProblem& p = slots[list[a]];
            ProblemSlot& sl = c->slot[list[a]];
            if (p.accept) {   // the trial point becomes the accepted point
                std::swap(sl.x, sl.xp);
                std::swap(sl.g, sl.gp);
            }
            int j = (p.end + 1) % kHistory;
            for (int b = 0; b < p.bound; ++b) {
                j = (j + kHistory - 1) % kHistory;
                order[a][b] = j;   // newest -> oldest
            }
            maxb = std::max(maxb, p.bound);
        }
        const int nlaunch = 1 + 2 * maxb;
        for (int step = 0; step < nlaunch; ++step) {
            RecurArgs q{};
            q.n = k;
            for (int a = 0; a < k; ++a) {
                BatchProblem& p = slots[list[a]];
                ProblemSlot& sl = c->slot[list[a]];
                const int bound = p.bound;
                const int my = step - 2 * (maxb - bound);   // this problem's own step index
                q.d[a] = sl.d; q.gp[a] = sl.gp; q.scal[a] = sl.scal;
                q.mode[a] = -1;
                if (my < 0) continue;
                const double* vdot = nullptr;
                bool to_dginit = false;
                if (my == 0) {
                    q.mode[a] = 0;
                    q.hist[a] = p.end;
                    q.finalize_sy[a] = p.accept ? 1 : 0;
                    if (bound == 0) { vdot = sl.gp; to_dginit = true; }
                    else vdot = sl.S[order[a][0]];
                } else if (my <= bound) {            // first loop, newest -> oldest
                    const int b = my - 1;
                    const bool last = (b == bound - 1);
                    q.mode[a] = 1;
                    q.hist[a] = order[a][b];
                    q.vaxpy[a] = sl.Yh[order[a][b]];
                    q.scale[a] = last ? 1 : 0;
                    vdot = last ? sl.Yh[order[a][b]] : sl.S[order[a][b + 1]];
                } else {                             // second loop, oldest -> newest
                    const int b = bound - 1 - (my - 1 - bound);
                    const bool last = (b == 0);
                    q.mode[a] = 2;
                    q.hist[a] = order[a][b];
                    q.vaxpy[a] = sl.S[order[a][b]];
                    if (last) { vdot = sl.gp; to_dginit = true; }
                    else vdot = sl.Yh[order[a][b - 1]];
                }
                q.vdot[a] = vdot;
                q.to_dginit[a] = to_dginit ? 1 : 0;
            }
            launch_recur(c, q, step);
            if (step + 1 < nlaunch) note(exchange(c, (step & 1) ? X_REC1 : X_REC0, k * g));
            else note(exchange(c, X_DGI, k * g));
        }
        MVec8 sc{};
        for (int a = 0; a < k; ++a) sc.p[a] = c->slot[list[a]].scal;
        launch_store_dginit(c, k, sc);
        for (int s : list) {
            slots[s].need_direction = false;
            slots[s].accept = false;
        }
    }

    int run(int ntheta, const double* thetas, const double* g0_host, size_t g0_stride, const double* G_host,
            int max_batch, double* results, double* w_opt, bioen_opt_result* infos) {
        for (int i = 0; i < ntheta; ++i) std::memset(&infos[i], 0, sizeof(bioen_opt_result));
        LbfgsMachine probe((int)std::min<long long>(c->n_global, 0x7fffffff), cfg);
        const int bad = probe.validate();
        if (bad != 0) {   // liblbfgs rejects the parameters before touching x (lbfgs.c:285-331)
            for (int i = 0; i < ntheta; ++i) {
                infos[i].lbfgs_code = bad;
                std::memcpy(results + (size_t)i * c->n_global, g0_host + (size_t)i * g0_stride,
                            (size_t)c->n_global * sizeof(double));
            }
            return 0;
        }
        int kb = std::max(1, std::min(std::min(max_batch, kMaxBatch), ntheta));
        for (int s = 0; s < kb; ++s) note(alloc_slot(c, s, true));
        if (rc) return rc;
        note(upload_n(c, c->fixed, G_host));
        const bool shared_start = (g0_stride == 0) || ntheta == 1;
        if (shared_start) {
            if (!c->g0) note(dalloc_zero(&c->g0, c->ld, c->stream));
            if (rc) return rc;
            note(upload_n(c, c->g0, g0_host));
        }
        {   // log sum exp(G) once, written into every slot of the batch
            int all[kMaxBatch];
            for (int s = 0; s < kb; ++s) all[s] = s;
            const Round r = make_round(c, all, kb, nullptr, nullptr);
            if (c->world == 1) {
                launch_logw_logs0(c, r);
            } else {   // G is sharded on the device but whole on the host: same value on every rank
                const double v = host_logsumexp(G_host, c->n_global);
                for (int s = 0; s < kb; ++s)
                    note(hipMemcpyAsync(c->slot[s].scal + S_LOGS0, &v, sizeof(double), hipMemcpyHostToDevice, c->stream),
                         "logs0");
                note(hipStreamSynchronize(c->stream), "sync");
            }
        }

        BatchProblem slots[kMaxBatch];
        std::vector<LbfgsMachine> machines;
        machines.reserve(ntheta);
        for (int i = 0; i < ntheta; ++i) machines.emplace_back((int)std::min<long long>(c->n_global, 0x7fffffff), cfg);
        int next = 0, active = 0;
        bool occupied[kMaxBatch] = {};

        auto start_problem = [&](int s) {
            BatchProblem& p = slots[s];
            p = BatchProblem();
            p.id = next;
            p.theta = thetas[next];
            p.machine = &machines[next];
            p.t0 = std::chrono::steady_clock::now();
            ProblemSlot& sl = c->slot[s];
            if (shared_start)
                note(hipMemcpyAsync(sl.xp, c->g0, c->ld * sizeof(double), hipMemcpyDeviceToDevice, c->stream), "copy g0");
            else
                note(upload_n(c, sl.xp, g0_host + (size_t)next * g0_stride));
            note(hipMemsetAsync(sl.d, 0, c->ld * sizeof(double), c->stream), "memset d");
            note(hipMemsetAsync(sl.gram, 0, kGramStride * sizeof(double), c->stream), "memset gram");
            // the Gram sweep multiplies with every history buffer, live or not: leftovers of an earlier
            // run in this slot (possibly non-finite after a diverged one) must not reach 0 * x
            for (int i = 0; i < kHistory; ++i) {
                note(hipMemsetAsync(sl.S[i], 0, c->ld * sizeof(double), c->stream), "memset S");
                note(hipMemsetAsync(sl.Yh[i], 0, c->ld * sizeof(double), c->stream), "memset Y");
            }
            occupied[s] = true;
            ++active;
            ++next;
        };
        auto finish_problem = [&](int s, int code, bool keep_trial) {
            BatchProblem& p = slots[s];
            ProblemSlot& sl = c->slot[s];
            bioen_opt_result& info = infos[p.id];
            info.lbfgs_code = code;
            info.iterations = p.machine->iterations();
            info.evaluations = p.machine->evaluations();
            info.fmin = p.machine->fx();
            const double* res = keep_trial ? sl.x : sl.xp;
            if (!keep_trial && !p.initial) {
                // line search failed: liblbfgs returns the previous point; re-establish w, chi^2, KL there
                note(hipMemcpyAsync(sl.x, sl.xp, c->ld * sizeof(double), hipMemcpyDeviceToDevice, c->stream), "revert");
                const int one[1] = {s};
                const Round r = make_round(c, one, 1, nullptr, &p.theta);
                launch_max(c, r);
                note(enqueue_logw_eval(c, r, false));
                note(read_scalars(c, kMaxBatch));
                res = sl.x;
            }
            const double* h = c->host_scal + (size_t)s * kScalStride;
            info.chi2 = 0.5 * h[S_CHI];
            info.kl = h[S_P] - h[S_LOGS] + h[S_LOGS0];
            note(download_n(c, results + (size_t)p.id * c->n_global, res));
            if (w_opt) {
                const int one[1] = {s};
                launch_scale_w(c, make_round(c, one, 1, nullptr, &p.theta));   // e -> w, only now
                note(download_n(c, w_opt + (size_t)p.id * c->n_global, sl.w));
            }
            note(hipStreamSynchronize(c->stream), "sync");   // pageable destination: complete before the slot is reused
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
            // ---- one round: every active problem evaluates its next point -------------------------
            int list[kMaxBatch];
            double stp[kMaxBatch], th[kMaxBatch];
            int k = 0;
            for (int s = 0; s < kb; ++s) {
                if (!occupied[s]) continue;
                list[k] = s;
                stp[k] = slots[s].initial ? 0.0 : slots[s].machine->trial_step();
                th[k] = slots[s].theta;
                ++k;
            }
            const Round r = make_round(c, list, k, stp, th);
            launch_trial(c, r);
            note(enqueue_logw_eval(c, r, true));
            note(read_scalars(c, kMaxBatch));
            note(check_launch());
            if (rc) break;

            std::vector<int> dir_list;
            for (int a = 0; a < k; ++a) {
                const int s = list[a];
                BatchProblem& p = slots[s];
                ProblemSlot& sl = c->slot[s];
                const double* h = c->host_scal + (size_t)s * kScalStride;
                LbfgsMachine::Action act;
                if (p.initial) {
                    act = p.machine->on_initial(h[S_F], h[S_GG], h[S_XX]);
                    if (act.kind != LbfgsMachine::DONE) {
                        std::swap(sl.g, sl.gp);      // gradient at the accepted (= start) point
                        p.initial = false;
                        p.need_direction = true;
                        p.accept = false;
                        p.end = 0;
                        p.bound = 0;
                        dir_list.push_back(s);
                    }
                } else {
                    TrialResult t{h[S_F], h[S_DG], h[S_GG], h[S_XX], h[S_DGINIT]};
                    act = p.machine->on_trial(t);
                    if (act.kind == LbfgsMachine::ACCEPT) {
                        p.need_direction = true;
                        p.accept = true;
                        p.end = act.end;
                        p.bound = act.bound;
                        dir_list.push_back(s);
                    }
                }
                if (act.kind == LbfgsMachine::DONE) {
                    finish_problem(s, act.code, act.keep_trial);
                    if (next < ntheta && !rc) start_problem(s);
                }
            }
            directions(slots, dir_list);
        }
        note(hipStreamSynchronize(c->stream), "sync");
        note(check_launch());
        return rc;
    }
};

