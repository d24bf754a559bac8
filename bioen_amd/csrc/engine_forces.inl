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
    int last_dir = 0, prev_dir = 0;     // the running line search's last two moves: -1 step decreased, +1 increased, 0 none yet
    std::string moves;                  // BIOEN_HIP_SPEC_DEBUG: the running search's moves ('d' | 'i'; upper case = adopted shadow; '|' = a new round)
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
        // results published by a kernel into a coherent host page (see k_forces_publish); BIOEN_HIP_FORCES_LIVE=0 or a
        // failed allocation: the two device-to-host copies + stream synchronisation of r02
        const char* e = std::getenv("BIOEN_HIP_FORCES_LIVE");
        live = !(e && e[0] == '0') && !c->live_off;
        if (live && !c->live_f) {
            const size_t n = cnt + (size_t)kMaxBatch * kScalStride + 8;
            if (hipHostMalloc(reinterpret_cast<void**>(&c->live_f), n * sizeof(double),
                              hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
                (void)hipGetLastError();
                c->live_f = nullptr;
                live = false;
            } else {
                std::memset(c->live_f, 0, n * sizeof(double));
            }
        }
    }
    bool live = false;
    int jitter_us = std::getenv("BIOEN_HIP_JITTER_US") ? std::max(0, std::atoi(std::getenv("BIOEN_HIP_JITTER_US"))) : 0;   // tests
    // BIOEN_HIP_FORCES_TIMING=1: where the host's share of a round goes (sums in microseconds, printed at the end of run)
    bool timing = std::getenv("BIOEN_HIP_FORCES_TIMING") != nullptr;
    double t_pack = 0, t_copy = 0, t_enq = 0, t_wait = 0, t_decide = 0;
    long long t_rounds = 0;
    std::chrono::steady_clock::time_point t_mark;
    double lap() {
        const auto now = std::chrono::steady_clock::now();
        const double us = std::chrono::duration<double, std::micro>(now - t_mark).count();
        t_mark = now;
        return us;
    }

    // wait for round `round` of k_forces_publish; the stream is polled now and then so that a failed launch ends the
    // wait with its error instead of hanging the caller
    int await_page(unsigned long long round) {
        const size_t scal_at = (size_t)c->mp * kMaxBatch;
        const volatile unsigned long long* flag =
            reinterpret_cast<const volatile unsigned long long*>(c->live_f + scal_at + (size_t)kMaxBatch * kScalStride);
        BoundedWait w(c, "a round's results (forces engine)");
        while (*flag != round) {
            const int t = w.tick([&] { return *flag == round; });
            if (t < 0) return t;
            if (t > 0) break;
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        if (const int te = transport_error(c)) return te;      // results behind a failed exchange are not results (api.hip: await_live)
        std::memcpy(c->host_scal, c->live_f + scal_at, (size_t)kMaxBatch * kScalStride * sizeof(double));
        return 0;
    }

    void note(int e) { if (e && !rc) rc = e; }
    void note(hipError_t e, const char* what) { if (e != hipSuccess && !rc) rc = hip_fail(e, what, __FILE__, __LINE__); }

    // evaluate the K points pts[a] (each m long); f -> host_scal, gradients -> gm_h (compact)
    void evaluate(const int* slots, int k, const double* const* pts, const double* thetas, bool with_grad) {
        const int m = c->m;
        if (rc) return;
        if (timing) { t_decide += lap(); ++t_rounds; }
        std::fill(um_h, um_h + (size_t)c->mp * k, 0.0);
        for (int a = 0; a < k; ++a)
            for (int i = 0; i < m; ++i) um_h[(size_t)i * k + a] = pts[a][i];
        if (timing) t_pack += lap();
        note(hipMemcpyAsync(c->um, um_h, (size_t)c->mp * k * sizeof(double), hipMemcpyHostToDevice, c->stream),
             "forces H2D");
        if (timing) t_copy += lap();
        const ForcesRound fr = make_forces_round(c, slots, k, thetas);
        const Round r = make_round(c, slots, k, nullptr, thetas);
        note(enqueue_forces_eval(c, fr, r, with_grad));
        if (live && !rc) {
            const unsigned long long round = ++c->forces_round;
            launch_forces_publish(c, with_grad ? c->mp * k : 0, round);
            note(check_launch());
            if (timing) t_enq += lap();
            if (!rc) note(await_page(round));
            if (timing) t_wait += lap();
            gm_h = c->live_f;                       // compact [row * k + problem], as c->gm
            return;
        }
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
        // Speculative line-search trials (r03; the log-weights engines have had them since r02): a backtracking search
        // moves its step by fixed factors (x 0.5 after a failed decrease test, x 2.1 after a failed curvature test,
        // lbfgs.c:686-727), so the steps it may ask for next are known before the trial returns.  Batch slots without
        // a problem evaluate them alongside the trial -- the matrix passes are shared, a column more costs little --
        // and a rejected trial finds its successor (and that one's successor ...) already evaluated.  The decisions,
        // their order and every number they see are those of the serial search: results do not change by a bit
        // (tests: BIOEN_HIP_SPECULATE=0 against the default).  A warm-started series at small N (the ala5 protocol:
        // 5.7 evaluations per iteration) is where this pays; on big matrices a wider batch costs real time and only
        // searches that have shown a rejection rate get shadows.
        bool spec = cfg.linesearch >= 1 && cfg.linesearch <= 3;
        if (const char* e = std::getenv("BIOEN_HIP_SPECULATE")) spec = spec && std::atoi(e) != 0;
        const int nslots = spec ? std::min(std::max(max_batch, 1), kMaxBatch) : kb;      // max_batch bounds the batch WIDTH
        const bool reuse_wanted = !std::getenv("BIOEN_HIP_FORCES_REEVALUATE");     // A/B: evaluate the returned point again, as until r03
        const bool big = 2.0 * c->mp * (double)c->ld * sizeof(double) > 1e9;             // a column more is not free there
        for (int s = 0; s < nslots; ++s) note(alloc_slot(c, s, false));
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
        // evalslot >= 0: the returned point is the one slot `evalslot` (column evalcol of the last round) has just
        // evaluated -- its scalars are on the host, x_j and the normalisation still in the slot: the weights come from
        // one N-vector kernel instead of another evaluation (the strip passes; elsewhere the point is evaluated again)
        auto finish_problem = [&](int s, int code, bool keep_trial, int evalslot, int evalcol) {
            ForcesProblem& p = probs[s];
            bioen_opt_result& info = infos[p.id];
            info.lbfgs_code = code;
            info.iterations = p.machine->iterations();
            info.evaluations = p.machine->evaluations();
            info.fmin = p.machine->fx();
            const std::vector<double>& res = keep_trial ? p.x : p.xp;
            std::memcpy(results + (size_t)p.id * m, res.data(), (size_t)m * sizeof(double));
            // weights, chi^2 and KL at the returned forces (forces.py:535-548 recomputes them too)
            int ws = s;
            if (evalslot >= 0 && reuse_wanted && c->Ys && forces_fused_blocks(c) > 0) {   // the last round ran the strip passes
                ws = evalslot;
                c->last_pos = evalcol;                    // bioen_hip_last_average: the column of that round's ybar_c
                if (w_opt) {
                    const int one[1] = {ws};
                    launch_forces_w_from_x(c, make_forces_round(c, one, 1, &p.theta));
                    note(check_launch());
                }
            } else {
                const int one[1] = {s};
                const double* pt[1] = {res.data()};
                evaluate(one, 1, pt, &p.theta, false);
            }
            const double* h = c->host_scal + (size_t)ws * kScalStride;
            info.chi2 = 0.5 * h[S_CHI];
            info.kl = h[S_KL];
            if (w_opt) {
                note(download_n(c, w_opt + (size_t)p.id * c->n_global, c->slot[ws].w));   // gathers the ranks' blocks
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

        struct Shadow {
            int owner;                  // slot of the problem it works for
            int slot;                   // free batch slot that evaluates it
            double stp;                 // formed as report_backtracking would form it: parent * 0.5 | parent * 2.1
            int col;                    // column of the round
            std::vector<double> x;
        };
        std::vector<Shadow> shadows;
        long long issued = 0, adopted = 0;
        const bool dbg = std::getenv("BIOEN_HIP_SPEC_DEBUG") != nullptr;

        for (int s = 0; s < kb && next < ntheta; ++s) start_problem(s);
        t_mark = std::chrono::steady_clock::now();
        while (active > 0 && !rc) {
            if (jitter_us > 0) std::this_thread::sleep_for(std::chrono::microseconds((unsigned)std::rand() % (unsigned)jitter_us));
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
            const int nown = k;
            // ---- shadows: the free slots go round the searching problems, each taking the next step of its candidate
            // order -- the chain of the search's last move first (a run of halvings is the common case), the other
            // move and the mixed step (x 0.5 x 2.1, the same number either way round) behind it
            shadows.clear();
            if (spec && nown < nslots) {
                int freeslot[kMaxBatch], nfree = 0;
                for (int s = 0; s < nslots; ++s)
                    if (s >= kb || !occupied[s]) freeslot[nfree++] = s;
                struct Plan { int owner; double cand[kMaxBatch]; int n, taken; bool keen; };
                Plan plan[kMaxBatch];
                int nplan = 0;
                for (int a = 0; a < nown; ++a) {
                    ForcesProblem& p = probs[list[a]];
                    if (p.initial) continue;
                    const long long ev = p.machine->evaluations(), it = p.machine->iterations();
                    Plan& q = plan[nplan++];
                    q.owner = list[a];
                    // big matrices: the strip passes take the same time for 1..4 columns (0.60 ms at N = 1e6 x M = 512)
                    // and 25 % more for 8: a search gets more than a free column only if a quarter of the problem's
                    // trials have been rejected so far
                    q.keen = !big || (ev >= 16 && (ev - it) * 4 >= ev);
                    q.n = q.taken = 0;
                    const double stp = p.machine->trial_step();
                    const bool wolfe = cfg.linesearch >= 2;
                    // The main chain follows the search's habit -- a run of one move goes on (halvings from a step far
                    // too long: the first search of a warm start), two different moves in a row alternate (a search
                    // caught between the decrease and the curvature test: x 0.5, x 2.1, x 0.5 ... until max_linesearch)
                    // -- six steps deep; the other move at its first node takes the slot left.
                    auto other = [](char mv) { return mv == 'd' ? 'i' : 'd'; };
                    auto apply = [](double v, char mv) { return mv == 'd' ? v * 0.5 : v * 2.1; };
                    char h1 = p.last_dir == 0 ? 0 : (p.last_dir < 0 ? 'd' : 'i'), h2 = p.prev_dir == 0 ? 0 : (p.prev_dir < 0 ? 'd' : 'i');
                    double v = stp, node[1] = {stp};
                    char nodemove[1] = {'d'};
                    for (int depth = 0; depth < (wolfe ? 6 : kMaxBatch - 1); ++depth) {
                        char mv = 'd';
                        if (wolfe && h1) mv = (h2 && h1 != h2) ? other(h1) : h1;
                        if (depth < 1) {
                            node[depth] = v;
                            nodemove[depth] = mv;
                        }
                        v = apply(v, mv);
                        if (q.n < kMaxBatch - 1) q.cand[q.n++] = v;
                        h2 = h1;
                        h1 = mv;
                    }
                    if (wolfe) {
                        // order: chain[0], the other move at the first node, chain[1..5]
                        const double s0 = apply(node[0], other(nodemove[0]));
                        double ordered[kMaxBatch] = {q.cand[0], s0, q.cand[1], q.cand[2], q.cand[3], q.cand[4], q.cand[5]};
                        q.n = 7;
                        for (int i = 0; i < q.n; ++i) q.cand[i] = ordered[i];
                    }
                }
                while (nfree > 0 && nplan > 0) {
                    bool any = false;
                    for (int q = 0; q < nplan && nfree > 0; ++q) {
                        Plan& pl = plan[q];
                        if (pl.taken >= pl.n || (!pl.keen && k >= 4)) continue;
                        const double stp = pl.cand[pl.taken++];
                        any = true;
                        if (!(stp >= kMinStep && stp <= kMaxStep)) continue;        // the search would end there (lbfgs.c:718-725)
                        ForcesProblem& p = probs[pl.owner];
                        Shadow sh;
                        sh.owner = pl.owner;
                        sh.slot = freeslot[--nfree];
                        sh.stp = stp;
                        sh.col = k;
                        sh.x.resize(m);
                        for (int i = 0; i < m; ++i) sh.x[i] = p.xp[i] + stp * p.d[i];
                        shadows.push_back(std::move(sh));
                        list[k] = shadows.back().slot;
                        th[k] = p.theta;
                        ++k;
                    }
                    if (!any) break;
                }
                for (const Shadow& sh : shadows) pts[sh.col] = sh.x.data();           // after the vector stopped growing
                issued += (long long)shadows.size();
            }
            evaluate(list, k, pts, th, true);
            if (rc) break;
            for (int a = 0; a < nown; ++a) {
                const int s = list[a];
                ForcesProblem& p = probs[s];
                const double f = c->host_scal[(size_t)s * kScalStride + S_F];
                std::vector<double>& grad = p.initial ? p.gp : p.g;
                for (int i = 0; i < m; ++i) grad[i] = gm_h[(size_t)i * k + a];
                LbfgsMachine::Action act;
                int evalslot = s, evalcol = a;
                if (p.initial) {
                    act = p.machine->on_initial(f, ForcesProblem::dot(p.gp, p.gp), ForcesProblem::dot(p.xp, p.xp));
                    if (act.kind != LbfgsMachine::DONE) {
                        for (int i = 0; i < m; ++i) p.d[i] = -p.gp[i];
                        p.initial = false;
                        p.last_dir = p.prev_dir = 0;
                    }
                } else {
                    double tried = p.machine->trial_step();
                    if (dbg) p.moves += '|';
                    TrialResult t{f, ForcesProblem::dot(p.g, p.d), ForcesProblem::dot(p.g, p.g),
                                  ForcesProblem::dot(p.x, p.x), ForcesProblem::dot(p.gp, p.d)};
                    act = p.machine->on_trial(t);
                    // the search asks for another step: if a shadow of this round sat exactly there, its values ARE that
                    // trial's -- feed them in, and so on down the chain
                    while (act.kind == LbfgsMachine::TRIAL) {
                        const double want = p.machine->trial_step();
                        p.prev_dir = p.last_dir;
                        p.last_dir = want < tried ? -1 : 1;
                        if (dbg) p.moves += p.last_dir < 0 ? 'd' : 'i';
                        const Shadow* hit = nullptr;
                        for (const Shadow& sh : shadows)
                            if (sh.owner == s && sh.stp == want) { hit = &sh; break; }
                        if (!hit) break;
                        if (dbg) p.moves.back() = (char)std::toupper(p.moves.back());
                        ++adopted;
                        tried = want;
                        evalslot = hit->slot;
                        evalcol = hit->col;
                        p.x = hit->x;
                        for (int i = 0; i < m; ++i) p.g[i] = gm_h[(size_t)i * k + hit->col];
                        const double fs = c->host_scal[(size_t)hit->slot * kScalStride + S_F];
                        TrialResult ts{fs, ForcesProblem::dot(p.g, p.d), ForcesProblem::dot(p.g, p.g),
                                       ForcesProblem::dot(p.x, p.x), ForcesProblem::dot(p.gp, p.d)};
                        act = p.machine->on_trial(ts);
                    }
                    if (act.kind == LbfgsMachine::ACCEPT) {
                        p.accept(act.end, act.bound);
                        p.last_dir = p.prev_dir = 0;
                    }
                    if (dbg && act.kind != LbfgsMachine::TRIAL) {
                        std::fprintf(stderr, "theta %g search %s %s\n", p.theta, p.moves.c_str(), act.kind == LbfgsMachine::ACCEPT ? "accepted" : "done");
                        p.moves.clear();
                    }
                }
                if (act.kind == LbfgsMachine::DONE) {
                    // the point handed out was evaluated in THIS round if it is the trial (keep_trial) or the start
                    const bool fresh = p.initial || act.keep_trial;
                    finish_problem(s, act.code, act.keep_trial && !p.initial, fresh ? evalslot : -1, evalcol);
                    if (next < ntheta && !rc) start_problem(s);
                }
            }
        }
        c->spec_launched += issued;
        c->spec_used += adopted;
        if (timing && t_rounds)
            std::fprintf(stderr, "forces engine, host side per round (us, %lld rounds): decisions + points %.1f, packing %.1f, H2D call %.1f, "
                         "launches %.1f, waiting for the device %.1f\n", t_rounds, t_decide / t_rounds, t_pack / t_rounds,
                         t_copy / t_rounds, t_enq / t_rounds, t_wait / t_rounds);
        if (verbose && issued)
            std::printf("\tspeculative line-search evaluations: %lld issued in idle batch slots, %lld adopted\n", issued, adopted);
        note(hipStreamSynchronize(c->stream), "sync");
        note(transport_error(c));              // no exchange of the run may have failed (api.hip: await_live)
        return rc;
    }
};

