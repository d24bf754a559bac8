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
    bool speculate = true;
    int max_shadows = 2;                             // shadow evaluations per round at most (BIOEN_HIP_HOST_SHADOWS): both
                                                     // steps of the slowest problem (r02: every idle slot)
    bool shadow_mixed_first = false;                 // third candidate of a search: the up-then-down step instead of stp / 4
    long long spec_launched = 0, spec_used = 0;      // shadow evaluations issued / adopted

    // Deliveries: the optimum and the weights of a finished problem (2 N doubles into the caller's pageable arrays,
    // 1.5 ms at N = 1e6) leave on a second stream, driven by a helper thread that blocks in the copy while this
    // thread goes on launching rounds for the other problems.  The slot's vectors stay untouched until its delivery
    // is done (no shadow evaluations in it, start_problem waits).  Unsharded contexts only; BIOEN_HIP_DELIVERY=0
    // copies on the compute stream as before.
    struct Delivery {
        std::thread th;
        std::atomic<int> done{0};
        int rc = 0;
    };
    std::unique_ptr<Delivery> pending[kMaxBatch];
    bool async_delivery = true;
    int jitter_us = 0, jitter_delivery_us = 0;   // BIOEN_HIP_JITTER_US / BIOEN_HIP_JITTER_DELIVERY_US (tests): random pauses of
                                                 // this rank's host thread in every round / of its delivery threads before they report

    LogwBatchEngine(bioen_hip_ctx* ctx, const bioen_lbfgs_config& config, bool verb)
        : c(ctx), cfg(config), verbose(verb) {
        const char* e = std::getenv("BIOEN_HIP_SPECULATE");
        speculate = !(e && e[0] == '0');
        if (const char* m = std::getenv("BIOEN_HIP_HOST_SHADOWS")) max_shadows = std::max(0, std::min((int)kMaxBatch, std::atoi(m)));
        if (const char* m = std::getenv("BIOEN_HIP_SHADOW_MIXED")) shadow_mixed_first = m[0] == '1';
        const char* d = std::getenv("BIOEN_HIP_DELIVERY");
        async_delivery = !(d && d[0] == '0');
        if (const char* j = std::getenv("BIOEN_HIP_JITTER_US")) jitter_us = std::max(0, std::atoi(j));
        if (const char* j = std::getenv("BIOEN_HIP_JITTER_DELIVERY_US")) jitter_delivery_us = std::max(0, std::atoi(j));
    }
    ~LogwBatchEngine() {
        for (int s = 0; s < kMaxBatch; ++s) {
            settle(s);
            if (gather[s]) (void)hipFree(gather[s]);
        }
    }

    bool slot_busy(int s) const { return pending[s] && !pending[s]->done.load(std::memory_order_acquire); }
    // May slot s evaluate a speculative trial now?  Unsharded: not while its delivery still reads the slot's vectors.  On a
    // sharded context a delivery reads the slot's own gather buffer (deliver_sharded: the vectors are free at once) -- and
    // the answer MUST NOT depend on how far a helper thread has come: every rank has to compose the same round (the stage
    // exchanges carry payloads that depend on the batch width).  Until r04 this asked slot_busy on every context: two
    // ranks that disagreed on a delivery in flight enqueued rounds of different width and the next exchange failed
    // (seen with a problem that ends in its first round while shadows want its slot; at scale any finishing theta).
    // tests (tests/test_hip_nshard.py): the ranks of a sharded run dawdle at random -- their host threads in every round,
    // ONE rank's delivery threads before they report -- and must still compose the same rounds and land on the same bits
    static void jitter_pause(int us, unsigned salt) {
        if (us <= 0) return;
        static std::atomic<unsigned> ctr{12345};
        unsigned v = ctr.fetch_add(2654435761u) ^ (salt * 40503u);
        v ^= v >> 13; v *= 0x5bd1e995u; v ^= v >> 15;
        std::this_thread::sleep_for(std::chrono::microseconds(v % (unsigned)us));
    }
    void jitter(unsigned salt = 0) const { jitter_pause(jitter_us, salt); }
    bool slot_blocked(int s) const { return c->world == 1 && slot_busy(s); }
    // the order of the thetas as a strict weak ordering whatever the caller passed: NaN sorts last (`<` alone is no
    // ordering with NaN in the series, and a comparator that is none is undefined behaviour in std::sort)
    static bool theta_before(double a, double b) {
        if (a != a) return false;
        if (b != b) return true;
        return a < b;
    }
    void settle(int s) {
        if (!pending[s]) return;
        if (pending[s]->th.joinable()) pending[s]->th.join();
        note(pending[s]->rc);
        pending[s].reset();
    }
    // x (and w) of slot s -> the caller's arrays, behind everything enqueued on the compute stream so far
    void deliver(int s, double* dst_x, const double* src_x, double* dst_w, const double* src_w) {
        settle(s);
        if (!c->copy_stream) note(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking), "hipStreamCreate");
        hipEvent_t ev = nullptr;
        note(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
        if (!rc) note(hipEventRecord(ev, c->stream), "hipEventRecord");
        if (rc) {
            if (ev) (void)hipEventDestroy(ev);
            return;
        }
        Delivery* d = new Delivery;
        pending[s].reset(d);
        const int dev = c->device, jit = jitter_delivery_us;
        hipStream_t cs = c->copy_stream;
        const size_t bytes = (size_t)c->n * sizeof(double);
        auto work = [=]() {
            hipError_t e = hipSetDevice(dev);
            if (e == hipSuccess) e = hipStreamWaitEvent(cs, ev, 0);
            if (e == hipSuccess) e = d2h_user(cs, dst_x, src_x, bytes);          // (api.hip: a caller's buffer the runtime will not pin is staged)
            if (e == hipSuccess && dst_w) e = d2h_user(cs, dst_w, src_w, bytes);
            if (e == hipSuccess) e = hipStreamSynchronize(cs);
            (void)hipEventDestroy(ev);
            d->rc = e == hipSuccess ? 0 : BIOEN_HIP_EHIP;
            jitter_pause(jit, 7);
            d->done.store(1, std::memory_order_release);
        };
        try {
            d->th = std::thread(work);
        } catch (...) {            // no thread to be had: copy here, on the second stream all the same
            work();
            note(d->rc);
            pending[s].reset();
        }
    }

    // The same for a structure-sharded context (r04): every rank returns GLOBAL vectors, so a result is gathered first.
    // The gather (one X_VEC stage exchange per vector, into a buffer of the slot's own instead of the shared stage
    // buffer) stays on the compute stream, in the order every rank issues it; the world x 2 copies into the caller's
    // pageable arrays -- 16 MB per theta at the headline, a millisecond the rounds of the other thetas need not wait
    // for -- leave on the second stream as above.  The slot's vectors are free as soon as the gather is queued.
    void deliver_sharded(int s, double* dst_x, const double* src_x, double* dst_w, const double* src_w) {
        settle(s);
        const size_t per = c->ld * (size_t)c->world;
        if (!gather[s]) note(dalloc_zero(&gather[s], 2 * per, c->stream));
        if (!c->copy_stream) note(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking), "hipStreamCreate");
        if (rc) return;
        double* const stage = c->xbuf[X_VEC];
        const double* srcs[2] = {src_x, dst_w ? src_w : nullptr};
        for (int v = 0; v < 2 && !rc; ++v) {
            if (!srcs[v]) continue;
            double* base = gather[s] + (size_t)v * per;
            note(hipMemcpyAsync(base + (size_t)c->rank * c->ld, srcs[v], c->ld * sizeof(double), hipMemcpyDeviceToDevice,
                                c->stream), "gather: own segment");
            c->xbuf[X_VEC] = base;                     // the exchange works in place on the stage buffer it is handed
            note(exchange_raw(c, X_VEC, c->ld));
            c->xbuf[X_VEC] = stage;
        }
        hipEvent_t ev = nullptr;
        note(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
        if (!rc) note(hipEventRecord(ev, c->stream), "hipEventRecord");
        if (rc) {
            if (ev) (void)hipEventDestroy(ev);
            return;
        }
        Delivery* d = new Delivery;
        pending[s].reset(d);
        const int dev = c->device, world = c->world, jit = jitter_delivery_us;
        hipStream_t cs = c->copy_stream;
        const size_t ld = c->ld;
        const bioen_hip_ctx* const cc = c;
        double* const gb = gather[s];
        auto work = [=]() {
            hipError_t e = hipSetDevice(dev);
            if (e == hipSuccess) e = hipStreamWaitEvent(cs, ev, 0);
            double* dsts[2] = {dst_x, dst_w};
            for (int v = 0; v < 2 && e == hipSuccess; ++v) {
                if (!dsts[v]) continue;
                for (int r = 0; r < world && e == hipSuccess; ++r) {
                    long long col0, nl;
                    rank_columns(cc, r, &col0, &nl);
                    if (nl > 0)
                        e = d2h_user(cs, dsts[v] + col0, gb + (size_t)v * ld * world + (size_t)r * ld, (size_t)nl * sizeof(double));
                }
            }
            if (e == hipSuccess) e = hipStreamSynchronize(cs);
            (void)hipEventDestroy(ev);
            d->rc = e == hipSuccess ? 0 : BIOEN_HIP_EHIP;
            jitter_pause(jit, 7);
            d->done.store(1, std::memory_order_release);
        };
        try {
            d->th = std::thread(work);
        } catch (...) {
            work();
            note(d->rc);
            pending[s].reset();
        }
    }
    double* gather[kMaxBatch] = {};       // per slot: [2][world][ld] results on their way out (sharded contexts)

    void note(int e) { if (e && !rc) rc = e; }
    void note(hipError_t e, const char* what) { if (e != hipSuccess && !rc) rc = hip_fail(e, what, __FILE__, __LINE__); }

    bool use_gram() const {
        return c->direction_mode != 1;   // auto = Gram form
    }

    // device-resident line-search decisions (engine_devls.inl)
    bool device_engine_applies() const;
    int ensure_device_state();
    int await_flight(const struct DevFlight& f);
    int run_device(int ntheta, const double* thetas, const double* g0_host, size_t g0_stride, const double* G_host,
                   int max_batch, double* results, double* w_opt, bioen_opt_result* infos);

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
        launch_gram_rank_reduce(c, ga.n);         // the local segments' totals: 39 per problem and segment (sharded: what is shipped)
        note(exchange(c, X_GRAMR, (size_t)kGramDots * ga.n));
        launch_gram_solve(c, ga);                 // ... added in segment order, on one GPU as on eight
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
        if (device_engine_applies()) {
            const int e = ensure_device_state();      // a context without a coherent host page falls back to the host-driven engine
            if (!e) return run_device(ntheta, thetas, g0_host, g0_stride, G_host, max_batch, results, w_opt, infos);
            if (!c->live_off) return e;
        }
        int kb = std::max(1, std::min(std::min(max_batch, kMaxBatch), ntheta));
        // Problems start in ascending theta (the slow ones first: the series ends with its slowest member).  A series
        // that fills the batch exactly leaves no slot to speculate in until its first member finishes -- and a line
        // search rejects most in its first iterations (the headline's slowest theta: 19 of 39 rejected trials in the
        // first 13 rounds).  Such a series keeps two slots back for shadows of the slowest problem from the first
        // round on; the two fastest thetas wait for the first slots to come free (they need a fraction of the rounds).
        std::vector<int> start_order(ntheta);
        for (int i = 0; i < ntheta; ++i) start_order[i] = i;
        std::stable_sort(start_order.begin(), start_order.end(), [&](int x, int y) { return theta_before(thetas[x], thetas[y]); });
        const bool backtracking = cfg.linesearch >= 1 && cfg.linesearch <= 3;
        if (speculate && backtracking && max_shadows >= 2 && kb == kMaxBatch && ntheta <= kMaxBatch) {
            const char* e = std::getenv("BIOEN_HIP_RESERVE");
            kb -= e ? std::max(0, std::min(4, std::atoi(e))) : 2;
        }
        for (int s = 0; s < kb; ++s) note(alloc_slot(c, s, true));
        if (rc) return rc;
        // cache policy of the round's N-vector kernels (kernels_logw.hip): nontemporal history loads and outputs once the
        // batch's vectors (19 per problem) cannot stay in the 256 MB of Infinity Cache anyway
        c->nvec_nt = c->nvec_nt_env >= 0 ? c->nvec_nt_env == 1 : (double)c->ld * sizeof(double) * 19.0 * kb > 256.0 * 1024 * 1024;
        note(upload_n(c, c->fixed, G_host));
        const bool shared_start = (g0_stride == 0) || ntheta == 1;
        if (shared_start) {
            if (!c->g0) note(dalloc_zero(&c->g0, c->ld, c->stream));
            if (rc) return rc;
            note(upload_n(c, c->g0, g0_host));
        }
        // Idle batch slots evaluate the steps a backtracking line search may ask for next (see the round loop):
        // they need the N-vectors of an evaluation (x, g, w, a), no history
        const int nslots = speculate && cfg.linesearch >= 1 && cfg.linesearch <= 3 ? kMaxBatch : kb;
        for (int s = kb; s < nslots; ++s) note(alloc_slot(c, s, false));
        if (rc) return rc;
        {   // log sum exp(G) once, written into every slot of the batch
            int all[kMaxBatch];
            for (int s = 0; s < nslots; ++s) all[s] = s;
            const Round r = make_round(c, all, nslots, nullptr, nullptr);
            note(enqueue_logs0(c, r));       // sharded: the ranks' block pairs are exchanged, every rank merges the same numbers
        }

        const bool dbg = std::getenv("BIOEN_HIP_SPEC_DEBUG") != nullptr;
        std::vector<int> dbg_rej(ntheta, 0), dbg_noslot(ntheta, 0), dbg_miss(ntheta, 0);
        BatchProblem slots[kMaxBatch];
        std::vector<LbfgsMachine> machines;
        machines.reserve(ntheta);
        for (int i = 0; i < ntheta; ++i) machines.emplace_back((int)std::min<long long>(c->n_global, 0x7fffffff), cfg);
        int next = 0, active = 0;
        bool occupied[kMaxBatch] = {};

        auto start_problem = [&](int s) {
            BatchProblem& p = slots[s];
            p = BatchProblem();
            const int id = start_order[next];
            p.id = id;
            p.theta = thetas[id];
            p.machine = &machines[id];
            p.t0 = std::chrono::steady_clock::now();
            ProblemSlot& sl = c->slot[s];
            settle(s);                               // the previous tenant's results have left
            if (shared_start)
                note(hipMemcpyAsync(sl.xp, c->g0, c->ld * sizeof(double), hipMemcpyDeviceToDevice, c->stream), "copy g0");
            else
                note(upload_n(c, sl.xp, g0_host + (size_t)id * g0_stride));
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
        auto finish_problem = [&](int s, int code, bool keep_trial, int column) {
            BatchProblem& p = slots[s];
            ProblemSlot& sl = c->slot[s];
            c->last_pos = column;                    // where this problem's averages sit in ybar_c (bioen_hip_last_average)
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
            if (w_opt) {
                const int one[1] = {s};
                launch_scale_w(c, make_round(c, one, 1, nullptr, &p.theta));   // e -> w, only now
            }
            if (async_delivery && c->world > 1) {
                deliver_sharded(s, results + (size_t)p.id * c->n_global, res,
                                w_opt ? w_opt + (size_t)p.id * c->n_global : nullptr, sl.w);
            } else if (async_delivery) {
                deliver(s, results + (size_t)p.id * c->n_global, res, w_opt ? w_opt + (size_t)p.id * c->n_global : nullptr,
                        sl.w);
            } else {
                note(download_n(c, results + (size_t)p.id * c->n_global, res));
                if (w_opt) note(download_n(c, w_opt + (size_t)p.id * c->n_global, sl.w));
                note(hipStreamSynchronize(c->stream), "sync");   // pageable destination: complete before the slot is reused
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

        // The evaluation-owned entries of a slot's scalars (the rest -- y.s, alpha, gp.d -- belongs to the problem)
        static const int kEvalScal[][2] = {{S_F, 4}, {S_LOGS, 4}, {S_KL, 2}, {S_INV, kMaxSeg + 2}};      // (S_INV[kMaxSeg], S_B0, S_UY)

        while (active > 0 && !rc) {
            jitter(1);
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
            // Speculation: a backtracking search that rejects its trial asks for stp/2 or 2.1 stp next -- known
            // now.  Slots no problem occupies evaluate those points in the SAME matrix passes (their cost does
            // not depend on the batch width); if the search then asks for one of them, its evaluation is already
            // there.  Same point, same kernels, same order of operations as the round that is saved: results
            // do not change by a bit (tests: BIOEN_HIP_SPECULATE=0 against the default).
            int shadow_owner[kMaxBatch], shadow_slot[kMaxBatch];
            double shadow_stp[kMaxBatch];
            int nshadow = 0;
            if (speculate && (nslots > kb || active < kb)) {
                int free_slots[kMaxBatch], nfree = 0;
                for (int s = 0; s < nslots; ++s)
                    if ((s >= kb || !occupied[s]) && !slot_blocked(s)) free_slots[nfree++] = s;
                // Who gets the idle slots: the series ends when its SLOWEST member does, so a saved evaluation shortens it
                // only on that member's path -- the smallest theta in every series measured (r03; r02 dealt stp / 2 to
                // everybody first and reached the straggler last: 22 of its 45 rejected trials saved at the headline).
                // Owners in ascending theta, both steps each, while slots last.
                int order[kMaxBatch];
                for (int a = 0; a < k; ++a) order[a] = a;
                std::sort(order, order + k, [&](int x, int y) { return theta_before(th[x], th[y]); });
                for (int i = 0; i < k && nfree > 0 && nshadow < max_shadows; ++i) {
                    const int a = order[i];
                    BatchProblem& p = slots[list[a]];
                    // The steps the search can ask for after this trial -- formed exactly as report_backtracking forms
                    // them (lbfgs.c:686-727): stp * 0.5 | stp * 2.1 -- and, should slots remain, the ones a rejected
                    // successor would ask for.  (Measured at the headline, r03: the two first-level steps for the slowest
                    // theta catch every first rejection, 29 of its 45 extra evaluations.  The other 16 are second rejections
                    // inside one search -- r04, BIOEN_HIP_SPEC_DEBUG: 4 x "up again" in the first search, 6 x "up, then down"
                    // in one ping-pong search around evaluation 20, 6 x "down again" -- all but one within the first 36
                    // evaluations, when the two reserved slots are the only free ones: a third candidate would need a third
                    // theta to wait, for at most 5-6 of 406 rounds.)
                    double c1[2], cand[5];
                    int nc = p.initial ? 0 : p.machine->speculative_steps(c1);
                    for (int i2 = 0; i2 < nc; ++i2) cand[i2] = c1[i2];
                    if (nc == 2) {
                        cand[2] = c1[0] * 0.5;
                        cand[3] = c1[1] * 2.1;
                        cand[4] = c1[1] * 0.5;           // up, then down (= down, then up: the halving is exact)
                        nc = 5;
                        if (shadow_mixed_first) std::swap(cand[2], cand[4]);
                    } else if (nc == 1) {
                        cand[1] = c1[0] * 0.5;
                        nc = 2;
                    }
                    for (int pass = 0; pass < nc && nfree > 0 && nshadow < max_shadows; ++pass) {
                        note(alloc_slot(c, free_slots[nfree - 1], false));
                        shadow_owner[nshadow] = a;
                        shadow_slot[nshadow] = free_slots[--nfree];
                        shadow_stp[nshadow] = cand[pass];
                        ++nshadow;
                    }
                }
            }
            Round r = make_round(c, list, k, stp, th);
            for (int q = 0; q < nshadow; ++q) {
                const int a = k + q;
                const ProblemSlot& own = c->slot[list[shadow_owner[q]]];
                const ProblemSlot& sh = c->slot[shadow_slot[q]];
                r.x[a] = sh.x; r.g[a] = sh.g; r.w[a] = sh.w; r.a[a] = sh.a; r.scal[a] = sh.scal; r.part[a] = sh.part;
                r.xp[a] = own.xp; r.gp[a] = own.gp; r.d[a] = own.d;
                r.stp[a] = shadow_stp[q];
                r.theta[a] = th[shadow_owner[q]];
            }
            r.n = k + nshadow;
            spec_launched += nshadow;
            launch_trial(c, r);
            if (!c->live_off) {
                // the round's scalars come back through the host-mapped page (ctx.hpp: live); positions k.. are shadows
                int where[kMaxBatch];
                for (int a = 0; a < k; ++a) where[a] = list[a];
                for (int q = 0; q < nshadow; ++q) where[k + q] = shadow_slot[q];
                const unsigned long long round = c->live_round = ++c->live_seq;
                note(enqueue_logw_eval(c, r, true));
                note(check_launch());
                if (!rc) note(await_live(c, round, where, r.n));
            } else {
                note(enqueue_logw_eval(c, r, true));
                note(read_scalars(c, kMaxBatch));
                note(check_launch());
            }
            if (rc) {
                c->live_round = 0;        // a round that failed before its last kernel must not lend its number to the next one
                break;
            }

            std::vector<int> dir_list;
            for (int a = 0; a < k; ++a) {
                const int s = list[a];
                BatchProblem& p = slots[s];
                ProblemSlot& sl = c->slot[s];
                const double* h = c->host_scal + (size_t)s * kScalStride;
                int column = a;                      // of the evaluation the problem ends up with
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
                    if (act.kind == LbfgsMachine::TRIAL && dbg) {
                        int mine = 0, hit = 0;
                        for (int q = 0; q < nshadow; ++q)
                            if (shadow_owner[q] == a) { ++mine; hit += shadow_stp[q] == p.machine->trial_step(); }
                        ++dbg_rej[p.id];
                        if (!mine) ++dbg_noslot[p.id]; else if (!hit) ++dbg_miss[p.id];
                        if (!hit) {
                            std::fprintf(stderr, "spec debug: theta %g evaluation %d: trial %.17g rejected, asks %.17g (x %.3g); shadows:",
                                         p.theta, p.machine->evaluations(), stp[a], p.machine->trial_step(), p.machine->trial_step() / stp[a]);
                            for (int q = 0; q < nshadow; ++q)
                                if (shadow_owner[q] == a) std::fprintf(stderr, " x %.3g", shadow_stp[q] / stp[a]);
                            std::fprintf(stderr, "\n");
                        }
                    }
                    // rejected: is the step it asks for next among this round's shadows?  An adopted evaluation may be
                    // rejected in its turn: its successor may be there too (second-level shadows)
                    bool used[kMaxBatch] = {};
                    double prev = 0.0;
                    while (act.kind == LbfgsMachine::TRIAL) {
                        prev = p.machine->trial_step();
                        int q = 0;
                        for (; q < nshadow; ++q)
                            if (shadow_owner[q] == a && !used[q] && shadow_stp[q] == prev) break;
                        if (q == nshadow) {
                            if (dbg && column != a)
                                std::fprintf(stderr, "spec debug: theta %g evaluation %d: adopted x %.3g rejected in its turn, asks x %.4g\n",
                                             p.theta, p.machine->evaluations(), c->host_scal[0] * 0 + shadow_stp[column - k] / stp[a], prev / stp[a]);
                            break;
                        }
                        used[q] = true;
                        ProblemSlot& sh = c->slot[shadow_slot[q]];
                        std::swap(sl.x, sh.x);           // the shadow's point, gradient, e and adjoint become the trial's
                        std::swap(sl.g, sh.g);
                        std::swap(sl.w, sh.w);
                        std::swap(sl.a, sh.a);
                        double* hs = c->host_scal + (size_t)shadow_slot[q] * kScalStride;
                        double* ho = c->host_scal + (size_t)s * kScalStride;
                        for (const auto& rg : kEvalScal) {
                            note(hipMemcpyAsync(sl.scal + rg[0], sh.scal + rg[0], rg[1] * sizeof(double),
                                                hipMemcpyDeviceToDevice, c->stream), "adopt scalars");
                            // (the shadow keeps the owner's former values: a second adoption must not read them back)
                            double tmp[kScalStride];     // (the longest range is S_INV .. S_UY: kMaxSeg + 2 values)
                            std::memcpy(tmp, ho + rg[0], rg[1] * sizeof(double));
                            std::memcpy(ho + rg[0], hs + rg[0], rg[1] * sizeof(double));
                            std::memcpy(hs + rg[0], tmp, rg[1] * sizeof(double));
                        }
                        ++spec_used;
                        column = k + q;
                        TrialResult t2{ho[S_F], ho[S_DG], ho[S_GG], ho[S_XX], ho[S_DGINIT]};
                        act = p.machine->on_trial(t2);
                    }
                    if (act.kind == LbfgsMachine::ACCEPT) {
                        p.need_direction = true;
                        p.accept = true;
                        p.end = act.end;
                        p.bound = act.bound;
                        dir_list.push_back(s);
                    }
                }
                if (act.kind == LbfgsMachine::DONE) {
                    finish_problem(s, act.code, act.keep_trial, column);
                    if (next < ntheta && !rc) start_problem(s);
                }
            }
            directions(slots, dir_list);
        }
        note(hipStreamSynchronize(c->stream), "sync");
        note(check_launch());
        note(transport_error(c));              // no exchange of the run may have failed (api.hip: await_live)
        for (int s = 0; s < kMaxBatch; ++s) settle(s);
        c->spec_launched += spec_launched;
        c->spec_used += spec_used;
        if (dbg)
            for (int i = 0; i < ntheta; ++i)
                std::fprintf(stderr, "spec debug: theta %g: %d rejected trials, %d without a shadow slot, %d with shadows but none matching\n",
                             thetas[i], dbg_rej[i], dbg_noslot[i], dbg_miss[i]);
        if (verbose && spec_launched)
            std::printf("\tspeculative line-search evaluations: %lld issued in idle batch slots, %lld adopted\n",
                        spec_launched, spec_used);
        c->nvec_nt = c->nvec_nt_env == 1;          // single evaluations outside a run: the cached policy
        return rc;
    }
};

