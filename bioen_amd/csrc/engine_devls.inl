// Device-resident variant of the log-weights batch engine (part of api.hip's translation unit; kernels:
// kernels_devls.hip).  Replaces the per-round host decisions of LogwBatchEngine::run -- liblbfgs' iteration logic,
// third-party/liblbfgs-1.10/lib/lbfgs.c:460-616, and its line searches -- by a decision kernel per round, so that
//   * a round is 8 launches instead of 11 and ends without a host turn-around,
//   * the host enqueues round r + 1 while round r runs (`depth` rounds ahead) and only watches a host-mapped page for
//     finished problems: it composes rounds (who is in the batch, which idle slots shadow whom), starts and finishes
//     problems, nothing else,
//   * a sharded context exchanges twice per round (ybar; gradient sums + Gram products) instead of three times.
// Results equal the host-driven engine's to the last bit (same kernels' arithmetic in the same order; the tests run
// both).  BIOEN_HIP_DEVICE_LS=0 selects the host-driven engine, BIOEN_HIP_QUEUE=0 makes the host wait for every round.

struct DevFlight {                   // a round in flight, as the host composed it
    unsigned long long round = 0;
    int n = 0, nown = 0;
    int slot[kMaxBatch] = {};
    int prob[kMaxBatch] = {};        // owners: index of the problem (theta) that occupied the slot at enqueue time
};

// Which engine.  Measured in one process on one context (tools/engine_ab.py, r03; best of 4 sweeps, 8 thetas):
//     M x N             host-driven (r02)   device-resident
//     64 x 2e4            9.6 ms              8.5 ms
//     256 x 1e5 (cfg 1)   195.7 ms            185.4 ms
//     1024 x 1.25e5       480.3 ms            473.3 ms      (a rank's share of the headline at 8 GPUs)
//     1024 x 1e6          1158.5 ms           1179.0 ms     (headline)
// A round of the device engine has fewer launches and no host turn-around; a round of the host engine evaluates
// speculative trials in EVERY idle slot, which only pays where the matrix passes dwarf the N-vector work (the
// headline: -5 % rounds for +3.5 % per round; configs[1]: -3 % rounds for +8 % per round).  So the device engine
// takes the problems whose two matrix passes move less than 4 GB per round and every sharded context (one exchange
// less per round, no host decision between the ranks' collectives); BIOEN_HIP_DEVICE_LS=1 / 0 forces the choice.
bool LogwBatchEngine::device_engine_applies() const {
    if (c->live_off) return false;                      // no coherent host memory to publish into
    if (!use_gram()) return false;                      // liblbfgs' literal two-loop on the vectors: host-driven engine
    if (cfg.past > kMaxPast) return false;
    const char* e = std::getenv("BIOEN_HIP_DEVICE_LS");
    if (e && e[0] == '0') return false;
    if (e && e[0] == '1') return true;
    if (c->world > 1) return true;
    return 2.0 * c->mp * (double)c->ld * sizeof(double) < 4e9;
}

int LogwBatchEngine::ensure_device_state() {
    if (!c->dev_tab) {
        void* p = nullptr;
        const size_t bytes = (size_t)kMaxBatch * sizeof(DevSlot) + 64 + 16 * sizeof(long long);   // counters | diagnostic stamps
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc (role table)", __FILE__, __LINE__);
        BIOEN_HIP_CHECK(hipMemsetAsync(p, 0, bytes, c->stream));
        c->dev_tab = p;
    }
    if (!c->live2) {
        const size_t cnt = (size_t)kLiveRing * kMaxBatch * (kLiveRec + 1);
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&c->live2), cnt * sizeof(double),
                                     hipHostMallocCoherent | hipHostMallocMapped);
        if (e != hipSuccess) {       // no coherent host page to publish into: the host-driven engine serves this context
            (void)hipGetLastError();
            c->live2 = nullptr;
            c->live_off = 1;
            return BIOEN_HIP_ENOMEM;
        }
        std::memset(c->live2, 0, cnt * sizeof(double));
    }
    return 0;
}

// wait until the `nown` decisions of round `f` are published; -> rc
int LogwBatchEngine::await_flight(const DevFlight& f) {
    const int pg = (int)(f.round % kLiveRing);
    const volatile unsigned long long* flag =
        reinterpret_cast<const volatile unsigned long long*>(c->live2 + (size_t)kLiveRing * kMaxBatch * kLiveRec) +
        (size_t)pg * kMaxBatch;
    BoundedWait w(c, "a round's decisions (device-resident engine)");
    for (int a = 0; a < f.nown; ++a) {
        while (flag[a] != f.round) {
            const int t = w.tick([&] { return flag[a] == f.round; });
            if (t < 0) return t;
            if (t > 0) break;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return transport_error(c);      // decisions taken behind a failed exchange are not results (api.hip: await_live)
}

int LogwBatchEngine::run_device(int ntheta, const double* thetas, const double* g0_host, size_t g0_stride,
                                const double* G_host, int max_batch, double* results, double* w_opt,
                                bioen_opt_result* infos) {
    int kb = std::max(1, std::min(std::min(max_batch, kMaxBatch), ntheta));
    const bool can_speculate = speculate && cfg.linesearch >= 1 && cfg.linesearch <= 3;
    // Shadow policy.  Unsharded contexts take this engine only at sizes where a shadow costs more than it saves (table
    // in DESIGN 6a): none.  SHARDED contexts pay two all-gathers per round on top of the kernels, and a shadow's
    // N-vector work is a fraction of the unsharded one: there the host engine's policy applies -- two slots kept back
    // from a series that would fill the batch, both steps of the slowest thetas' searches evaluated from the first round
    // on (436 -> 407 rounds at the headline) -- with the shadows sweeping their own (s, y) pair and its products in the
    // main pass (`sgram`): the late Gram pass of an adopted and accepted shadow would be a third all-gather per round,
    // more than the rounds saved are worth.
    bool sgram = c->world > 1;
    if (const char* e = std::getenv("BIOEN_HIP_SHADOW_GRAM")) sgram = std::atoi(e) != 0;      // one-GPU tests of the sharded form
    const bool sharded_policy = c->world > 1;
    double shadow_rate = sharded_policy ? 0.0 : 0.08;
    int min_evals = sharded_policy ? 0 : 24, max_shadows = sharded_policy ? 2 : 0, reserve = sharded_policy ? 2 : 0;
    if (const char* e = std::getenv("BIOEN_HIP_SHADOW_RATE")) shadow_rate = std::atof(e);
    if (const char* e = std::getenv("BIOEN_HIP_SHADOW_MINEV")) min_evals = std::max(0, std::atoi(e));
    if (const char* e = std::getenv("BIOEN_HIP_SHADOWS")) max_shadows = std::max(0, std::min((int)kMaxBatch, std::atoi(e)));
    if (const char* e = std::getenv("BIOEN_HIP_DEV_RESERVE")) reserve = std::max(0, std::min(4, std::atoi(e)));
    if (can_speculate && cfg.linesearch >= 2 && max_shadows >= 2 && reserve > 0 && kb == kMaxBatch && ntheta <= kMaxBatch)
        kb -= reserve;
    // slots: the owners' kb, plus as many as may shadow at a time (none on unsharded contexts by default) -- a slot's
    // seven N-vectors (and the spare pair of a shadow that sweeps its own) are allocated only if it can be used
    const int nslots = can_speculate ? std::min((int)kMaxBatch, kb + max_shadows) : kb;
    for (int s = 0; s < nslots; ++s) note(alloc_slot(c, s, s < kb, s < kb || sgram));   // history: owners; spare pair: owners (shadows too when they sweep their own)
    if (rc) return rc;
    // cache policy of the round's N-vector kernels (device_utils.hpp: ld_hist / st_vec): nontemporal history loads and
    // outputs once the batch's vectors (21 per problem here) cannot stay in the 256 MB of Infinity Cache anyway -- a rank's
    // share of the headline at 2 or 4 GPUs; not at 8, not at configs[1], not for a K = 1 chain at N = 5e5
    c->nvec_nt = c->nvec_nt_env >= 0 ? c->nvec_nt_env == 1 : (double)c->ld * sizeof(double) * 21.0 * nslots > 256.0 * 1024 * 1024;
    note(upload_n(c, c->fixed, G_host));
    const bool shared_start = (g0_stride == 0) || ntheta == 1;
    if (shared_start) {
        if (!c->g0) note(dalloc_zero(&c->g0, c->ld, c->stream));
        if (rc) return rc;
        note(upload_n(c, c->g0, g0_host));
    }
    {   // log sum exp(G) once, written into every slot of the batch
        int all[kMaxBatch];
        for (int s = 0; s < nslots; ++s) all[s] = s;
        const Round r = make_round(c, all, nslots, nullptr, nullptr);
        note(enqueue_logs0(c, r));           // sharded: the ranks' block pairs are exchanged, every rank merges the same numbers
    }
    DevSlot* tab = static_cast<DevSlot*>(c->dev_tab);
    unsigned long long* spec_dev = reinterpret_cast<unsigned long long*>(tab + kMaxBatch);
    launch_dev_table_init(c, nslots);
    note(hipMemsetAsync(spec_dev, 0, 2 * sizeof(unsigned long long), c->stream), "memset");   // adopted shadows | problems alive

    int depth = 1;
    if (const char* e = std::getenv("BIOEN_HIP_QUEUE")) depth = std::max(0, std::min(kLiveRing - 2, std::atoi(e)));

    // host view of the slots
    bool occupied[kMaxBatch] = {};
    int prob[kMaxBatch];                               // problem index per occupied slot
    unsigned long long initial_round[kMaxBatch] = {};  // the round that evaluates the slot's start point
    std::chrono::steady_clock::time_point t0[kMaxBatch];
    int shadow_owner[kMaxBatch], shadow_cand[kMaxBatch];   // idle slots: slot of the owner they shadow (-1: none)
    unsigned long long release_round[kMaxBatch] = {};      // ... the round in which they lost their owner
    for (int s = 0; s < kMaxBatch; ++s) { prob[s] = -1; shadow_owner[s] = -1; shadow_cand[s] = 0; }
    int seen_ev[kMaxBatch] = {}, seen_dec[kMaxBatch] = {}, seen_inc[kMaxBatch] = {};   // from the newest record of each owner
    int max_positions = kMaxBatch;
    if (const char* e = std::getenv("BIOEN_HIP_SHADOW_MAXPOS")) max_positions = std::max(1, std::min((int)kMaxBatch, std::atoi(e)));
    int next = 0, active = 0;
    std::deque<DevFlight> inflight;
    std::vector<int> start_order(ntheta);            // ascending theta: the slow problems first
    for (int i = 0; i < ntheta; ++i) start_order[i] = i;
    std::stable_sort(start_order.begin(), start_order.end(), [&](int x, int y) { return theta_before(thetas[x], thetas[y]); });

    auto start_problem = [&](int s) {
        settle(s);                                   // the previous tenant's results have left
        ProblemSlot& sl = c->slot[s];
        DevStart st{};
        st.n = 1;
        st.slot[0] = s;
        st.d[0] = sl.d;
        st.gram[0] = sl.gram;
        st.tab = tab;
        const int id = start_order[next];
        if (shared_start) {
            st.g0[0] = c->g0;
        } else {                                     // staged in the slot's adjoint buffer (scratch between rounds)
            note(upload_n(c, sl.a, g0_host + (size_t)id * g0_stride));
            st.g0[0] = sl.a;
        }
        launch_dev_start(c, st, cfg);
        occupied[s] = true;
        prob[s] = id;
        initial_round[s] = c->dev_round + 1;         // the next round enqueued
        t0[s] = std::chrono::steady_clock::now();
        if (shadow_owner[s] >= 0) release_round[s] = c->dev_round + 1;
        shadow_owner[s] = -1;
        seen_ev[s] = seen_dec[s] = seen_inc[s] = 0;
        ++active;
        ++next;
    };

    auto finish_problem = [&](int s, const DevRecord& rec, int column, int width) {
        ProblemSlot& sl = c->slot[s];
        const double theta = thetas[prob[s]];
        // where this problem's averages sit in ybar_c (bioen_hip_last_average): column and row pitch of the round that
        // PUBLISHED the record -- not of a (dead) round queued behind it, whose width enqueue_round has noted since;
        // that round's gated kernels leave ybar_c as the publishing round wrote it
        c->last_pos = column;
        c->last_width = width;
        c->last_centered = false;
        bioen_opt_result& info = infos[prob[s]];
        info.lbfgs_code = rec.code;
        info.iterations = rec.iterations;
        info.evaluations = rec.evaluations;
        info.fmin = rec.fx;
        const double* res = rec.keep_trial ? rec.x : rec.xp;
        const double* h = rec.scal;
        const double* wsrc = rec.w;
        Round rr{};
        rr.n = 1;
        rr.x[0] = rec.x; rr.xp[0] = rec.xp; rr.g[0] = rec.g; rr.gp[0] = rec.gp;
        rr.d[0] = sl.d; rr.w[0] = sl.w; rr.a[0] = sl.a; rr.scal[0] = sl.scal; rr.part[0] = sl.part;
        rr.theta[0] = theta;
        if (!rec.keep_trial && !rec.was_initial) {
            // line search failed: liblbfgs returns the previous point; re-establish w, chi^2, KL there
            note(hipMemcpyAsync(rec.x, rec.xp, c->ld * sizeof(double), hipMemcpyDeviceToDevice, c->stream), "revert");
            launch_max(c, rr);
            note(enqueue_logw_eval(c, rr, false));
            note(read_scalars(c, kMaxBatch));
            res = rec.x;
            h = c->host_scal + (size_t)s * kScalStride;
            wsrc = sl.w;
        }
        info.chi2 = 0.5 * h[S_CHI];
        info.kl = h[S_P] - h[S_LOGS] + h[S_LOGS0];
        if (w_opt) {
            if (wsrc != sl.w)                        // the evaluation the problem stands on was an adopted shadow's
                note(hipMemcpyAsync(sl.w, wsrc, c->ld * sizeof(double), hipMemcpyDeviceToDevice, c->stream), "adopt e");
            launch_scale_w(c, rr);                   // e -> w, only now
        }
        if (async_delivery && c->world > 1) {
            deliver_sharded(s, results + (size_t)prob[s] * c->n_global, res,
                            w_opt ? w_opt + (size_t)prob[s] * c->n_global : nullptr, sl.w);
        } else if (async_delivery) {
            deliver(s, results + (size_t)prob[s] * c->n_global, res,
                    w_opt ? w_opt + (size_t)prob[s] * c->n_global : nullptr, sl.w);
        } else {
            note(download_n(c, results + (size_t)prob[s] * c->n_global, res));
            if (w_opt) note(download_n(c, w_opt + (size_t)prob[s] * c->n_global, sl.w));
            note(hipStreamSynchronize(c->stream), "sync");   // pageable destination: complete before the slot is reused
        }
        info.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0[s]).count();
        if (verbose) {
            std::printf("\ttheta = %g\n", theta);
            print_summary(c, info);
        }
        occupied[s] = false;
        prob[s] = -1;
        --active;
    };

    // compose and enqueue one round from what the host knows now
    auto enqueue_round = [&]() {
        DevRound r{};
        DevFlight f;
        r.tab = tab;
        const unsigned long long round = c->dev_round + 1;
        int pos_of_slot[kMaxBatch];
        for (int s = 0; s < kMaxBatch; ++s) pos_of_slot[s] = -1;
        int k = 0, first_mask = 0;
        for (int s = 0; s < kb; ++s) {
            if (!occupied[s]) continue;
            pos_of_slot[s] = k;
            r.slot[k] = s;
            r.owner[k] = k;
            r.cand[k] = 0;
            f.prob[k] = prob[s];
            if (initial_round[s] == round) first_mask |= 1 << k;
            ++k;
        }
        r.nown = k;
        for (int a = 0; a < kMaxBatch; ++a) r.shadow[a][0] = r.shadow[a][1] = -1;
        // Speculation: idle slots evaluate the steps a backtracking search may ask for next (stp / 2, 2.1 stp) in the
        // round's matrix passes.  A shadow costs ten N-vector passes, widens the batch and adds the two (gated) late-Gram
        // launches to the round; it pays only where a saved evaluation shortens the SERIES -- for the stragglers -- and
        // where trials are rejected often enough.  Measured (tools/attic/spec_probe.py, r03): shadows in every idle slot made
        // the N = 1e5 x M = 256 series (4-5 % of the stragglers' trials rejected) 8 % slower than none, while the
        // headline series (10-12 %) gains 3 %.  So: a problem is shadowed once it has shown a rejection rate of
        // `shadow_rate` (8 %) over >= 24 evaluations, and at most `max_shadows` (2) slots shadow at a time: both steps of
        // the smallest theta that qualifies (the series' rounds are the evaluations of its slowest member, which is
        // the smallest theta in every series measured; rejections are spread over the whole run, so waiting for the
        // tail of the series -- when slots abound -- catches almost none of them).
        // Assignments persist from round to round, and a slot that loses its owner rests until the host has seen the
        // rounds it took part in (the host runs `depth` rounds behind the device: the shadow may have been adopted there).
        if (can_speculate && r.nown < max_positions) {
            const int ncand = cfg.linesearch == 1 ? 1 : 2;
            int order[kMaxBatch], no = 0;                    // owners worth shadowing, by ascending theta
            for (int s = 0; s < kb; ++s) {
                if (!occupied[s] || initial_round[s] >= round) continue;
                const int rejected = seen_dec[s] + seen_inc[s];
                if (seen_ev[s] < min_evals || rejected < shadow_rate * seen_ev[s]) continue;
                order[no++] = s;
            }
            std::sort(order, order + no, [&](int x, int y) { return theta_before(thetas[prob[x]], thetas[prob[y]]); });
            int want_owner[kMaxBatch], want_cand[kMaxBatch], nwant = 0;
            const int room = std::min(std::min(max_positions - r.nown, kMaxBatch - r.nown), max_shadows);
            for (int i = 0; i < no && nwant < room; ++i)
                for (int pass = 0; pass < ncand && nwant < room; ++pass) {
                    const int first = (ncand == 2 && seen_inc[order[i]] > seen_dec[order[i]]) ? 2 : 1;
                    want_owner[nwant] = order[i];
                    want_cand[nwant] = pass == 0 ? first : 3 - first;
                    ++nwant;
                }
            bool have[kMaxBatch] = {};
            for (int s = 0; s < nslots; ++s) {               // keep what is still wanted, release the rest
                if (shadow_owner[s] < 0) continue;
                bool keep = !(s < kb && occupied[s]) && !slot_blocked(s);
                int w = -1;
                for (int i = 0; keep && i < nwant; ++i)
                    if (!have[i] && want_owner[i] == shadow_owner[s] && want_cand[i] == shadow_cand[s]) w = i;
                if (w < 0) {
                    shadow_owner[s] = -1;
                    release_round[s] = round;
                } else {
                    have[w] = true;
                }
            }
            for (int i = 0; i < nwant; ++i) {
                if (have[i]) continue;
                for (int s = nslots - 1; s >= 0; --s) {
                    if ((s < kb && occupied[s]) || shadow_owner[s] >= 0 || slot_blocked(s)) continue;
                    if (round < release_round[s] + (unsigned long long)depth + 1) continue;
                    shadow_owner[s] = want_owner[i];
                    shadow_cand[s] = want_cand[i];
                    break;
                }
            }
            for (int s = 0; s < nslots && k < kMaxBatch; ++s) {
                if (shadow_owner[s] < 0) continue;
                const int o = shadow_owner[s];
                r.slot[k] = s;
                r.owner[k] = pos_of_slot[o];
                r.cand[k] = shadow_cand[s];
                r.shadow[pos_of_slot[o]][shadow_cand[s] - 1] = k;
                ++k;
                ++spec_launched;
            }
        } else {
            for (int s = 0; s < nslots; ++s)
                if (shadow_owner[s] >= 0) {
                    shadow_owner[s] = -1;
                    release_round[s] = round;
                }
        }
        r.n = k;
        r.sgram = sgram ? 1 : 0;
        Vec8 wv{};
        MVec8 av{}, sv{};
        Round rd{};
        DevGate gate{};
        rd.n = k;
        gate.tab = tab;
        for (int a = 0; a < k; ++a) {
            const ProblemSlot& sl = c->slot[r.slot[a]];
            const ProblemSlot& ow = c->slot[r.slot[r.owner[a]]];
            r.theta[a] = thetas[prob[r.slot[r.owner[a]]]];
            r.d[a] = ow.d;
            r.gram[a] = sl.gram;
            r.w[a] = sl.w;
            r.a[a] = sl.a;
            r.scal[a] = sl.scal;
            wv.p[a] = sl.w;
            av.p[a] = sl.a;
            sv.p[a] = sl.scal;
            rd.scal[a] = sl.scal;
            rd.part[a] = sl.part;
            rd.theta[a] = r.theta[a];
            gate.slot[a] = r.slot[r.owner[a]];
            gate.cand[a] = r.cand[a];
            f.slot[a] = r.slot[a];
        }
        c->dev_round = round;
        f.round = round;
        f.n = k;
        f.nown = r.nown;
        c->last_width = k;
        c->last_pos = 0;
        c->last_centered = false;

        launch_dev_step(c, r);
        launch_dev_exp(c, r);
        int nblk = fwd_strip_blocks(c);
        if (nblk > 0) {
            int e = ensure_strip_copy(c, 0);
            if (!e) e = ensure_strip_copy_colsum(c);
            if (e && !c->strips_unavailable) note(e);
            if (e) nblk = 0;
        }
        if (nblk > 0) {
            launch_fwd_strip(c, k, wv, nblk);
            launch_fwd_rows_local(c, k, true, nblk, true);
            note(exchange(c, X_YBAR, (size_t)ybar_payload(c, k, true)));
            launch_rows_combine(c, rd, true, c->strip_center, true, &gate);
            launch_adj_strip(c, k, c->r_c, av, sv, nblk);
        } else {
            note(ensure_rowmajor(c));
            launch_fwd_partial(c, k, wv);
            launch_fwd_rows_local(c, k, true);
            note(exchange(c, X_YBAR, (size_t)ybar_payload(c, k, true)));
            launch_rows_combine(c, rd, true, nullptr, true, &gate);
            launch_adj(c, k, c->r_c, av, true);
        }
        launch_dev_grad_gram(c, r);
        launch_dev_rank_reduce(c, r);               // the local segments' totals of the 39 + 3 sums (sharded: what is shipped)
        note(exchange(c, X_GRAMR, (size_t)kDevRankSums * k));
        launch_dev_decide(c, r, cfg, round);
        if (r.n > r.nown && !sgram) {               // a round with shadows: one of them may have been adopted and accepted
            launch_dev_late_gram(c, r);
            note(exchange(c, X_GRAMR, (size_t)kDevRankSums * k));
            launch_dev_late_solve(c, r);
        }
        if (first_mask) {
            launch_dev_first_direction(c, r, first_mask);
            note(exchange(c, X_DGI, (size_t)r.nown * vec_grid(c)));
            launch_dev_store_dginit(c, r, first_mask);
        }
        note(check_launch());
        inflight.push_back(f);
    };

    for (int s = 0; s < kb && next < ntheta; ++s) start_problem(s);

    while (active > 0 && !rc) {
        jitter(2);
        while ((int)inflight.size() <= depth && !rc) enqueue_round();
        if (rc) break;
        const DevFlight f = inflight.front();
        inflight.pop_front();
        note(await_flight(f));
        if (rc) break;
        const double* page = c->live2 + (size_t)(f.round % kLiveRing) * kMaxBatch * kLiveRec;
        for (int a = 0; a < f.nown; ++a) {
            const int s = f.slot[a];
            if (!occupied[s] || prob[s] != f.prob[a]) continue;      // finished before; the slot may have a new tenant
            DevRecord rec;
            std::memcpy(&rec, page + (size_t)a * kLiveRec, sizeof rec);
            seen_ev[s] = rec.evaluations;
            seen_dec[s] = rec.rej_dec;
            seen_inc[s] = rec.rej_inc;
            if (rec.status != DS_DONE) continue;
            finish_problem(s, rec, rec.evalpos, f.n);
            if (next < ntheta && !rc) start_problem(s);
        }
    }
    note(hipStreamSynchronize(c->stream), "sync");      // rounds still in flight are dead ones: every problem has finished
    note(check_launch());
    note(transport_error(c));                           // ... and none of their exchanges may have failed: deliveries are settled below
    for (int s = 0; s < kMaxBatch; ++s) settle(s);
    if (!rc) {
        unsigned long long used = 0;
        note(hipMemcpy(&used, spec_dev, sizeof used, hipMemcpyDeviceToHost), "spec");
        spec_used += (long long)used;
    }
    if (std::getenv("BIOEN_HIP_DECIDE_STAMPS")) {     // diagnostic builds (-DDECIDE_STAMPS): phase cycles of k_dev_decide, block 0
        long long st[8] = {};
        (void)hipMemcpy(st, reinterpret_cast<long long*>(spec_dev) + 8, sizeof st, hipMemcpyDeviceToHost);
        if (st[0] > 0)
            std::fprintf(stderr, "k_dev_decide stamps (100 MHz ticks per call): loads+sums %.1f, gram dots %.1f, decision %.1f, solve %.1f, publish %.1f over %lld calls\n",
                         (double)st[1] / st[0], (double)st[2] / st[0], (double)st[3] / st[0], (double)st[4] / st[0], (double)st[5] / st[0], st[0]);
        (void)hipMemset(reinterpret_cast<long long*>(spec_dev) + 8, 0, sizeof st);
    }
    c->spec_launched += spec_launched;
    c->spec_used += spec_used;
    if (verbose && spec_launched)
        std::printf("\tspeculative line-search evaluations: %lld issued in idle batch slots, %lld adopted\n",
                    spec_launched, spec_used);
    c->nvec_nt = c->nvec_nt_env == 1;
    return rc;
}
