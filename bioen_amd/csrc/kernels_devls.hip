// Kernels of the device-resident log-weights L-BFGS engine (engine_devls.inl): the line-search decision of a round is
// taken ON THE DEVICE by a one-block kernel per problem, so rounds are enqueued back to back and the host only watches
// a host-mapped page for finished problems.
//
// Reference: the iteration logic of liblbfgs-1.10's lbfgs() (third-party/liblbfgs-1.10/lib/lbfgs.c:460-616) and its
// backtracking / More-Thuente searches (:645-734, :812-1123) -- the decisions are the plain functions of
// lbfgs_state.hpp, the same source the host engines run.  BioEn's per-evaluation work (c_bioen_kernels_logw.c:525-561)
// is unchanged; what changes is who strings the kernels together:
//
//   k_dev_step       [an accepted step: d = sum_c cf_c B_c ;]  x = xp + stp d ; block maxima        (k_combine + k_trial)
//   k_dev_exp        e = exp(x - m) ; prior partials                                                 (k_logw_exp)
//   k_strip_fwd, k_fwd_rows_local_t, k_rows_combine (gated), k_strip_adj                             (unchanged)
//   k_dev_grad_gram  gradient epilogue + g.d, g.g, x.x ; (s, y) of the PENDING trial -> spare pair ; the 39 Gram
//                    products -- before the decision is known                                        (k_logw_grad + k_gram)
//   k_dev_decide     finish the sums ; line-search decision ; accepted: Gram update + two-loop recursion on 13
//                    coefficients, the spare pair joins the history ring, the roles of x/xp and g/gp swap ; publish
//                                                                                (k_finish_eval + host + k_gram_solve)
//
// Eight launches per round instead of eleven plus a host turn-around, and on sharded contexts the Gram products ride
// on the gradient's exchange.  Every sum is formed in exactly the order of the host-driven engine (engine_logw.inl):
// the two engines agree to the last bit (tests/test_hip_parity.py: test_speculative_line_search_changes_no_bit runs
// both), which is what keeps the pinned bench workload where it was.
//
// Roles live in HBM (DevSlot): which buffer is the trial point, which the accepted one, where the next (s, y) pair
// goes.  A round's kernels read them through the table, only the decision kernel writes them.  Dead problems (finished,
// not yet noticed by the host that runs a round ahead) are skipped by every N-vector kernel of the round.
//
// Speculative trials (engine_logw.inl) carry over: idle batch slots evaluate stp / 2 and 2.1 stp of their owner's
// pending trial in the same matrix passes; a rejected trial whose successor is among them ADOPTS it inside the decision
// kernel -- the owner and the shadow swap the buffers of x, g and the spare pair, the evaluation scalars are copied over.
#include "device_utils.hpp"

namespace bioen {

__device__ __forceinline__ bool dev_alive(int status) { return status == DS_INITIAL || status == DS_RUNNING; }

// (behind the kMaxBatch table entries: [0] adopted shadows, a 64-bit counter.  A round queued ahead of the host in which
// every problem has meanwhile finished still runs its matrix passes at full cost -- they do not depend on the table; all
// its N-vector work is gated off by the status words -- once per series.)

// position a of the round takes part: its owner is alive, and a speculative trial needs a line search in progress
__device__ __forceinline__ bool dev_pos_live(const DevRound& r, int a, int* owner_status) {
    const int st = r.tab[r.slot[r.owner[a]]].status;
    *owner_status = st;
    if (!dev_alive(st)) return false;
    return r.cand[a] == 0 || st == DS_RUNNING;
}

// ------------------------------------------------------------------------------------------------------------
struct DevTableInit {
    int n;
    double* x[kMaxBatch]; double* xp[kMaxBatch]; double* g[kMaxBatch]; double* gp[kMaxBatch];
    double* S[kMaxBatch][kHistory]; double* Y[kMaxBatch][kHistory];
    double* Ssp[kMaxBatch]; double* Ysp[kMaxBatch];
};

__global__ void k_dev_table_init(DevTableInit t, DevSlot* tab) {
    const int s = threadIdx.x;
    if (s >= t.n) return;
    DevSlot& T = tab[s];
    T.x = t.x[s]; T.xp = t.xp[s]; T.g = t.g[s]; T.gp = t.gp[s];
    for (int k = 0; k < kHistory; ++k) {
        T.S[k] = t.S[s][k];
        T.Y[k] = t.Y[s][k];
    }
    T.Ssp = t.Ssp[s]; T.Ysp = t.Ysp[s];
    T.status = DS_IDLE;
    T.combine = 0;
    T.code = 0; T.keep_trial = 0; T.was_initial = 0;
    T.late_end = 0; T.late_bound = 0; T.rej_dec = 0; T.rej_inc = 0; T.pad = 0;
}

// (re)start: accepted point <- start vector; direction, history and Gram state zeroed (the Gram sweep multiplies
// with every history buffer, live or not: leftovers of an earlier tenant -- possibly non-finite -- must not reach
// 0 * x); state machine reset; status = INITIAL (the next round evaluates the start point with stp = 0, d = 0).
__global__ __launch_bounds__(kBlock) void k_dev_start(DevStart s, bioen_lbfgs_config cfg, int n2) {
    const int i = blockIdx.y;
    DevSlot& T = s.tab[s.slot[i]];
    double* __restrict__ xp = T.xp;
    const double* __restrict__ g0 = s.g0[i];
    double* __restrict__ d = s.d[i];
    const d2 zero = {0.0, 0.0};
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < n2; p += gridDim.x * kBlock) {
        const int j = 2 * p;
        *reinterpret_cast<d2*>(xp + j) = *reinterpret_cast<const d2*>(g0 + j);
        *reinterpret_cast<d2*>(d + j) = zero;
#pragma unroll
        for (int k = 0; k < kHistory; ++k) {
            *reinterpret_cast<d2*>(T.S[k] + j) = zero;
            *reinterpret_cast<d2*>(T.Y[k] + j) = zero;
        }
    }
    if (blockIdx.x == 0) {
        for (int q = threadIdx.x; q < kGramStride; q += kBlock) s.gram[i][q] = 0.0;
        if (threadIdx.x == 0) {
            lb::machine_reset(T.m, cfg, T.pf);
            T.status = DS_INITIAL;
            T.combine = 0;
            T.code = 0;
            T.keep_trial = 0;
            T.was_initial = 0;
            T.rej_dec = 0;
            T.rej_inc = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// x = xp + stp d ; block maxima of x (k_trial).  After an accepted step d is formed first from the Gram coefficients
// (k_combine: d = sum_c cf_c B_c over {S_0..5, Y_0..5, gp}).  One set of blocks per OWNER: the trial points of its
// shadows (stp / 2, 2.1 stp) are written in the same sweep from the same d -- a shadow position with blocks of its own
// would have to form d a second time (it runs beside the owner's blocks, not behind them).
template <bool POLICY>
__global__ __launch_bounds__(kBlock) void k_dev_step(DevRound r, int n, Xch xo) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;                   // owner position
    int ost;
    if (!dev_pos_live(r, a, &ost)) return;
    const DevSlot& T = r.tab[r.slot[a]];
    const double stp = T.m.stp;                 // INITIAL: 0 (and d = 0): x = xp
    const int q1 = ost == DS_RUNNING ? r.shadow[a][0] : -1, q2 = ost == DS_RUNNING ? r.shadow[a][1] : -1;
    double stp1 = stp, stp2 = stp;
    stp1 *= 0.5;                                // as lb::speculative_steps / report_backtracking form them
    stp2 *= 2.1;
    double* __restrict__ x = T.x;
    double* __restrict__ x1 = q1 >= 0 ? r.tab[r.slot[q1]].x : nullptr;
    double* __restrict__ x2 = q2 >= 0 ? r.tab[r.slot[q2]].x : nullptr;
    const double* __restrict__ xp = T.xp;
    double* __restrict__ d = r.d[a];
    const bool combine = T.combine != 0;
    double cf[kBasis];
#pragma unroll
    for (int c = 0; c < kBasis; ++c) cf[c] = 0.0;
    const double* __restrict__ gn = T.gp;
    const double* Sk[kHistory];
    const double* Yk[kHistory];
#pragma unroll
    for (int k = 0; k < kHistory; ++k) {
        Sk[k] = T.S[k];
        Yk[k] = T.Y[k];
    }
    if (combine) {
        const double* coef = r.gram[a] + kBasis * kBasis;
#pragma unroll
        for (int c = 0; c < kBasis; ++c) cf[c] = coef[c];
    }
    double mx = -DBL_MAX, mx1 = -DBL_MAX, mx2 = -DBL_MAX;
    const SegPos sp = seg_pos(xo.npl, xo.segcols, n);   // 16-byte pairs; vectors are zero-padded to an even length
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(xo.npl)) {
        d2 dv;
        if (combine) {
            const d2 gv = *reinterpret_cast<const d2*>(gn + j);
            dv = d2{cf[2 * kHistory] * gv.x, cf[2 * kHistory] * gv.y};
#pragma unroll
            for (int k = 0; k < kHistory; ++k) {
                if (cf[k] != 0.0) {               // unused history slots may hold another problem's leftovers
                    const d2 v = ld_hist<POLICY>(Sk[k] + j);
                    dv.x = fma(cf[k], v.x, dv.x);
                    dv.y = fma(cf[k], v.y, dv.y);
                }
                if (cf[kHistory + k] != 0.0) {
                    const d2 v = ld_hist<POLICY>(Yk[k] + j);
                    dv.x = fma(cf[kHistory + k], v.x, dv.x);
                    dv.y = fma(cf[kHistory + k], v.y, dv.y);
                }
            }
            st_vec<POLICY>(d + j, dv);
        } else {
            dv = *reinterpret_cast<const d2*>(d + j);
        }
        const d2 pv = *reinterpret_cast<const d2*>(xp + j);
        const d2 v = {fma(stp, dv.x, pv.x), fma(stp, dv.y, pv.y)};
        st_vec<POLICY>(x + j, v);
        mx = fmax(mx, v.x);
        if (j + 1 < sp.jend) mx = fmax(mx, v.y);
        if (x1) {
            const d2 u = {fma(stp1, dv.x, pv.x), fma(stp1, dv.y, pv.y)};
            st_vec<POLICY>(x1 + j, u);
            mx1 = fmax(mx1, u.x);
            if (j + 1 < sp.jend) mx1 = fmax(mx1, u.y);
        }
        if (x2) {
            const d2 u = {fma(stp2, dv.x, pv.x), fma(stp2, dv.y, pv.y)};
            st_vec<POLICY>(x2 + j, u);
            mx2 = fmax(mx2, u.x);
            if (j + 1 < sp.jend) mx2 = fmax(mx2, u.y);
        }
    }
    mx = block_max(mx, sh);
    if (threadIdx.x == 0) xput<1>(xo, a, 0, mx);
    if (x1) {
        mx1 = block_max(mx1, sh);
        if (threadIdx.x == 0) xput<1>(xo, q1, 0, mx1);
    }
    if (x2) {
        mx2 = block_max(mx2, sh);
        if (threadIdx.x == 0) xput<1>(xo, q2, 0, mx2);
    }
}

// e = exp(x - m) ; partials of sum e and sum e (x - G): k_logw_exp with the trial point taken from the role table
template <bool POLICY>
__global__ __launch_bounds__(kBlock) void k_dev_exp(DevRound r, const double* __restrict__ G, int n, Xch xmx, Xch xo) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    int ost;
    if (!dev_pos_live(r, a, &ost)) return;
    const double* __restrict__ x = r.tab[r.slot[a]].x;
    double* __restrict__ e = r.w[a];
    const double gmax = xmax_local<1>(xmx, a, 0);
    double s = 0.0, pp = 0.0;
    const SegPos sp = seg_pos(xo.npl, xo.segcols, n);
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(xo.npl)) {
        const d2 xv = *reinterpret_cast<const d2*>(x + j);
        const d2 Gv = *reinterpret_cast<const d2*>(G + j);
        d2 ev;
        ev.x = exp(xv.x - gmax);
        ev.y = (j + 1 < sp.jend) ? exp(xv.y - gmax) : 0.0;
        st_vec<POLICY>(e + j, ev);
        s += ev.x;
        pp = fma(ev.x, xv.x - Gv.x, pp);
        s += ev.y;
        pp = fma(ev.y, xv.y - Gv.y, pp);
    }
    s = block_sum(s, sh);
    pp = block_sum(pp, sh);
    if (threadIdx.x == 0) {
        xput<3>(xo, a, 0, s);
        xput<3>(xo, a, 1, pp);
        if (sp.b == 0) xput<3>(xo, a, 2, gmax);
    }
}

// Gradient epilogue (k_logw_grad) and, in the same sweep, the Gram sweep of the PENDING trial (k_gram): s = x - xp,
// y = g - gp go to the position's spare pair -- the history ring is untouched until the trial is accepted -- and the 39
// inner products of (s, y, g) with the basis {S_0..5, Y_0..5, g} (slot `end` standing for the new pair) are left as
// block partials.  A rejected trial wasted the Gram part (13 % of the evaluations of the headline sweep); an accepted
// one saved a launch and the re-reading of x, xp, g, gp.
template <bool POLICY>
__global__ __launch_bounds__(kBlock) void k_dev_grad_gram(DevRound r, const double* __restrict__ G, int n, Xch xg, Xch xm) {
    __shared__ double sh[kWaves];
    __shared__ double shg[kWaves][64];
    const int a = blockIdx.y;
    int ost;
    if (!dev_pos_live(r, a, &ost)) return;
    const int o = r.owner[a];
    const DevSlot& T = r.tab[r.slot[o]];
    const DevSlot& P = r.tab[r.slot[a]];
    const double* __restrict__ x = P.x;
    const double* __restrict__ w = r.w[a];
    const double* __restrict__ av = r.a[a];
    const double* __restrict__ d = r.d[o];
    double* __restrict__ g = P.g;
    const double theta = r.theta[a];
    const double Pp = r.scal[a][S_P];
    const SegPos sp = seg_pos(xg.npl, xg.segcols, n);
    const double inv = r.scal[a][S_INV + sp.v];      // w = e * inv (the segment's own shift)
    // the evaluation of the start point has no pair to form; a shadow's is formed only if it is adopted AND accepted
    // (k_dev_late_gram): 14 vector passes per shadow and round for a pair that is used once in ten rounds -- unless the
    // round says otherwise (r.sgram: sharded contexts, where those passes are short and the late pass costs an all-gather)
    const bool gram = ost == DS_RUNNING && (r.cand[a] == 0 || r.sgram);
    const int e = T.m.end;                    // history slot the pair would take
    const double* __restrict__ xo_ = T.xp;
    const double* __restrict__ go = T.gp;
    double* __restrict__ Ssp = P.Ssp;
    double* __restrict__ Ysp = P.Ysp;
    const double* Sk[kHistory];
    const double* Yk[kHistory];
#pragma unroll
    for (int k = 0; k < kHistory; ++k) {
        Sk[k] = T.S[k];
        Yk[k] = T.Y[k];
    }
    double dg = 0.0, gg = 0.0, xx = 0.0;
    double acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.0;
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(xg.npl)) {
        const d2 xv = *reinterpret_cast<const d2*>(x + j);
        d2 wv = ld_hist<POLICY>(w + j);                        // pad: e = 0  =>  g = 0  (e and a: read for the last time)
        wv.x *= inv;
        wv.y *= inv;
        const d2 Gv = *reinterpret_cast<const d2*>(G + j);
        const d2 aa = ld_hist<POLICY>(av + j);
        const d2 dv = *reinterpret_cast<const d2*>(d + j);
        d2 gv;
        gv.x = wv.x * (theta * ((xv.x - Gv.x) - Pp) + aa.x);
        gv.y = wv.y * (theta * ((xv.y - Gv.y) - Pp) + aa.y);
        st_vec<POLICY>(g + j, gv);
        dg = fma(gv.x, dv.x, dg);
        gg = fma(gv.x, gv.x, gg);
        xx = fma(xv.x, xv.x, xx);
        dg = fma(gv.y, dv.y, dg);
        gg = fma(gv.y, gv.y, gg);
        xx = fma(xv.y, xv.y, xx);
        if (gram) {
            const d2 a1 = *reinterpret_cast<const d2*>(xo_ + j);
            const d2 g1 = *reinterpret_cast<const d2*>(go + j);
            const d2 sv = {xv.x - a1.x, xv.y - a1.y};
            const d2 yv = {gv.x - g1.x, gv.y - g1.y};
            d2 B[kBasis];
#pragma unroll
            for (int k = 0; k < kHistory; ++k) {
                B[k] = (k == e) ? sv : ld_hist<POLICY>(Sk[k] + j);
                B[kHistory + k] = (k == e) ? yv : ld_hist<POLICY>(Yk[k] + j);
            }
            B[2 * kHistory] = gv;
            st_vec<POLICY>(Ssp + j, sv);
            st_vec<POLICY>(Ysp + j, yv);
#pragma unroll
            for (int c = 0; c < kBasis; ++c) {
                acc[c] = fma(sv.x, B[c].x, acc[c]);
                acc[c] = fma(sv.y, B[c].y, acc[c]);
                acc[kBasis + c] = fma(yv.x, B[c].x, acc[kBasis + c]);
                acc[kBasis + c] = fma(yv.y, B[c].y, acc[kBasis + c]);
                acc[2 * kBasis + c] = fma(gv.x, B[c].x, acc[2 * kBasis + c]);
                acc[2 * kBasis + c] = fma(gv.y, B[c].y, acc[2 * kBasis + c]);
            }
        }
    }
    dg = block_sum(dg, sh);
    gg = block_sum(gg, sh);
    xx = block_sum(xx, sh);
    if (threadIdx.x == 0) {
        xput<3>(xg, a, 0, dg);
        xput<3>(xg, a, 1, gg);
        xput<3>(xg, a, 2, xx);
    }
    if (gram) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        wave_multi_reduce<64>(acc, lane);          // lane l now holds the wave total of value l
        shg[wave][lane] = acc[0];
        __syncthreads();
        if (threadIdx.x < kGramDots) {
            const double v = (shg[0][threadIdx.x] + shg[1][threadIdx.x]) + (shg[2][threadIdx.x] + shg[3][threadIdx.x]);
            xput<kGramDots>(xm, a, (int)threadIdx.x, v);
        }
    }
}

// One wave per (sum, position, local segment) totals the segment's block partials of the 39 Gram products and of
// g.d, g.g, x.x into the compact X_GRAMR stage (wave_seg_total: THE sum of a segment's partials): 42 doubles per position
// and segment serve both the line-search decision and the direction -- on sharded contexts after ONE all-gather (the
// host-driven engine exchanges the gradient's sums and the Gram products separately: three all-gathers per round
// instead of two), on one GPU straight away: the decision reads the same numbers in the same order either way.
__global__ void k_dev_rank_reduce(DevRound r, Xch xg, Xch xm, Xch xo) {
    const int c = blockIdx.x, a = blockIdx.y, v = threadIdx.x >> 6;
    int ost;
    if (!dev_pos_live(r, a, &ost)) return;
    double t = 0.0;
    if (c < kGramDots) {
        if (ost == DS_RUNNING && (r.cand[a] == 0 || r.sgram)) t = xsum_seg<kGramDots>(xm, xm.rank + v, a, c);
    } else {
        t = xsum_seg<3>(xg, xg.rank + v, a, c - kGramDots);
    }
    if ((threadIdx.x & 63) == 0) xo.base[(size_t)(xo.rank + v) * xo.payload + (size_t)a * kDevRankSums + c] = t;
}

// sum over the segments' totals (segment order: identical on every rank and GPU count); every thread computes it for itself
__device__ __forceinline__ double dev_ranks_sum(const Xch& xr, int a, int c) {
    return seg_order_sum(xr.base + (size_t)a * kDevRankSums + c, (size_t)xr.payload, xr.world);
}

// d = -gp and the partials of gp . d for the owners in `mask` whose start point was not already minimal
// (k_recur mode 0 with bound = 0); k_dev_store_dginit finishes the sum (k_store_dginit).
__global__ __launch_bounds__(kBlock) void k_dev_first_direction(DevRound r, int mask, int n, Xch xdgi) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    if (!((mask >> a) & 1)) return;
    const DevSlot& T = r.tab[r.slot[a]];
    if (T.status != DS_RUNNING) return;
    double* __restrict__ d = r.d[a];
    const double* __restrict__ gp = T.gp;
    double acc = 0.0;
    const SegPos sp = seg_pos(xdgi.npl, xdgi.segcols, n);
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(xdgi.npl)) {
        const d2 gv = *reinterpret_cast<const d2*>(gp + j);
        d2 dv;
        dv.x = -gv.x;
        dv.y = -gv.y;
        *reinterpret_cast<d2*>(d + j) = dv;
        acc = fma(gv.x, dv.x, acc);
        acc = fma(gv.y, dv.y, acc);
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) xput<1>(xdgi, a, 0, acc);
}

__global__ __launch_bounds__(kBlock) void k_dev_store_dginit(DevRound r, int mask, Xch xd) {
    __shared__ double sh[kShRed];
    const int a = blockIdx.y;
    if (!((mask >> a) & 1)) return;
    if (r.tab[r.slot[a]].status != DS_RUNNING) return;
    const double di = xsum<1>(xd, a, 0, sh);
    if (threadIdx.x == 0) r.scal[a][S_DGINIT] = di;
}

// ------------------------------------------------------------------------------------------------------------
// The decision.  One block per OWNER position:
//   * finish g.d, g.g, x.x of the evaluation (k_finish_eval's sums, same order);
//   * INITIAL: lb::on_initial; the gradient becomes the accepted one (roles of g / gp swap);
//   * RUNNING: lb::on_trial; a rejected trial whose next step is one of the owner's shadow positions adopts that
//     evaluation on the spot (second on_trial): x, g and the spare pair change hands, the evaluation-owned scalars are
//     copied;
//   * ACCEPT: finish the 39 Gram sums of the evaluation the problem stands on (in-block when there are few partials,
//     else k_dev_gram_reduce has left them), update the Gram matrix, two-loop recursion on coefficients
//     (gram_solve_thread0 = k_gram_solve), the spare pair takes slot `end` of the ring, x <-> xp, g <-> gp, and the next
//     k_dev_step forms d;
//   * publish the problem's record into the round's page of the host-mapped ring, then the round number into its flag.
// `spec` counts adopted shadows (device word, read by the host at the end of the run).
// Every sum = the segments' totals out of the X_GRAMR stage (`xm`: k_dev_rank_reduce, and on sharded contexts the
// all-gather behind it), added in segment order.
//
// Latency, not work, is what this kernel costs (one block per problem; its predecessor version took 15 us of a 160 us
// round at N = 1e5 x M = 256): the problem's table entry, its scalar slot, the Gram matrix and ALL 42 sums of the
// evaluation are fetched up front, side by side, into LDS -- one memory round trip -- and the single thread that then
// walks the decision, the Gram update and the two-loop recursion touches LDS only; the entry is written back by the
// whole block.  The sums are formed in the order of xsum / k_gram_solve: not a bit changes.
__device__ __forceinline__ void dev_gram_dots(const Xch& xm, int pos, double* dots) {
    for (int i = threadIdx.x; i < kGramDots; i += kBlock) dots[i] = dev_ranks_sum(xm, pos, i);
}

constexpr int kSlotWords = (int)(sizeof(DevSlot) / sizeof(unsigned long long));
static_assert(sizeof(DevSlot) % sizeof(unsigned long long) == 0, "DevSlot is copied in 8-byte words");

__global__ __launch_bounds__(kBlock) void k_dev_decide(DevRound r, bioen_lbfgs_config cfg, Xch xm,
                                                       double* __restrict__ page, unsigned long long* __restrict__ flags,
                                                       unsigned long long round, unsigned long long* __restrict__ spec) {
    __shared__ double dots[kGramDots];
    __shared__ double Gs[kBasis * kBasis + kBasis];   // the Gram matrix | the direction's 13 coefficients
    __shared__ double alpha[kHistory];
    __shared__ int ctl[8];                    // kind, adopt position, end, bound, code, keep_trial
    __shared__ double rec[kLiveRec];
    __shared__ double scs[kScalStride];       // the problem's scalar slot
    __shared__ unsigned long long Tw[kSlotWords];   // the problem's table entry
    const int a = blockIdx.y;
    DevSlot* Tg = r.tab + r.slot[a];
    DevSlot& T = *reinterpret_cast<DevSlot*>(Tw);
    double* scg = r.scal[a];
    double* G = r.gram[a];
#ifdef DECIDE_STAMPS
    long long st_[8];
    int sti_ = 0;
#define DSTAMP() { if (threadIdx.x == 0 && a == 0 && sti_ < 8) st_[sti_++] = __builtin_amdgcn_s_memtime(); }
#else
#define DSTAMP()
#endif
    DSTAMP()
    // ---- everything the decision may need, fetched side by side ----
    for (int i = threadIdx.x; i < kSlotWords; i += kBlock) Tw[i] = reinterpret_cast<const unsigned long long*>(Tg)[i];
    if (threadIdx.x < kScalStride) scs[threadIdx.x] = scg[threadIdx.x];
    for (int i = threadIdx.x; i < kBasis * kBasis; i += kBlock) Gs[i] = G[i];
    double sums[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) sums[q] = dev_ranks_sum(xm, a, kGramDots + q);
    __syncthreads();                          // Tw, scs, Gs are in place
    DSTAMP()                                                                     // 1: loads + three sums
    const int status = T.status;
    const bool alive = dev_alive(status);
    if (alive && status == DS_RUNNING) dev_gram_dots(xm, a, dots);               // block-uniform
    DSTAMP()                                                                     // 2: gram dots
    int kind = ACT_DONE, evalpos = a;
    bool dirty = false;
    if (alive) {
        if (threadIdx.x == 0) {
            T.m.pf = T.pf;                    // the LDS copy's own array while the decision runs
            LbfgsAction act;
            int adopt = -1;
            if (status == DS_INITIAL) {
                act = lb::on_initial(T.m, cfg, scs[S_F], sums[1], sums[2]);
                T.was_initial = 1;
                if (act.kind != ACT_DONE) {
                    double* t = T.g; T.g = T.gp; T.gp = t;      // gradient at the accepted (= start) point
                    T.status = DS_RUNNING;
                    T.was_initial = 0;
                    act.kind = ACT_TRIAL;
                }
            } else {
                const double prev = T.m.stp;
                const TrialResult t{scs[S_F], sums[0], sums[1], sums[2], scs[S_DGINIT]};
                act = lb::on_trial(T.m, cfg, t);
                if (act.kind == ACT_TRIAL) {
                    if (T.m.stp < prev) ++T.rej_dec; else ++T.rej_inc;      // what the host's shadow policy goes by
                    // rejected: is the step it asks for next among this round's shadows?
                    double cand[2];
                    const int nc = lb::speculative_steps(prev, cfg.linesearch, cand);
                    for (int q = r.nown; q < r.n && adopt < 0; ++q)
                        if (r.owner[q] == a && r.cand[q] >= 1 && r.cand[q] <= nc && cand[r.cand[q] - 1] == T.m.stp) adopt = q;
                }
            }
            scs[S_DG] = sums[0];
            scs[S_GG] = sums[1];
            scs[S_XX] = sums[2];
            ctl[0] = act.kind; ctl[1] = adopt; ctl[2] = act.end; ctl[3] = act.bound; ctl[4] = act.code; ctl[5] = act.keep_trial;
            ctl[6] = 0;
        }
        DSTAMP()                                                                 // 3: decision
        __syncthreads();
        const int adopt = ctl[1];
        if (adopt >= 0) {                      // block-uniform, rare: the shadow's sums are fetched now
#pragma unroll
            for (int q = 0; q < 3; ++q) sums[q] = dev_ranks_sum(xm, adopt, kGramDots + q);
            __syncthreads();                                       // (dots: every thread has read the owner's)
            if (r.sgram) dev_gram_dots(xm, adopt, dots);           // the shadow swept its own pair: its 39 products
            if (threadIdx.x == 0) {
                DevSlot* Q = r.tab + r.slot[adopt];           // the shadow's entry (nobody else touches it in this kernel)
                double* t;
                t = T.x; T.x = Q->x; Q->x = t;                 // the shadow's point and gradient become the trial's
                t = T.g; T.g = Q->g; Q->g = t;
                if (r.sgram) {                                 // ... and its pending (s, y) pair the problem's
                    t = T.Ssp; T.Ssp = Q->Ssp; Q->Ssp = t;
                    t = T.Ysp; T.Ysp = Q->Ysp; Q->Ysp = t;
                }
                const double* qs = r.scal[adopt];
                // the evaluation-owned entries of a slot's scalars (the rest -- y.s, alpha, gp.d -- belongs to the problem)
                for (int i = S_F; i < S_F + 4; ++i) scs[i] = qs[i];
                for (int i = S_LOGS; i < S_LOGS + 4; ++i) scs[i] = qs[i];
                for (int i = S_KL; i < S_KL + 2; ++i) scs[i] = qs[i];
                for (int i = S_INV; i < S_INV + kMaxSeg + 2; ++i) scs[i] = qs[i];      // S_INV[kMaxSeg], S_B0, S_UY
                scs[S_DG] = sums[0];
                scs[S_GG] = sums[1];
                scs[S_XX] = sums[2];
                const TrialResult t2{scs[S_F], sums[0], sums[1], sums[2], scs[S_DGINIT]};
                const LbfgsAction act = lb::on_trial(T.m, cfg, t2);
                ctl[0] = act.kind; ctl[2] = act.end; ctl[3] = act.bound; ctl[4] = act.code; ctl[5] = act.keep_trial;
                atomicAdd(spec, 1ull);
            }
            __syncthreads();
            evalpos = adopt;
        }
        kind = ctl[0];
        if (threadIdx.x == 0) {
            if (kind == ACT_ACCEPT) {
                const int e = ctl[2];
                if (evalpos == a || r.sgram) {
                    // the LDS image is also the destination of the update (a single lane's ~90 stores to HBM would
                    // sit in front of the publishing fence): the block writes it back below
                    gram_solve_thread0(Gs, Gs, dots, alpha, e, ctl[3], scs);
                    T.combine = 1;
                    ctl[6] = 1;
                } else {      // an adopted shadow carries no Gram products: k_dev_late_gram / _solve form them now
                    T.combine = 2;
                    T.late_end = e;
                    T.late_bound = ctl[3];
                }
                double* t;
                t = T.S[e]; T.S[e] = T.Ssp; T.Ssp = t;         // the pending pair joins the ring
                t = T.Y[e]; T.Y[e] = T.Ysp; T.Ysp = t;
                t = T.x; T.x = T.xp; T.xp = t;                 // the trial point becomes the accepted point
                t = T.g; T.g = T.gp; T.gp = t;
            } else {
                T.combine = 0;
                if (kind == ACT_DONE) {
                    T.status = DS_DONE;
                    T.code = ctl[4];
                    T.keep_trial = ctl[5];
                }
            }
            T.m.pf = Tg->pf;                  // back to the entry's own array
        }
        DSTAMP()                                                                 // 4: solve
        dirty = true;
    }
    // ---- the record, built in LDS ----
    if (threadIdx.x == 0) {
        DevRecord* R = reinterpret_cast<DevRecord*>(rec);
        for (int i = 0; i < kScalStride; ++i) R->scal[i] = scs[i];
        R->x = T.x; R->xp = T.xp; R->g = T.g; R->gp = T.gp;
        R->w = r.w[evalpos];
        R->fx = T.m.fx; R->stp = T.m.stp;
        R->status = T.status; R->code = T.code; R->keep_trial = T.keep_trial; R->was_initial = T.was_initial;
        R->iterations = T.m.iterations; R->evaluations = T.m.evaluations;
        R->adopted = evalpos != a; R->evalpos = evalpos;
        R->rej_dec = T.rej_dec; R->rej_inc = T.rej_inc;
    }
    __syncthreads();
    // ---- write back (the whole block), then publish ----
    if (dirty) {
        for (int i = threadIdx.x; i < kSlotWords; i += kBlock) reinterpret_cast<unsigned long long*>(Tg)[i] = Tw[i];
        if (threadIdx.x < kScalStride) scg[threadIdx.x] = scs[threadIdx.x];
        if (ctl[6])
            for (int i = threadIdx.x; i < kBasis * kBasis + kBasis; i += kBlock) G[i] = Gs[i];
    }
    static_assert(kLiveRec <= 64, "the record is published by ONE wave: its fence covers the stores of all its lanes");
    if (threadIdx.x < 64) {
        if (threadIdx.x < kLiveRec) page[(size_t)a * kLiveRec + threadIdx.x] = rec[threadIdx.x];
        __threadfence_system();               // wave-wide: s_waitcnt vmcnt(0) + write-back cover every lane's store
        if (threadIdx.x == 0)
            __hip_atomic_store(flags + a, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    DSTAMP()                                                                     // 5: record, write-back, publish
#ifdef DECIDE_STAMPS
    if (threadIdx.x == 0 && a == 0) {
        long long* out = reinterpret_cast<long long*>(spec) + 8;
        for (int i = 1; i < sti_; ++i) out[i] += st_[i] - st_[i - 1];
        out[0] += 1;
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------
// Late Gram sweep (k_gram through the role table) for the owners whose accepted step was an adopted shadow's:
// s = xp - x, y = gp - g (the roles are already swapped) go to ring slot `late_end`, the 39 products to the
// owner's X_GRAM partials.  Returns at once for everybody else.
__global__ __launch_bounds__(kBlock) void k_dev_late_gram(DevRound r, int n, Xch xo) {
    __shared__ double sh[kWaves][64];
    const int a = blockIdx.y;
    const DevSlot& T = r.tab[r.slot[a]];
    if (T.status != DS_RUNNING || T.combine != 2) return;
    const int e = T.late_end;
    const double* __restrict__ xn = T.xp;
    const double* __restrict__ xo_ = T.x;
    const double* __restrict__ gn = T.gp;
    const double* __restrict__ go = T.g;
    double* Sk[kHistory];
    double* Yk[kHistory];
#pragma unroll
    for (int k = 0; k < kHistory; ++k) {
        Sk[k] = T.S[k];
        Yk[k] = T.Y[k];
    }
    double* __restrict__ Se = T.S[e];
    double* __restrict__ Ye = T.Y[e];
    double acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.0;
    const SegPos sp = seg_pos(xo.npl, xo.segcols, n);
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(xo.npl)) {
        const d2 a0 = *reinterpret_cast<const d2*>(xn + j), a1 = *reinterpret_cast<const d2*>(xo_ + j);
        const d2 gv = *reinterpret_cast<const d2*>(gn + j), g1 = *reinterpret_cast<const d2*>(go + j);
        const d2 sv = {a0.x - a1.x, a0.y - a1.y};
        const d2 yv = {gv.x - g1.x, gv.y - g1.y};
        d2 B[kBasis];
#pragma unroll
        for (int k = 0; k < kHistory; ++k) {
            B[k] = (k == e) ? sv : *reinterpret_cast<const d2*>(Sk[k] + j);
            B[kHistory + k] = (k == e) ? yv : *reinterpret_cast<const d2*>(Yk[k] + j);
        }
        B[2 * kHistory] = gv;
        *reinterpret_cast<d2*>(Se + j) = sv;
        *reinterpret_cast<d2*>(Ye + j) = yv;
#pragma unroll
        for (int c = 0; c < kBasis; ++c) {
            acc[c] = fma(sv.x, B[c].x, acc[c]);
            acc[c] = fma(sv.y, B[c].y, acc[c]);
            acc[kBasis + c] = fma(yv.x, B[c].x, acc[kBasis + c]);
            acc[kBasis + c] = fma(yv.y, B[c].y, acc[kBasis + c]);
            acc[2 * kBasis + c] = fma(gv.x, B[c].x, acc[2 * kBasis + c]);
            acc[2 * kBasis + c] = fma(gv.y, B[c].y, acc[2 * kBasis + c]);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    wave_multi_reduce<64>(acc, lane);          // lane l now holds the wave total of value l
    sh[wave][lane] = acc[0];
    __syncthreads();
    if (threadIdx.x < kGramDots) {
        const double v = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
        xput<kGramDots>(xo, a, (int)threadIdx.x, v);
    }
}

// the local segments' totals of the late sweep's 39 products -> X_GRAMR (the 3 gradient sums are not needed again)
__global__ void k_dev_late_rank_reduce(DevRound r, Xch xm, Xch xo) {
    const int c = blockIdx.x, a = blockIdx.y, v = threadIdx.x >> 6;
    const DevSlot& T = r.tab[r.slot[a]];
    if (T.status != DS_RUNNING || T.combine != 2) return;
    const double t = xsum_seg<kGramDots>(xm, xm.rank + v, a, c);
    if ((threadIdx.x & 63) == 0) xo.base[(size_t)(xo.rank + v) * xo.payload + (size_t)a * kDevRankSums + c] = t;
}

// ... and the recursion (k_gram_solve).  The 39 sums are formed as the decision kernel forms them (the segments' totals
// in segment order), so a direction does not depend on whether its step came out of a shadow.
__global__ __launch_bounds__(kBlock) void k_dev_late_solve(DevRound r, Xch xm) {
    __shared__ double dots[kGramDots];
    __shared__ double Gs[kBasis * kBasis];
    __shared__ double alpha[kHistory];
    const int a = blockIdx.y;
    DevSlot& T = r.tab[r.slot[a]];
    if (T.status != DS_RUNNING || T.combine != 2) return;
    double* G = r.gram[a];
    for (int i = threadIdx.x; i < kBasis * kBasis; i += kBlock) Gs[i] = G[i];
    dev_gram_dots(xm, a, dots);
    __syncthreads();
    if (threadIdx.x == 0) {
        gram_solve_thread0(G, Gs, dots, alpha, T.late_end, T.late_bound, r.scal[a]);
        T.combine = 1;
    }
}

// ------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------
void launch_dev_table_init(bioen_hip_ctx* c, int nslots) {
    DevTableInit t{};
    t.n = nslots;
    for (int s = 0; s < nslots; ++s) {
        const ProblemSlot& sl = c->slot[s];
        t.x[s] = sl.xa; t.xp[s] = sl.xb; t.g[s] = sl.ga; t.gp[s] = sl.gb;
        for (int k = 0; k < kHistory; ++k) {
            t.S[s][k] = sl.S[k];
            t.Y[s][k] = sl.Yh[k];
        }
        t.Ssp[s] = sl.Ssp; t.Ysp[s] = sl.Ysp;
    }
    hipLaunchKernelGGL(k_dev_table_init, dim3(1), dim3(64), 0, c->stream, t, static_cast<DevSlot*>(c->dev_tab));
}

void launch_dev_start(bioen_hip_ctx* c, const DevStart& s, const bioen_lbfgs_config& cfg) {
    hipLaunchKernelGGL(k_dev_start, dim3(vec_blocks(c), s.n), dim3(kBlock), 0, c->stream, s, cfg, (int)(c->ld / 2));
}

void launch_dev_step(bioen_hip_ctx* c, const DevRound& r) {
    if (c->nvec_nt) hipLaunchKernelGGL(k_dev_step<true>, dim3(vec_blocks(c), r.nown), dim3(kBlock), 0, c->stream, r, c->n,
                                       make_xch(c, X_MAX, r.n * vec_grid(c)));
    else hipLaunchKernelGGL(k_dev_step<false>, dim3(vec_blocks(c), r.nown), dim3(kBlock), 0, c->stream, r, c->n,
                            make_xch(c, X_MAX, r.n * vec_grid(c)));
}

void launch_dev_exp(bioen_hip_ctx* c, const DevRound& r) {
    if (c->nvec_nt) hipLaunchKernelGGL(k_dev_exp<true>, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                                       make_xch(c, X_MAX, r.n * vec_grid(c)), make_xch(c, X_EXP, 3 * r.n * vec_grid(c)));
    else hipLaunchKernelGGL(k_dev_exp<false>, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                            make_xch(c, X_MAX, r.n * vec_grid(c)), make_xch(c, X_EXP, 3 * r.n * vec_grid(c)));
}

void launch_dev_grad_gram(bioen_hip_ctx* c, const DevRound& r) {
    if (c->nvec_nt) hipLaunchKernelGGL(k_dev_grad_gram<true>, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                                       make_xch(c, X_GRAD, 3 * r.n * vec_grid(c)), make_xch(c, X_GRAM, kGramDots * r.n * vec_grid(c)));
    else hipLaunchKernelGGL(k_dev_grad_gram<false>, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                            make_xch(c, X_GRAD, 3 * r.n * vec_grid(c)), make_xch(c, X_GRAM, kGramDots * r.n * vec_grid(c)));
}

int dev_all_fused(const bioen_hip_ctx*) { return 0; }      // (r05: every context finishes its sums from the segments' totals)

static Xch dev_rank_view(const bioen_hip_ctx* c, int n) {      // X_GRAMR: 42 values per (segment, position)
    Xch x = make_xch(c, X_GRAMR, kDevRankSums * n);
    x.npl = 1;
    return x;
}

void launch_dev_rank_reduce(bioen_hip_ctx* c, const DevRound& r) {
    hipLaunchKernelGGL(k_dev_rank_reduce, dim3(kDevRankSums, r.n), dim3(64 * c->vr), 0, c->stream, r,
                       make_xch(c, X_GRAD, 3 * r.n * vec_grid(c)), make_xch(c, X_GRAM, kGramDots * r.n * vec_grid(c)),
                       dev_rank_view(c, r.n));
}

void launch_dev_decide(bioen_hip_ctx* c, const DevRound& r, const bioen_lbfgs_config& cfg, unsigned long long round) {
    const int pg = (int)(round % kLiveRing);
    double* page = c->live2 + (size_t)pg * kMaxBatch * kLiveRec;
    unsigned long long* flags =
        reinterpret_cast<unsigned long long*>(c->live2 + (size_t)kLiveRing * kMaxBatch * kLiveRec) + (size_t)pg * kMaxBatch;
    unsigned long long* spec = reinterpret_cast<unsigned long long*>(static_cast<DevSlot*>(c->dev_tab) + kMaxBatch);
    hipLaunchKernelGGL(k_dev_decide, dim3(1, r.nown), dim3(kBlock), 0, c->stream, r, cfg, dev_rank_view(c, r.n), page, flags,
                       round, spec);
}

void launch_dev_late_gram(bioen_hip_ctx* c, const DevRound& r) {
    const Xch xm = make_xch(c, X_GRAM, kGramDots * r.n * vec_grid(c));
    hipLaunchKernelGGL(k_dev_late_gram, dim3(vec_blocks(c), r.nown), dim3(kBlock), 0, c->stream, r, c->n, xm);
    hipLaunchKernelGGL(k_dev_late_rank_reduce, dim3(kGramDots, r.nown), dim3(64 * c->vr), 0, c->stream, r, xm,
                       dev_rank_view(c, r.n));
}

void launch_dev_late_solve(bioen_hip_ctx* c, const DevRound& r) {
    hipLaunchKernelGGL(k_dev_late_solve, dim3(1, r.nown), dim3(kBlock), 0, c->stream, r, dev_rank_view(c, r.n));
}

void launch_dev_first_direction(bioen_hip_ctx* c, const DevRound& r, int mask) {
    hipLaunchKernelGGL(k_dev_first_direction, dim3(vec_blocks(c), r.nown), dim3(kBlock), 0, c->stream, r, mask, c->n,
                       make_xch(c, X_DGI, r.nown * vec_grid(c)));
}

void launch_dev_store_dginit(bioen_hip_ctx* c, const DevRound& r, int mask) {
    hipLaunchKernelGGL(k_dev_store_dginit, dim3(1, r.nown), dim3(kBlock), 0, c->stream, r, mask,
                       make_xch(c, X_DGI, r.nown * vec_grid(c)));
}

}  // namespace bioen
