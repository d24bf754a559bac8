// gfx950 (CDNA4, wave64) kernels of the BioEn log-weights / forces hot path.
//
// Two kernels touch the M x N matrix and carry >99 % of the bytes:
//   k_fwd_partial : ybar_a = yTilde . v_a    (replaces _bioen_chi_squared's GEMV,
//                                             c_bioen_common.c:76-86, and _getAve,
//                                             c_bioen_kernels_forces.c:93-109)
//   k_adj         : out_a  = yTilde^T . u_a  (replaces the transposed-cache walks of
//                                             c_bioen_kernels_logw.c:185-205 and
//                                             c_bioen_kernels_forces.c:127-150,300-320)
// for a = 0..K-1: up to K = 8 optimisation problems (thetas of a series) share one pass,
// so the matrix bytes per problem drop by K while the arithmetic per problem -- and its
// order -- is exactly that of a K = 1 launch (batched runs are bitwise equal to single
// runs).  Both kernels stream the row-major matrix once with 16-byte-per-lane loads (one
// aligned KiB per wave instruction) straight into registers: the operand is read once and
// not shared across waves, so an LDS round trip would be pure overhead.  FP64 FMA issue
// stays far below the HBM-bound budget up to K = 8 (2K flop per 8 bytes), which is why the
// batch runs on the vector ALU rather than on v_mfma_f64 (whose 16x16x4 shape would also
// force 64-byte row fragments instead of KiB-wide coalesced loads).
// Every reduction has a fixed shape => results are bitwise reproducible run to run.
#include "device_utils.hpp"

namespace bioen {

// ------------------------------------------------------------------------------
// forward pass: partial[(row*K + a)*ctiles + tile] = sum_{j in tile} Y[row][j] v_a[j]
//   block = 4 waves stacked over rows, R rows per wave; a wave walks its column tile in
//   128-column (1 KiB) steps, STEPS steps in flight, K x R accumulators.
//   CENTER: sum_j (Y[row][j] - ybar_a[row]) v_a[j] -- the centred form the reference uses for the
//   forces gradient (c_bioen_kernels_forces.c:330-338); ybar_a[row] is wave-uniform and read
//   through the scalar cache (compact layout [row*K + a]).
// ------------------------------------------------------------------------------
template <int R, int K, int STEPS, bool NT, bool CENTER>
__global__ __launch_bounds__(kBlock) void k_fwd_partial(const double* __restrict__ Y, size_t ld, Vec8 v,
                                                        const double* __restrict__ ybar_c,
                                                        double* __restrict__ partial, int ctiles,
                                                        int steps_per_tile, int seg_tiles, int seg_steps) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // blockIdx.x = row block (fast index): consecutive blocks share the column tile, so the
    // tile's slice of v_a is fetched from HBM once per XCD and then served by L2
    // tile t of segment v (seg_tiles tiles of steps_per_tile 128-column steps each, the last one shorter): the same
    // columns on every GPU count
    const int tile = blockIdx.y;
    const int row0 = (blockIdx.x * kWaves + wave) * R;
    const int tseg = tile / seg_tiles, tl = tile - tseg * seg_tiles;
    int s = tseg * seg_steps + tl * steps_per_tile;
    int s_end = s + steps_per_tile;
    if (s_end > (tseg + 1) * seg_steps) s_end = (tseg + 1) * seg_steps;

    const size_t col = (size_t)s * 128 + lane * 2;
    const double* yp = Y + (size_t)row0 * ld + col;

    constexpr int KP = next_pow2(K);
    double acc[KP * R];
#pragma unroll
    for (int i = 0; i < KP * R; ++i) acc[i] = 0.0;
    const double* ybp = ybar_c + (size_t)row0 * K;   // [r*K + k], wave-uniform

    size_t off = col;
    if constexpr (STEPS == 0) {
        // software pipeline: the NEXT step's rows of Y are in flight while this step's K x R
        // products are formed (two register sets, no moves)
        d2 ya[R], yb2[R];
        d2 vv[K];
        auto loadY = [&](d2* y, int step) {
#pragma unroll
            for (int r = 0; r < R; ++r) y[r] = ldg2<NT>(yp + (size_t)r * ld + (size_t)step * 128);
        };
        auto work = [&](const d2* y, int step) {
#pragma unroll
            for (int k = 0; k < K; ++k) vv[k] = *reinterpret_cast<const d2*>(v.p[k] + off + (size_t)step * 128);
#pragma unroll
            for (int k = 0; k < K; ++k) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double yb = CENTER ? ybp[r * K + k] : 0.0;
                    acc[k * R + r] = fma(y[r].x - yb, vv[k].x, acc[k * R + r]);
                    acc[k * R + r] = fma(y[r].y - yb, vv[k].y, acc[k * R + r]);
                }
            }
        };
        const int nsteps = s_end - s;
        int t = 0;
        if (nsteps > 0) loadY(ya, 0);
        for (; t + 1 < nsteps; t += 2) {
            loadY(yb2, t + 1);
            work(ya, t);
            if (t + 2 < nsteps) loadY(ya, t + 2);
            work(yb2, t + 1);
        }
        if (t < nsteps) work(ya, t);
        s = s_end;
    }
    constexpr int ST = STEPS == 0 ? 1 : STEPS;
    for (; s + ST <= s_end; s += ST) {
        d2 y[ST][R];
        d2 vv[ST][K];
#pragma unroll
        for (int t = 0; t < ST; ++t) {
#pragma unroll
            for (int r = 0; r < R; ++r) y[t][r] = ldg2<NT>(yp + (size_t)r * ld + t * 128);
#pragma unroll
            for (int k = 0; k < K; ++k) vv[t][k] = *reinterpret_cast<const d2*>(v.p[k] + off + t * 128);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double yb = CENTER ? ybp[r * K + k] : 0.0;
#pragma unroll
                for (int t = 0; t < ST; ++t) {
                    acc[k * R + r] = fma(y[t][r].x - yb, vv[t][k].x, acc[k * R + r]);
                    acc[k * R + r] = fma(y[t][r].y - yb, vv[t][k].y, acc[k * R + r]);
                }
            }
        }
        yp += ST * 128;
        off += ST * 128;
    }
    for (; s < s_end; ++s) {   // tail (only when STEPS = 2 and the tile has an odd step count)
        d2 y[R];
        d2 vv[K];
#pragma unroll
        for (int r = 0; r < R; ++r) y[r] = ldg2<NT>(yp + (size_t)r * ld);
#pragma unroll
        for (int k = 0; k < K; ++k) vv[k] = *reinterpret_cast<const d2*>(v.p[k] + off);
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double yb = CENTER ? ybp[r * K + k] : 0.0;
                acc[k * R + r] = fma(y[r].x - yb, vv[k].x, acc[k * R + r]);
                acc[k * R + r] = fma(y[r].y - yb, vv[k].y, acc[k * R + r]);
            }
        }
        yp += 128;
        off += 128;
    }

    constexpr int NV = KP * R;
    wave_multi_reduce<NV>(acc, lane);
    constexpr int SHIFT = (NV == 8) ? 3 : (NV == 16) ? 2 : (NV == 32) ? 1 : 0;   // 6 - log2(NV)
    if ((lane & ((1 << SHIFT) - 1)) == 0) {
        const int idx = lane >> SHIFT;
        const int k = idx / R, r = idx % R;
        if (k < K) partial[((size_t)(row0 + r) * K + k) * ctiles + tile] = acc[0];
    }
}

// reduce the column tiles of one (row, problem) per wave (fixed order): a SEGMENT's share of
// ybar, written into that segment of the X_YBAR stage (compact layout [row*K + a]); blockIdx.z = local segment.
// WITH_EXP (log-weights rounds): block 0 of each problem also totals the segment's softmax partials
// and appends {sum e, sum e (x - G), m_v} to the segment, so the normalisation needs no
// exchange of its own.
template <bool WITH_EXP>
__global__ __launch_bounds__(kBlock) void k_fwd_rows_local(const double* __restrict__ partial, int ctiles, int seg_tiles,
                                                           int mp, int K, Xch xo, Xch xe) {
    const int a = blockIdx.y, v = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double* out = xo.base + (size_t)(xo.rank + v) * xo.payload;
    if (WITH_EXP && blockIdx.x == 0) {
        const double s = xsum_seg<3>(xe, xe.rank + v, a, 0);
        const double pp = xsum_seg<3>(xe, xe.rank + v, a, 1);
        if (threadIdx.x == 0) {
            double* tail = out + (size_t)mp * K + 3 * a;
            tail[0] = s;
            tail[1] = pp;
            tail[2] = xseg_ptr<3>(xe, xe.rank + v, a, 2)[0];
        }
    }
    for (int row = blockIdx.x * kWaves + wave; row < mp; row += gridDim.x * kWaves) {
        const double* p = partial + ((size_t)row * K + a) * ctiles + (size_t)v * seg_tiles;
        double s = 0.0;
        for (int k = lane; k < seg_tiles; k += 64) s += p[k];
        s = wave_sum(s);
        if (lane == 0) out[(size_t)row * K + a] = s;
    }
}

// k_fwd_rows_local on transposed partials (the strip kernels: partial[set * n + entry], entry = row K + a, n = mp K).
// blockIdx.y = local segment v, whose sets are [v gs fold, (v + 1) gs fold).  THE share of an entry in a segment
// (kernels.hpp: StripSets): the values of its gs groups -- each the sum of its `fold` chunk sets in turn from +0.0 where
// the strip kernel has not folded them itself -- are dealt to EIGHT running sums (part p: groups p, p + 8, ... in turn,
// from +0.0), which meet as ((s0 + s4) + (s2 + s6)) + ((s1 + s5) + (s3 + s7)).  A block = 8 parts x 32 entries: every
// load of a wave is 32 consecutive entries of one set (256 B), 2048 blocks at the headline; the classic
// wave-per-entry tree of r02-r04 (tiles_sum16) had 32 of its 64 lanes idle on the 32 groups of a segment and cost 8 x its
// time over the eight segments.  The blocks beyond the entry tiles (one per problem) total the softmax partials.
template <bool WITH_EXP>
__global__ __launch_bounds__(kBlock) void k_fwd_rows_local_t(const double* __restrict__ partial, int gs, int fold,
                                                             int mp, int K, int ntile, Xch xo, Xch xe) {
    __shared__ double lds[8][32];
    const int v = blockIdx.y;
    double* out = xo.base + (size_t)(xo.rank + v) * xo.payload;
    if (WITH_EXP && (int)blockIdx.x >= ntile) {
        const int a = blockIdx.x - ntile;
        const double s = xsum_seg<3>(xe, xe.rank + v, a, 0);
        const double pp = xsum_seg<3>(xe, xe.rank + v, a, 1);
        if (threadIdx.x == 0) {
            double* tail = out + (size_t)mp * K + 3 * a;
            tail[0] = s;
            tail[1] = pp;
            tail[2] = xseg_ptr<3>(xe, xe.rank + v, a, 2)[0];
        }
        return;
    }
    const size_t n = (size_t)mp * K;
    const size_t idx = (size_t)blockIdx.x * 32 + (threadIdx.x & 31);
    const bool valid = idx < n;
    const double s = sets_sum8(partial + (size_t)v * gs * fold * n + (valid ? idx : 0), n, gs, fold, lds, TermAdd());
    if (threadIdx.x < 32 && valid) out[idx] = s;
}

// forces gradient from transposed partials (the strip kernels).  blockIdx.y = local segment v, whose sets are
// [v gs fold, (v + 1) gs fold) -- gs values of `fold` chunk sets each, as in k_fwd_rows_local_t (the two-pass strip
// kernels: fold = 1, gs = the segment's sets): the segment's share  sum_sets partial - ybar' T_v  (T_v = the sets'
// shares of sum_j t_j, totalled in the same fixed order by every block) goes to out + v * out_stride -- the segment's
// part of the X_YBAR stage (k_sum_ranks adds the segments up in order), or, with one "segment" holding every set, gm
// itself (the streaming-order fallback of unsharded contexts).
__global__ __launch_bounds__(kBlock) void k_fwd_rows_forces_grad_t(const double* __restrict__ partial, int gs, int fold,
                                                                   int mp, int K, double* __restrict__ out, size_t out_stride,
                                                                   const double* __restrict__ ybar_c, MVec8 tpart) {
    __shared__ double lds[8][32];
    __shared__ double T[kMaxBatch];
    const int v = blockIdx.y;
    const int seg_sets = gs * fold;
    if (ybar_c)
        for (int a = 0; a < K; ++a) {
            const double t = sum_partials(tpart.p[a] + (size_t)P_KL * kPartStride + (size_t)v * seg_sets, seg_sets, &lds[0][0]);
            if (threadIdx.x == 0) T[a] = t;
        }
    __syncthreads();
    const size_t n = (size_t)mp * K;
    const size_t idx = (size_t)blockIdx.x * 32 + (threadIdx.x & 31);
    const bool valid = idx < n;
    const double s = sets_sum8(partial + (size_t)v * seg_sets * n + (valid ? idx : 0), n, gs, fold, lds, TermAdd());
    if (threadIdx.x < 32 && valid) (out + (size_t)v * out_stride)[idx] = ybar_c ? fma(-ybar_c[idx], T[idx % K], s) : s;
}

// add the segments' shares (segment order) -> ybar, r = ybar - YT (compact) ; per-block partials of
// sum r^2 and sum ybar r.  Every rank computes the same numbers.
// Affine observable model (ctx.hpp): ybar_eff_i = off_i + sc_i (Y w)_i  (sum w = 1), while ybar_c
// keeps the RAW Y w, which is what the centred passes subtract: the offset cancels in
// sum_i r_i (yTilde_eff_ik - ybar_eff_i) = sum_i (r_i sc_i) (Y_ik - (Y w)_i), so the adjoint's
// operand r_c is stored pre-multiplied by sc_i.  (off, sc) = (0, 1): the plain model, same bits.
// One block per problem (M is a few thousand rows at most in BioEn's use), so the block also
// finishes the sums: LOGW: f = theta (P - log s + log s0) + 0.5 sum r^2  (c_bioen_kernels_logw.c:124-147)
// lands in scal[S_F] without a further launch; forces: partial 0 feeds k_forces_scalars.
template <bool LOGW>
__global__ __launch_bounds__(kBlock) void k_rows_combine(Xch xi, int mp, int K, const double* __restrict__ YT,
                                                         const double* __restrict__ row_offset,
                                                         const double* __restrict__ row_scale,
                                                         const double* __restrict__ center, bool store_raw,
                                                         double* __restrict__ ybar_c, double* __restrict__ r_c,
                                                         MVec8 part, Round rd, DevGate gate) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    if (gate.tab) {      // device-resident engine: a finished problem keeps the scalars of its last evaluation
        const int st = gate.tab[gate.slot[a]].status;
        if (!(st == DS_INITIAL || st == DS_RUNNING) || (gate.cand[a] != 0 && st != DS_RUNNING)) return;
    }
    // LOGW: the shares are yTilde . e_v with e_v = exp(x - m_v) unnormalised (segment v's own shift); global shift
    // M = max_v m_v, S = sum_v e^{m_v - M} S_v, and segment v's share enters with e^{m_v - M} / S.  _get_weights' normalisation (c_bioen_kernels_logw.c:84-90) is
    // thereby applied to the M sums instead of the N weights.
    __shared__ double fac[kShRed];     // LOGW: e^{m_v - M} / S per segment
    double gmax = 0.0, invS = 1.0;
    const size_t tail_at = (size_t)mp * K + 3 * a;
    // Up to eight segments (every world that divides 8, a single GPU included): their values are loaded TOGETHER and then
    // used in segment order -- with the loops' trip count a kernel argument the loads went out one by one, and this
    // one-block kernel is nothing but latency (r05: 14 us per round, twice r04's with one segment)
    const bool few = xi.world <= kMaxSeg;
    if (LOGW) {
        double S = 0.0, PP = 0.0;
        if (few) {
            double t0[kMaxSeg], t1[kMaxSeg], t2[kMaxSeg];
#pragma unroll
            for (int r = 0; r < kMaxSeg; ++r) {
                const double* tail = xi.base + (size_t)(r < xi.world ? r : 0) * xi.payload + tail_at;
                t0[r] = tail[0];
                t1[r] = tail[1];
                t2[r] = tail[2];
            }
            gmax = -DBL_MAX;
#pragma unroll
            for (int r = 0; r < kMaxSeg; ++r)
                if (r < xi.world) gmax = fmax(gmax, t2[r]);
#pragma unroll
            for (int r = 0; r < kMaxSeg; ++r)
                if (r < xi.world) {
                    const double fr = exp(t2[r] - gmax);
                    S = fma(fr, t0[r], S);
                    PP = fma(fr, t1[r], PP);
                }
            invS = 1.0 / S;
            if (threadIdx.x < kMaxSeg) {
                double mine = t2[0];
#pragma unroll
                for (int r = 1; r < kMaxSeg; ++r)
                    if ((int)threadIdx.x == r) mine = t2[r];
                if ((int)threadIdx.x < xi.world) fac[threadIdx.x] = exp(mine - gmax) * invS;
            }
        } else {
            gmax = -DBL_MAX;
            for (int r = 0; r < xi.world; ++r) gmax = fmax(gmax, xi.base[(size_t)r * xi.payload + tail_at + 2]);
            for (int r = 0; r < xi.world; ++r) {
                const double* tail = xi.base + (size_t)r * xi.payload + tail_at;
                const double fr = exp(tail[2] - gmax);
                S = fma(fr, tail[0], S);
                PP = fma(fr, tail[1], PP);
            }
            invS = 1.0 / S;
            for (int r = threadIdx.x; r < xi.world; r += kBlock)
                fac[r] = exp(xi.base[(size_t)r * xi.payload + tail_at + 2] - gmax) * invS;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double* sc = rd.scal[a];
            sc[S_LOGS] = gmax + log(S);
            sc[S_P] = PP * invS;
        }
        if ((int)threadIdx.x < xi.vr)      // w = e * S_INV[v] in local segment v (its own shift m_v): the factor of segment rank + v
            rd.scal[a][S_INV + threadIdx.x] = fac[xi.rank + threadIdx.x];
    }
    double chi = 0.0, cc = 0.0, b0 = 0.0, uy = 0.0;
    // a row's finish: the sums below run over a thread's rows in row order, whichever way the shares were fetched
    auto finish_row = [&](int row, double s, double sc, double off, double yt, double cen) {
        const double raw = s + cen;
        const double eff = fma(sc, raw, off);
        const double res = eff - yt;
        ybar_c[(size_t)row * K + a] = store_raw ? raw : s;
        r_c[(size_t)row * K + a] = res * sc;
        chi = fma(res, res, chi);
        cc = fma(eff, res, cc);
        b0 = fma(cen, res * sc, b0);
        uy = fma(raw, res * sc, uy);
    };
    if (few) {
        // four of the thread's rows at a time (all of them for M <= 1024): 32 shares and the rows' constants in flight together
        constexpr int RB = 4;
        for (int row0 = threadIdx.x; row0 < mp; row0 += RB * kBlock) {
            double v[RB][kMaxSeg], sc[RB], off[RB], yt[RB], cen[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int row = row0 + i * kBlock < mp ? row0 + i * kBlock : row0;
#pragma unroll
                for (int r = 0; r < kMaxSeg; ++r) v[i][r] = xi.base[(size_t)(r < xi.world ? r : 0) * xi.payload + (size_t)row * K + a];
                sc[i] = row_scale[row];
                off[i] = row_offset[row];
                yt[i] = YT[row];
                cen[i] = center ? center[row] : 0.0;             // shares of the centred copy: ybar_raw = s + center
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if (row0 + i * kBlock >= mp) break;
                double s = 0.0;
#pragma unroll
                for (int r = 0; r < kMaxSeg; ++r)
                    if (r < xi.world) {
                        if (LOGW) s = fma(fac[r], v[i][r], s);
                        else s += v[i][r];
                    }
                finish_row(row0 + i * kBlock, s, sc[i], off[i], yt[i], cen[i]);
            }
        }
    } else {
        for (int row = threadIdx.x; row < mp; row += kBlock) {
            double s = 0.0;
            for (int r = 0; r < xi.world; ++r) {
                const double v = xi.base[(size_t)r * xi.payload + (size_t)row * K + a];
                if (LOGW) s = fma(fac[r], v, s);
                else s += v;
            }
            finish_row(row, s, row_scale[row], row_offset[row], YT[row], center ? center[row] : 0.0);
        }
    }
    chi = block_sum(chi, sh);
    cc = block_sum(cc, sh);
    b0 = block_sum(b0, sh);
    uy = block_sum(uy, sh);
    if (threadIdx.x == 0) {
        if (LOGW) {
            double* sc = rd.scal[a];            // S_P, S_LOGS: written above by this same thread
            sc[S_CHI] = chi;
            sc[S_C] = cc;
            sc[S_B0] = b0;
            sc[S_UY] = uy;
            sc[S_KL] = sc[S_P] - sc[S_LOGS] + sc[S_LOGS0];     // theta's factor: KL(w || w0) in both methods
            sc[S_F] = rd.theta[a] * sc[S_KL] + 0.5 * chi;
        } else {
            double* pa = part.p[a];
            pa[(size_t)P_CHI * kPartStride] = chi;
            pa[(size_t)P_C * kPartStride] = cc;
        }
    }
}

// forces gradient (c_bioen_kernels_forces.c:330-338).  Streaming path: the partials already hold the centred
// sums  sum_j (Y_ij - ybar_i) t_j ; only the column tiles remain to be added up.  Strip passes on the
// centred copy (kernels_strip.hip): the partials hold sum_j Y'_ij t_j and  ybar'_i sum_j t_j  is taken off
// here, T = the blocks' shares of sum_j t_j in the problem's P_KL partials (every block re-sums them in the
// same fixed order).
__global__ __launch_bounds__(kBlock) void k_fwd_rows_forces_grad(const double* __restrict__ partial, int ctiles,
                                                                 int mp, int K, double* __restrict__ gm_c,
                                                                 const double* __restrict__ ybar_c, MVec8 tpart) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double T = 0.0;
    if (ybar_c) T = sum_partials(tpart.p[a] + (size_t)P_KL * kPartStride, ctiles, sh);
    for (int row = blockIdx.x * kWaves + wave; row < mp; row += gridDim.x * kWaves) {
        const double* p = partial + ((size_t)row * K + a) * ctiles;
        double s = 0.0;
        for (int k = lane; k < ctiles; k += 64) s += p[k];
        s = wave_sum(s);
        if (lane == 0) gm_c[(size_t)row * K + a] = ybar_c ? fma(-ybar_c[(size_t)row * K + a], T, s) : s;
    }
}

// ------------------------------------------------------------------------------
// adjoint pass: out_a[j] = sum_i Y[i][j] u_a[i]      (u, ybar compact: [i*K + a])
//   block = one 128-column strip (a lane owns 2 adjacent columns = 16 B), the 4 waves split
//   the rows; U rows (U KiB) in flight per wave; the K operands of a row are wave-uniform
//   and come through the scalar cache with one load.
//   CENTER: out_a[j] = sum_i u_a[i] (Y[i][j] - ybar_a[i]) -- the reference's centred gradient
//   sum (c_bioen_kernels_logw.c:185-195); padded columns then hold -u.ybar, which nobody reads.
// ------------------------------------------------------------------------------
//   LDSOPS (K = 4..7; at K = 8 it measured 3 % slower than the scalar path): 16 K doubles of row operands per 8-row step exceed what the scalar cache path
//   can keep in SGPRs (a diagnostic build with constant operands runs K = 8 at the K = 1 time), so a
//   wave stages the {u, ybar} pairs of its next 32 rows in LDS -- inside its own slice of the final
//   reduction buffer, which it does not need before its loop ends -- and reads them back as
//   same-address (broadcast) 16-byte loads.  Same operands, same order: bitwise the scalar path.
template <int U, int K, bool NT, bool CENTER, bool LDSOPS>
__global__ __launch_bounds__(kBlock) void k_adj(const double* __restrict__ Y, size_t ld, int rows_per_wave,
                                                const double* __restrict__ u_c,
                                                const double* __restrict__ ybar_c, MVec8 out) {
    __shared__ d2 red[kWaves][K][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t col = (size_t)blockIdx.x * 128 + lane * 2;
    const int r0 = wave * rows_per_wave;
    const double* yp = Y + (size_t)r0 * ld + col;
    const double* up = u_c + (size_t)r0 * K;
    const double* bp = ybar_c + (size_t)r0 * K;

    d2 acc0[K], acc1[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        acc0[k] = d2{0.0, 0.0};
        acc1[k] = d2{0.0, 0.0};
    }
    d2* const ops = &red[wave][0][0];          // 64 K entries; 32 K used as [row of the chunk][k] = {u, ybar}
    for (int i = 0; i < rows_per_wave; i += U) {
        if (LDSOPS && (i & 31) == 0) {    // before this step's matrix loads: vmcnt retires in order
            int cnt = rows_per_wave - i;
            cnt = (cnt < 32 ? cnt : 32) * K;
            for (int e = lane; e < cnt; e += 64) {
                d2 o;
                o.x = up[(size_t)i * K + e];
                o.y = CENTER ? bp[(size_t)i * K + e] : 0.0;
                ops[e] = o;
            }
            // wave-private LDS: the wave's own program order is the synchronisation; fences for the compiler
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        d2 y[U];
#pragma unroll
        for (int q = 0; q < U; ++q) y[q] = ldg2<NT>(yp + (size_t)q * ld);
#pragma unroll
        for (int q = 0; q < U; q += 2) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                double u0, u1, b0, b1;
                if (LDSOPS) {
                    const d2 o0 = ops[((i & 31) + q) * K + k], o1 = ops[((i & 31) + q + 1) * K + k];
                    u0 = o0.x; b0 = o0.y; u1 = o1.x; b1 = o1.y;
                } else {
                    u0 = up[(size_t)(i + q) * K + k];
                    u1 = up[(size_t)(i + q + 1) * K + k];
                    b0 = CENTER ? bp[(size_t)(i + q) * K + k] : 0.0;
                    b1 = CENTER ? bp[(size_t)(i + q + 1) * K + k] : 0.0;
                }
                acc0[k].x = fma(y[q].x - b0, u0, acc0[k].x);
                acc0[k].y = fma(y[q].y - b0, u0, acc0[k].y);
                acc1[k].x = fma(y[q + 1].x - b1, u1, acc1[k].x);
                acc1[k].y = fma(y[q + 1].y - b1, u1, acc1[k].y);
            }
        }
        yp += (size_t)U * ld;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) red[wave][k][lane] = d2{acc0[k].x + acc1[k].x, acc0[k].y + acc1[k].y};
    __syncthreads();
    for (int k = wave; k < K; k += kWaves) {
        const d2 a0 = red[0][k][lane], a1 = red[1][k][lane], a2 = red[2][k][lane], a3 = red[3][k][lane];
        d2 o = {(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y)};
        *reinterpret_cast<d2*>(out.p[k] + col) = o;
    }
}


// ==============================================================================
// host-side launchers
// ==============================================================================

int vec_grid(const bioen_hip_ctx* c) {
    // blocks per SEGMENT: from segcols and nseg alone (identical on every rank, and on every GPU count whose rank
    // count divides 8), 2 pairs (4 elements) per thread
    long long b = ((long long)c->segcols + 4 * kBlock - 1) / (4 * kBlock);
    const long long cap = kMaxPartials / c->nseg;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

int vec_blocks(const bioen_hip_ctx* c) { return vec_grid(c) * c->vr; }

SegMap seg_map(const bioen_hip_ctx* c) { return SegMap{vec_grid(c), c->segcols}; }

void rank_columns(const bioen_hip_ctx* c, int rank, long long* col0, long long* n_local) {
    const long long per = (long long)c->vr * c->segcols;
    *col0 = per * rank;
    long long nl = c->n_global - *col0;
    if (nl > per) nl = per;
    if (nl < 0) nl = 0;
    *n_local = nl;
}

Xch make_xch(const bioen_hip_ctx* c, int stage, int payload) {
    Xch x;
    x.base = c->xbuf[stage];
    x.payload = payload;
    x.world = c->nseg;
    x.rank = c->seg0;
    x.npl = vec_grid(c);
    x.vr = c->vr;
    x.segcols = c->segcols;
    return x;
}

int rows_grid(const bioen_hip_ctx* c) {
    int b = c->mp / kWaves;
    if (b > kMaxPartials) b = kMaxPartials;
    return b;
}



// ---- forward ---------------------------------------------------------------------------
template <int K, int STEPS, bool NT, bool CENTER>
static void fwd_launch(bioen_hip_ctx* c, const Vec8& v) {
    dim3 grid(c->mp / kRowAlign, c->fwd_ctiles);
    BIOEN_LAUNCH_TIMED(c, (k_fwd_partial<8, K, STEPS, NT, CENTER>), grid, dim3(kBlock), 0, c->Y, c->ld, v,
                       c->ybar_c, c->fwd_partial, c->fwd_ctiles, c->fwd_steps, c->fwd_ctiles / c->vr, c->segcols / 128);
}

// STEPS = 0: software-pipelined (next step's Y rows in flight during the FMAs).  Measured on the
// N = 1e6 x M = 1024 sweep (rocprofv3, r01): 0.6-1.7 % faster than the plain loop for K <= 6; at
// K = 7, 8 the second register set drops the occupancy to one wave per SIMD and it loses 4-9 %.
template <bool NT, bool CENTER>
static void fwd_dispatch(bioen_hip_ctx* c, int K, const Vec8& v) {
    switch (K) {
        case 1: fwd_launch<1, 0, NT, CENTER>(c, v); break;
        case 2: fwd_launch<2, 0, NT, CENTER>(c, v); break;
        case 3: fwd_launch<3, 0, NT, CENTER>(c, v); break;
        case 4: fwd_launch<4, 0, NT, CENTER>(c, v); break;
        case 5: fwd_launch<5, 0, NT, CENTER>(c, v); break;
        case 6: fwd_launch<6, 0, NT, CENTER>(c, v); break;
        case 7: fwd_launch<7, 1, NT, CENTER>(c, v); break;
        default: fwd_launch<8, 1, NT, CENTER>(c, v); break;
    }
}

void launch_fwd_partial(bioen_hip_ctx* c, int K, const Vec8& v, bool centred) {
    TimedLaunch tl(c, 0, K);
    if (c->nontemporal) {
        if (centred) fwd_dispatch<true, true>(c, K, v); else fwd_dispatch<true, false>(c, K, v);
    } else {
        if (centred) fwd_dispatch<false, true>(c, K, v); else fwd_dispatch<false, false>(c, K, v);
    }
}

int ybar_payload(const bioen_hip_ctx* c, int K, bool logw) { return c->mp * K + (logw ? 3 * K : 0); }

void launch_fwd_rows_local(bioen_hip_ctx* c, int K, bool logw, int ctiles, bool tposed) {
    const Xch xo = make_xch(c, X_YBAR, ybar_payload(c, K, logw));
    const Xch xe = make_xch(c, X_EXP, 3 * K * vec_grid(c));
    if (tposed) {          // the strip kernels' sets (kernels_strip.hip: strip_sets)
        const StripSets ss = strip_sets(c);
        const int ntile = (c->mp * K + 31) / 32;
        if (logw)
            hipLaunchKernelGGL(k_fwd_rows_local_t<true>, dim3(ntile + K, c->vr), dim3(kBlock), 0, c->stream, c->fwd_partial,
                               ss.gs, ss.fold ? 1 : ss.nch, c->mp, K, ntile, xo, xe);
        else
            hipLaunchKernelGGL(k_fwd_rows_local_t<false>, dim3(ntile, c->vr), dim3(kBlock), 0, c->stream, c->fwd_partial,
                               ss.gs, ss.fold ? 1 : ss.nch, c->mp, K, ntile, xo, xe);
        return;
    }
    const int ct = ctiles > 0 ? ctiles : c->fwd_ctiles;
    if (logw)
        hipLaunchKernelGGL(k_fwd_rows_local<true>, dim3(rows_grid(c), K, c->vr), dim3(kBlock), 0, c->stream, c->fwd_partial,
                           ct, ct / c->vr, c->mp, K, xo, xe);
    else
        hipLaunchKernelGGL(k_fwd_rows_local<false>, dim3(rows_grid(c), K, c->vr), dim3(kBlock), 0, c->stream,
                           c->fwd_partial, ct, ct / c->vr, c->mp, K, xo, xe);
}

// w = e * scal[S_INV]: the weights themselves are only needed when a result is handed out
// The VALID entries only: the padding of e (columns n .. ld) is zero and must stay zero whatever the factor -- the matrix
// passes multiply it with the zero columns of the strip copies, and k_logw_exp never rewrites it: scaled by a
// non-finite 1 / sum e (a run on NaN input) it turned into NaN for good and 0 x NaN poisoned every later evaluation on
// the context (r04, found by tools/attic/nan_probe.py's successor in tests/test_hip_edgecases.py).
__global__ __launch_bounds__(kBlock) void k_scale_w(Round r, int n, SegMap sm) {
    const int a = blockIdx.y;
    const SegPos sp = seg_pos(sm.npl, sm.segcols, n);
    const double inv = r.scal[a][S_INV + sp.v];      // the segment's own shift
    double* __restrict__ w = r.w[a];
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(sm.npl)) {
        d2 v = *reinterpret_cast<d2*>(w + j);
        v.x *= inv;
        v.y = (j + 1 < sp.jend) ? v.y * inv : 0.0;
        *reinterpret_cast<d2*>(w + j) = v;
    }
}

void launch_scale_w(bioen_hip_ctx* c, const Round& r) {
    hipLaunchKernelGGL(k_scale_w, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->n, seg_map(c));
}

int combine_grid(const bioen_hip_ctx*) { return 1; }

void launch_rows_combine(bioen_hip_ctx* c, const Round& r, bool logw, const double* center, bool store_raw,
                         const DevGate* gatep) {
    MVec8 part;
    for (int a = 0; a < kMaxBatch; ++a) part.p[a] = a < r.n ? r.part[a] : nullptr;
    DevGate gate{};
    if (gatep) gate = *gatep;
    if (logw)
        hipLaunchKernelGGL(k_rows_combine<true>, dim3(1, r.n), dim3(kBlock), 0, c->stream,
                           make_xch(c, X_YBAR, ybar_payload(c, r.n, true)), c->mp, r.n, c->YT, c->row_offset, c->row_scale,
                           center, store_raw, c->ybar_c, c->r_c, part, r, gate);
    else
        hipLaunchKernelGGL(k_rows_combine<false>, dim3(1, r.n), dim3(kBlock), 0, c->stream,
                           make_xch(c, X_YBAR, c->mp * r.n), c->mp, r.n, c->YT, c->row_offset, c->row_scale,
                           center, store_raw, c->ybar_c, c->r_c, part, r, gate);
}

static MVec8 tsum_parts(const ForcesRound* fr) {
    MVec8 t{};
    if (fr)
        for (int a = 0; a < fr->n; ++a) t.p[a] = fr->part[a];
    return t;
}

void launch_fwd_rows_forces_grad(bioen_hip_ctx* c, int K, int ctiles) {      // streaming fallback (unsharded contexts)
    hipLaunchKernelGGL(k_fwd_rows_forces_grad, dim3(rows_grid(c), K), dim3(kBlock), 0, c->stream, c->fwd_partial,
                       ctiles, c->mp, K, c->gm, static_cast<const double*>(nullptr), MVec8{});
}

// strip passes (M <= 1024) and row panels (M > 1024): every local segment's share of the forces gradient -> its X_YBAR segment; after the
// exchange k_sum_ranks adds the shares in segment order (identical on every rank, and on every GPU count) -> gm
void launch_fwd_rows_forces_grad_share(bioen_hip_ctx* c, int K, int seg_sets, const ForcesRound* tsum, bool tposed) {
    const Xch xo = make_xch(c, X_YBAR, c->mp * K);
    // tposed = false (r05): the sets of the log-weights forward kernel -- the row panels of a matrix taller than 1024
    // rows -- whose groups are `nch` chunk sets each unless the kernel has folded them (kernels.hpp: StripSets)
    int gs = seg_sets, fold = 1;
    if (!tposed) {
        const StripSets ss = strip_sets(c);
        gs = ss.gs;
        fold = ss.fold ? 1 : ss.nch;
    }
    hipLaunchKernelGGL(k_fwd_rows_forces_grad_t, dim3((c->mp * K + 31) / 32, c->vr), dim3(kBlock), 0, c->stream,
                       c->fwd_partial, gs, fold, c->mp, K, xo.base + (size_t)xo.rank * xo.payload, (size_t)xo.payload,
                       tsum ? c->ybar_c : nullptr, tsum_parts(tsum));
}

__global__ __launch_bounds__(kBlock) void k_sum_ranks(Xch xi, int count, double* __restrict__ out) {
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) {
        double s = 0.0;
        for (int r = 0; r < xi.world; ++r) s += xi.base[(size_t)r * xi.payload + i];
        out[i] = s;
    }
}

void launch_forces_grad_sum_ranks(bioen_hip_ctx* c, int K) {
    const int count = c->mp * K;
    hipLaunchKernelGGL(k_sum_ranks, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream,
                       make_xch(c, X_YBAR, count), count, c->gm);
}

// ---- adjoint ---------------------------------------------------------------------------
template <int K, bool NT, bool CENTER>
static void adj_launch(bioen_hip_ctx* c, const double* u_c, const MVec8& out) {
    dim3 grid((unsigned)(c->ld / 128));
    BIOEN_LAUNCH_TIMED(c, (k_adj<8, K, NT, CENTER, (K >= 4 && K <= 7)>), grid, dim3(kBlock), 0, c->Y, c->ld,
                       c->mp / kWaves, u_c, c->ybar_c, out);
}

template <bool NT, bool CENTER>
static void adj_dispatch(bioen_hip_ctx* c, int K, const double* u_c, const MVec8& out) {
    switch (K) {
        case 1: adj_launch<1, NT, CENTER>(c, u_c, out); break;
        case 2: adj_launch<2, NT, CENTER>(c, u_c, out); break;
        case 3: adj_launch<3, NT, CENTER>(c, u_c, out); break;
        case 4: adj_launch<4, NT, CENTER>(c, u_c, out); break;
        case 5: adj_launch<5, NT, CENTER>(c, u_c, out); break;
        case 6: adj_launch<6, NT, CENTER>(c, u_c, out); break;
        case 7: adj_launch<7, NT, CENTER>(c, u_c, out); break;
        default: adj_launch<8, NT, CENTER>(c, u_c, out); break;
    }
}

void launch_adj(bioen_hip_ctx* c, int K, const double* u_c, const MVec8& out, bool centred) {
    TimedLaunch tl(c, 1, K);
    if (c->nontemporal) {
        if (centred) adj_dispatch<true, true>(c, K, u_c, out); else adj_dispatch<true, false>(c, K, u_c, out);
    } else {
        if (centred) adj_dispatch<false, true>(c, K, u_c, out); else adj_dispatch<false, false>(c, K, u_c, out);
    }
}


}  // namespace bioen
