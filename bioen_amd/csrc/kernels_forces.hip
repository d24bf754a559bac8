// Forces method: N-vector kernels of the streaming path and the LDS-strip passes (gfx950).
#include "device_utils.hpp"

namespace bioen {

// ------------------------------------------------------------------------------
// forces N-vector kernels (blockIdx.y = batch position; xj / b live in the slot's `a`,
// t in the slot's `d`, which the forces method does not otherwise use)
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_forces_max(ForcesRound r, int n) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double* __restrict__ v = r.a[a];
    double mx = -DBL_MAX;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) mx = fmax(mx, v[j]);
    mx = block_max(mx, sh);
    if (threadIdx.x == 0) r.part[a][(size_t)P_MAX * kMaxPartials + blockIdx.x] = mx;
}

// _get_weights_from_forces (c_bioen_kernels_forces.c:152-176), first half
__global__ __launch_bounds__(kBlock) void k_forces_exp(ForcesRound r, const double* __restrict__ w0, int n, int np) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double* __restrict__ xj = r.a[a];
    double* __restrict__ w = r.w[a];
    double* pa = r.part[a];
    const double xmax = max_partials(pa + (size_t)P_MAX * kMaxPartials, np, sh);
    double s = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const double ev = w0[j] * exp(xj[j] - xmax);
        w[j] = ev;
        s += ev;
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) pa[(size_t)P_SUM * kMaxPartials + blockIdx.x] = s;
}

// second half + relative entropy terms (c_bioen_kernels_forces.c:246-258)
__global__ __launch_bounds__(kBlock) void k_forces_norm(ForcesRound r, const double* __restrict__ w0, int n, int np) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    double* __restrict__ w = r.w[a];
    double* pa = r.part[a];
    const double inv = 1.0 / sum_partials(pa + (size_t)P_SUM * kMaxPartials, np, sh);
    double kl = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const double wv = inv * w[j];
        w[j] = wv;
        const double w0v = w0[j];
        if (wv >= DBL_MIN && w0v >= DBL_MIN) kl = fma(log(wv) - log(w0v), wv, kl);
    }
    kl = block_sum(kl, sh);
    if (threadIdx.x == 0) pa[(size_t)P_KL * kMaxPartials + blockIdx.x] = kl;
}

// t_j = (theta (1 + log w_j - log w0_j) + b_j) w_j     (c_bioen_kernels_forces.c:320-328)
// tpart != nullptr: the block's share of T = sum_j t_j as well (row panels, M > 1024: the last matrix pass forms
// sum_j (Y_ij - c_i) t_j and the gradient takes (ybar_i - c_i) T off, as the strip passes do)
__global__ __launch_bounds__(kBlock) void k_forces_t(ForcesRound r, const double* __restrict__ w0, int n, double* __restrict__ tpart) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double* __restrict__ w = r.w[a];
    const double* __restrict__ b = r.a[a];
    double* __restrict__ t = r.t[a];
    const double theta = r.theta[a];
    double s = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const double wv = w[j], w0v = w0[j];
        double dd = 1.0;
        if (wv >= DBL_MIN && w0v >= DBL_MIN) dd += log(wv) - log(w0v);
        const double tv = (dd * theta + b[j]) * wv;
        t[j] = tv;
        s += tv;
    }
    if (tpart) {
        s = block_sum(s, sh);
        if (threadIdx.x == 0) tpart[(size_t)a * gridDim.x + blockIdx.x] = s;
    }
}

// the blocks' shares in block order -> share 0 of the `sets` P_KL shares k_fwd_rows_forces_grad_t totals; the others zero
__global__ __launch_bounds__(kBlock) void k_forces_tsum(const double* __restrict__ tpart, int nblk, ForcesRound r, int sets) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += kBlock) s += tpart[(size_t)a * nblk + b];
    s = block_sum(s, sh);
    double* share = r.part[a] + (size_t)P_KL * kMaxPartials;
    for (int b = threadIdx.x; b < sets; b += kBlock) share[b] = b == 0 ? s : 0.0;
}

// ------------------------------------------------------------------------------
// forces gradient in ONE matrix pass for M <= 512 (F3, c_bioen_kernels_forces.c:280-340):
//     b_j = sum_i Y_ij r_i ;  t_j = (theta (1 + log w_j/w0_j) + b_j) w_j ;  grad_i = sum_j (Y_ij - ybar_i) t_j
// b_j needs a whole column and grad_i a whole row, so a block keeps a STRIP of all mp rows x 16
// columns (128-byte row segments, 68 KB at mp = 512) in LDS -- two blocks per CU -- and walks the
// strips of its share of the columns:
//   stash   : the strip prefetched into registers during the previous strip's work goes to LDS
//   phase 1 : thread t owns rows t and t+256: products Y_ic r_i for 4 columns x K problems at a
//             time, reduced over the wave by one transposing butterfly, over the 4 waves through LDS
//   phase 2 : 16 x K threads turn the column sums into t_c
//   phase 3 : the same two rows per thread: grad_i += sum_c (Y_ic - ybar_i) t_c, kept in registers
// The reference spends two full passes on this (3 of its 5), the unfused device path two of four.
// Output: partial[(row*K + a) * nblk + block], finished by k_fwd_rows_forces_grad.
// ------------------------------------------------------------------------------
constexpr int kOldStripCols = 16;

template <int WAVES, class F>
__device__ __forceinline__ double sum_waves(F f) {          // fixed pairing, 4 or 8 waves
    if constexpr (WAVES == 4) return (f(0) + f(1)) + (f(2) + f(3));
    else return ((f(0) + f(1)) + (f(2) + f(3))) + ((f(4) + f(5)) + (f(6) + f(7)));
}

// THREADS = 256: up to 512 rows, two 70-KB blocks per CU; THREADS = 512: up to 1024 rows, one
// 148-KB block per CU.  Either way a thread owns two rows, 2 waves per SIMD, 128 KB per CU in flight.
template <int K, bool NT, int THREADS>
__global__ __launch_bounds__(THREADS, 2) void k_forces_xy(const double* __restrict__ Y, size_t ld, int mp, int nstrips,
                                                         int n, const double* __restrict__ f_c, ForcesRound fr,
                                                         const double* __restrict__ w0,
                                                         double* __restrict__ partial, int nblk) {
    constexpr int C = kOldStripCols;
    constexpr int KP = next_pow2(K);
    constexpr int CB = KP >= 8 ? 2 : 4;
    constexpr int NV = CB * KP;
    constexpr int SHIFT = (NV == 4) ? 4 : (NV == 8) ? 3 : 2;
    constexpr int ROWS = 2 * THREADS;
    constexpr int WAVES = THREADS / 64;
    constexpr int PIECES = ROWS * (C / 2) / THREADS;
    __shared__ double tile[ROWS][C + 1];
    __shared__ double red[WAVES][C][K];
    __shared__ double xs[C][K];       // x of the strip, then e
    __shared__ double scale[K];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int row0 = t, row1 = t + THREADS;
    const bool has0 = row0 < mp, has1 = row1 < mp;

    double f0[K], f1[K], acc0[K], acc1[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        f0[k] = has0 ? f_c[(size_t)row0 * K + k] : 0.0;
        f1[k] = has1 ? f_c[(size_t)row1 * K + k] : 0.0;
        acc0[k] = 0.0;
        acc1[k] = 0.0;
    }
    double m_run = -DBL_MAX, zacc = 0.0, pxacc = 0.0;           // live in the threads t < C*K
    double* ak = fr.a[0];                                       // this thread's problem t % K: chosen by comparison, once
#pragma unroll
    for (int k = 1; k < K; ++k)
        if (t % K == k) ak = fr.a[k];
    d2 pre[PIECES];
    auto fetch = [&](int strip) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int p = t + THREADS * i, row = p >> 3, part = p & 7;
            // rows beyond mp re-read the last row (their operands f / r are zero, their sums are never stored):
            // no branch around a load, so that the compiler can count the loads in flight (kernels_strip.hip, "pitfall")
            pre[i] = ldg2<NT>(Y + (size_t)(row < mp ? row : mp - 1) * ld + (size_t)strip * C + part * 2);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int p = t + THREADS * i, row = p >> 3, part = p & 7;
            tile[row][part * 2] = pre[i].x;
            tile[row][part * 2 + 1] = pre[i].y;
        }
    };
    int s = blockIdx.x;
    fetch(s < nstrips ? s : 0);
    for (; s < nstrips; s += gridDim.x) {
        __syncthreads();
        stash();
        __syncthreads();
        double w0v = 0.0;                                       // before the prefetch (vmcnt retires in order)
        const size_t col = (size_t)s * C + t / K;
        if (t < C * K) w0v = w0[col];
        fetch(s + (int)gridDim.x < nstrips ? s + (int)gridDim.x : s);         // unconditional: the tail re-reads its own strip
        // ---- phase 1: x_c = sum_i Y_ic f_i ----
#pragma unroll
        for (int q = 0; q < C / CB; ++q) {
            double v[NV];
#pragma unroll
            for (int cc = 0; cc < CB; ++cc) {
                const double y0 = tile[row0][CB * q + cc], y1 = tile[row1 & (ROWS - 1)][CB * q + cc];
#pragma unroll
                for (int k = 0; k < KP; ++k) v[cc * KP + k] = k < K ? fma(y1, f1[k < K ? k : 0], y0 * f0[k < K ? k : 0]) : 0.0;
            }
            wave_multi_reduce<NV>(v, lane);
            if ((lane & ((1 << SHIFT) - 1)) == 0) {
                const int idx = lane >> SHIFT, cc = idx / KP, k = idx % KP;
                if (k < K) red[wave][CB * q + cc][k] = v[0];
            }
        }
        __syncthreads();
        // ---- phase 2: x out; running maximum; e = w0 exp(x - m) ----
        double x = 0.0;
        bool valid = false;
        if (t < C * K) {
            const int c = t / K, k = t % K;
            x = sum_waves<WAVES>([&](int wv_) { return red[wv_][c][k]; });
            valid = col < (size_t)n;
            ak[col] = valid ? x : 0.0;
            xs[c][k] = valid ? x : -DBL_MAX;
        }
        __syncthreads();
        double e = 0.0;
        if (t < C * K) {
            const int c = t / K, k = t % K;
            double smax = -DBL_MAX;
#pragma unroll
            for (int cc = 0; cc < C; ++cc) smax = fmax(smax, xs[cc][k]);
            const double m_new = fmax(m_run, smax);
            const double sc = exp(m_run - m_new);               // 1 when the maximum stands, 0 the first time
            e = valid ? w0v * exp(x - m_new) : 0.0;
            zacc = fma(zacc, sc, e);
            pxacc = fma(pxacc, sc, valid ? e * x : 0.0);
            m_run = m_new;
            if (c == 0) scale[k] = sc;
        }
        __syncthreads();                                        // everyone has read xs
        if (t < C * K) xs[t / K][t % K] = e;
        __syncthreads();
        // ---- phase 3: ybar_raw_i = ybar_raw_i * scale + sum_c Y_ic e_c ----
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double sc = scale[k];
            acc0[k] *= sc;
            acc1[k] *= sc;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double y0 = tile[row0][c], y1 = tile[row1 & (ROWS - 1)][c];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double ek = xs[c][k];
                acc0[k] = fma(y0, ek, acc0[k]);
                acc1[k] = fma(y1, ek, acc1[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (has0) partial[((size_t)row0 * K + k) * nblk + blockIdx.x] = acc0[k];
        if (has1) partial[((size_t)row1 * K + k) * nblk + blockIdx.x] = acc1[k];
    }
    // block statistics per problem: shift, sum e, sum e x  (the 16 column threads of a problem hold the same shift)
    __syncthreads();
    if (t < C * K) {
        red[0][t / K][t % K] = zacc;
        red[1][t / K][t % K] = pxacc;
    }
    __syncthreads();
    if (t < K) {
        double z = 0.0, px = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            z += red[0][c][t];
            px += red[1][c][t];
        }
        double* pa = fr.part[0];
#pragma unroll
        for (int k = 1; k < K; ++k)
            if (t == k) pa = fr.part[k];
        pa[(size_t)P_MAX * kMaxPartials + blockIdx.x] = m_run;   // thread t = (c 0, k t)
        pa[(size_t)P_SUM * kMaxPartials + blockIdx.x] = z;
        pa[(size_t)P_PP * kMaxPartials + blockIdx.x] = px;
    }
}

template <int K, bool NT, bool FROMX, int THREADS>
__global__ __launch_bounds__(THREADS, 2) void k_forces_bt(const double* __restrict__ Y, size_t ld, int mp, int nstrips,
                                                      const double* __restrict__ r_c,
                                                      const double* __restrict__ ybar_c, ForcesRound fr,
                                                      const double* __restrict__ w0,
                                                      double* __restrict__ partial, int nblk) {
    constexpr int C = kOldStripCols;
    constexpr int KP = next_pow2(K);
    constexpr int CB = KP >= 8 ? 2 : 4;                         // columns per butterfly
    constexpr int NV = CB * KP;                                 // values per butterfly (4 .. 16)
    constexpr int SHIFT = (NV == 4) ? 4 : (NV == 8) ? 3 : 2;    // 6 - log2(NV)
    constexpr int ROWS = 2 * THREADS;
    constexpr int WAVES = THREADS / 64;
    constexpr int PIECES = ROWS * (C / 2) / THREADS;       // 16-byte pieces per thread and strip: 16
    __shared__ double tile[ROWS][C + 1];                  // +1: rows 17 doubles apart, conflict-free columns
    __shared__ double red[WAVES][C][K];
    __shared__ double tv[C][K];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int row0 = t, row1 = t + THREADS;
    const bool has0 = row0 < mp, has1 = row1 < mp;

    double r0[K], r1[K], yb0[K], yb1[K], acc0[K], acc1[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        r0[k] = has0 ? r_c[(size_t)row0 * K + k] : 0.0;
        r1[k] = has1 ? r_c[(size_t)row1 * K + k] : 0.0;
        yb0[k] = has0 ? ybar_c[(size_t)row0 * K + k] : 0.0;
        yb1[k] = has1 ? ybar_c[(size_t)row1 * K + k] : 0.0;
        acc0[k] = 0.0;
        acc1[k] = 0.0;
    }
    const double* ak = fr.a[0];                                 // this thread's problem t % K: chosen by comparison, once
    const double* wk = fr.w[0];
    double logsk = 0.0, thk = fr.theta[0];
    {
        const double* sck = fr.scal[0];
#pragma unroll
        for (int k = 1; k < K; ++k)
            if (t % K == k) {
                ak = fr.a[k];
                wk = fr.w[k];
                sck = fr.scal[k];
                thk = fr.theta[k];
            }
        if (FROMX && t < C * K) logsk = sck[S_LOGS];
    }
    d2 pre[PIECES];
    auto fetch = [&](int strip) {                               // global -> registers, 8 lanes per row segment
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int p = t + THREADS * i, row = p >> 3, part = p & 7;
            // rows beyond mp re-read the last row (their operands f / r are zero, their sums are never stored):
            // no branch around a load, so that the compiler can count the loads in flight (kernels_strip.hip, "pitfall")
            pre[i] = ldg2<NT>(Y + (size_t)(row < mp ? row : mp - 1) * ld + (size_t)strip * C + part * 2);
        }
    };
    auto stash = [&]() {                                        // registers -> LDS
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int p = t + THREADS * i, row = p >> 3, part = p & 7;
            tile[row][part * 2] = pre[i].x;
            tile[row][part * 2 + 1] = pre[i].y;
        }
    };
    int s = blockIdx.x;
    fetch(s < nstrips ? s : 0);
    for (; s < nstrips; s += gridDim.x) {
        __syncthreads();                                        // phase 3 of the previous strip is done with the tile
        stash();
        __syncthreads();
        // phase 2's operands first, THEN the prefetch: vmcnt retires in order, so waiting for a
        // load issued after the prefetch would wait for the whole next strip as well
        double wv = 0.0, w0v = 0.0, lr = 0.0;                   // lr = log(w / w0)
        if (t < C * K) {
            const size_t col = (size_t)s * C + t / K;           // < ld; padded columns carry w0 = w = 0
            w0v = w0[col];
            if (FROMX) {                                        // weights from x (k_forces_xy): no log needed
                lr = ak[col] - logsk;
                wv = w0v * exp(lr);
            } else {
                wv = wk[col];
            }
        }
        fetch(s + (int)gridDim.x < nstrips ? s + (int)gridDim.x : s);         // unconditional: the tail re-reads its own strip // in flight during the three phases
        // ---- phase 1 ----
#pragma unroll
        for (int q = 0; q < C / CB; ++q) {
            double v[NV];
#pragma unroll
            for (int cc = 0; cc < CB; ++cc) {
                const double y0 = tile[row0][CB * q + cc], y1 = tile[row1 & (ROWS - 1)][CB * q + cc];
#pragma unroll
                for (int k = 0; k < KP; ++k) v[cc * KP + k] = k < K ? fma(y1, r1[k < K ? k : 0], y0 * r0[k < K ? k : 0]) : 0.0;
            }
            wave_multi_reduce<NV>(v, lane);
            if ((lane & ((1 << SHIFT) - 1)) == 0) {
                const int idx = lane >> SHIFT, cc = idx / KP, k = idx % KP;
                if (k < K) red[wave][CB * q + cc][k] = v[0];
            }
        }
        __syncthreads();
        // ---- phase 2 ----
        if (t < C * K) {
            const int c = t / K, k = t % K;
            const double b = sum_waves<WAVES>([&](int wv_) { return red[wv_][c][k]; });
            double dd = 1.0;
            if (wv >= DBL_MIN && w0v >= DBL_MIN) dd += FROMX ? lr : log(wv) - log(w0v);
            tv[c][k] = (dd * thk + b) * wv;
        }
        __syncthreads();
        // ---- phase 3 ----
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double y0 = tile[row0][c], y1 = tile[row1 & (ROWS - 1)][c];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double tk = tv[c][k];
                acc0[k] = fma(y0 - yb0[k], tk, acc0[k]);
                acc1[k] = fma(y1 - yb1[k], tk, acc1[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (has0) partial[((size_t)row0 * K + k) * nblk + blockIdx.x] = acc0[k];
        if (has1) partial[((size_t)row1 * K + k) * nblk + blockIdx.x] = acc1[k];
    }
}

__global__ __launch_bounds__(kBlock) void k_forces_scalars(ForcesRound r, int npchi, int npkl) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double* pa = r.part[a];
    const double chi = sum_partials(pa + (size_t)P_CHI * kMaxPartials, npchi, sh);
    const double kl = sum_partials(pa + (size_t)P_KL * kMaxPartials, npkl, sh);
    if (threadIdx.x == 0) {
        double* sc = r.scal[a];
        sc[S_CHI] = chi;
        sc[S_KL] = kl;
        sc[S_F] = kl * r.theta[a] + 0.5 * chi;
    }
}


// ---- forces evaluation over LDS-resident column strips (M <= 1024) --------------------------
static int old_strip_threads(const bioen_hip_ctx* c) { return c->mp <= 512 ? 256 : 512; }

int forces_fused_blocks_old(const bioen_hip_ctx* c) {
    if (c->mp > 1024) return 0;
    const int nstrips = (int)(c->ld / kOldStripCols);
    // 256 threads: two 70-KB blocks per CU; 512 threads: one 148-KB block per CU
    return std::min(old_strip_threads(c) == 256 ? 512 : 256, nstrips);
}

template <int K, bool NT, int THREADS>
static void forces_strip_launch(bioen_hip_ctx* c, const ForcesRound& fr, int nblk, int pass) {
    const int nstrips = (int)(c->ld / kOldStripCols);
    if (pass == 1)
        BIOEN_LAUNCH_TIMED(c, (k_forces_xy<K, NT, THREADS>), dim3(nblk), dim3(THREADS), 0, c->Y, c->ld, c->mp,
                           nstrips, c->n, c->um, fr, c->fixed, c->fwd_partial, nblk);
    else
        BIOEN_LAUNCH_TIMED(c, (k_forces_bt<K, NT, true, THREADS>), dim3(nblk), dim3(THREADS), 0, c->Y, c->ld,
                           c->mp, nstrips, c->r_c, c->ybar_c, fr, c->fixed, c->fwd_partial, nblk);
}

template <bool NT, int THREADS>
static void forces_strip_dispatch_k(bioen_hip_ctx* c, const ForcesRound& fr, int nblk, int pass) {
    switch (fr.n) {
        case 1: forces_strip_launch<1, NT, THREADS>(c, fr, nblk, pass); break;
        case 2: forces_strip_launch<2, NT, THREADS>(c, fr, nblk, pass); break;
        case 3: forces_strip_launch<3, NT, THREADS>(c, fr, nblk, pass); break;
        case 4: forces_strip_launch<4, NT, THREADS>(c, fr, nblk, pass); break;
        case 5: forces_strip_launch<5, NT, THREADS>(c, fr, nblk, pass); break;
        case 6: forces_strip_launch<6, NT, THREADS>(c, fr, nblk, pass); break;
        case 7: forces_strip_launch<7, NT, THREADS>(c, fr, nblk, pass); break;
        default: forces_strip_launch<8, NT, THREADS>(c, fr, nblk, pass); break;
    }
}

template <bool NT>
static void forces_strip_dispatch(bioen_hip_ctx* c, const ForcesRound& fr, int nblk, int pass) {
    if (old_strip_threads(c) == 256) forces_strip_dispatch_k<NT, 256>(c, fr, nblk, pass);
    else forces_strip_dispatch_k<NT, 512>(c, fr, nblk, pass);
}

// pass 1: x = yTilde^T f, online softmax, raw ybar per block; then the block merge and ybar -> X_YBAR
void launch_forces_xy_old(bioen_hip_ctx* c, const ForcesRound& fr, int nblk) {
    {
        TimedLaunch tl(c, 1, fr.n);
        if (c->nontemporal) forces_strip_dispatch<true>(c, fr, nblk, 1); else forces_strip_dispatch<false>(c, fr, nblk, 1);
    }
    launch_forces_blockmerge(c, fr, nblk);
}

// pass 2: b = yTilde^T r, t, centred yTilde . t
void launch_forces_bt_old(bioen_hip_ctx* c, const ForcesRound& fr, int nblk) {
    TimedLaunch tl(c, 0, fr.n);
    if (c->nontemporal) forces_strip_dispatch<true>(c, fr, nblk, 2); else forces_strip_dispatch<false>(c, fr, nblk, 2);
}

// ---- forces ------------------------------------------------------------------------------------
void launch_forces_max(bioen_hip_ctx* c, const ForcesRound& r) {
    hipLaunchKernelGGL(k_forces_max, dim3(vec_grid(c), r.n), dim3(kBlock), 0, c->stream, r, c->n);
}

void launch_forces_exp(bioen_hip_ctx* c, const ForcesRound& r) {
    hipLaunchKernelGGL(k_forces_exp, dim3(vec_grid(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                       vec_grid(c));
}

void launch_forces_norm(bioen_hip_ctx* c, const ForcesRound& r) {
    hipLaunchKernelGGL(k_forces_norm, dim3(vec_grid(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                       vec_grid(c));
}

void launch_forces_t(bioen_hip_ctx* c, const ForcesRound& r, int tsum_sets) {
    if (tsum_sets > 0) {
        const int g = vec_grid(c);
        double* tpart = c->xbuf[X_GRAM];            // kGramDots values per block and problem: room for one; idle in the forces method
        hipLaunchKernelGGL(k_forces_t, dim3(g, r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n, tpart);
        hipLaunchKernelGGL(k_forces_tsum, dim3(r.n), dim3(kBlock), 0, c->stream, tpart, g, r, tsum_sets);
        return;
    }
    hipLaunchKernelGGL(k_forces_t, dim3(vec_grid(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                       static_cast<double*>(nullptr));
}

void launch_forces_scalars(bioen_hip_ctx* c, const ForcesRound& r) {
    hipLaunchKernelGGL(k_forces_scalars, dim3(1, r.n), dim3(kBlock), 0, c->stream, r, combine_grid(c), vec_grid(c));
}


// ---- the round's results to the host without a copy engine and without a stream synchronisation (r03) -------------
// page = [gradients, compact mp x k | scalars of all slots | flag]: one block copies both arrays into the host-mapped,
// coherent page, every thread fences its own stores, the last instruction publishes the round number.  The host spins
// on the flag (ForcesBatchEngine::evaluate): a round used to end with two device-to-host copies and
// hipStreamSynchronize.
__global__ __launch_bounds__(kBlock) void k_forces_publish(const double* __restrict__ gm, int ngrad,
                                                           const double* __restrict__ scal, int nscal,
                                                           double* __restrict__ page, int scal_at,
                                                           unsigned long long* __restrict__ flag, unsigned long long round) {
    for (int i = threadIdx.x; i < ngrad; i += kBlock) page[i] = gm[i];
    for (int i = threadIdx.x; i < nscal; i += kBlock) page[scal_at + i] = scal[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

void launch_forces_publish(bioen_hip_ctx* c, int ngrad, unsigned long long round) {
    const int scal_at = c->mp * kMaxBatch;
    unsigned long long* flag = reinterpret_cast<unsigned long long*>(c->live_f + scal_at + kMaxBatch * kScalStride);
    hipLaunchKernelGGL(k_forces_publish, dim3(1), dim3(kBlock), 0, c->stream, c->gm, ngrad, c->scal, kMaxBatch * kScalStride,
                       c->live_f, scal_at, flag, round);
}

}  // namespace bioen
