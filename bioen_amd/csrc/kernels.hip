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
#include "kernels.hpp"

#include <cfloat>

namespace bioen {

typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

// ------------------------------------------------------------------------------
// reductions (fixed order => deterministic)
// ------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;   // every lane holds the sum
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// Butterfly-reduce NV (power of two, <= 64) per-lane values at once: at every stage the two
// lanes of a pair split the remaining values between them, so the whole thing costs NV-1
// shuffles instead of 6*NV.  Each value's sum is formed in the same pair order
// (32,16,8,4,2,1) as wave_sum, hence bitwise equal to it.  On return v[0] of lane l is the
// total of value number  l >> (6 - log2 NV).
template <int CNT, int O, int NV>
__device__ __forceinline__ void multi_stage(double (&v)[NV], int lane) {
    if constexpr (O >= 1) {
        if constexpr (CNT > 1) {
            const bool upper = (lane & O) != 0;
            constexpr int half = CNT / 2;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const double send = upper ? v[i] : v[i + half];
                const double keep = upper ? v[i + half] : v[i];
                v[i] = keep + __shfl_xor(send, O, 64);
            }
            multi_stage<half, O / 2, NV>(v, lane);
        } else {
            v[0] += __shfl_xor(v[0], O, 64);
            multi_stage<1, O / 2, NV>(v, lane);
        }
    }
}

template <int NV>
__device__ __forceinline__ void wave_multi_reduce(double (&v)[NV], int lane) {
    multi_stage<NV, 32, NV>(v, lane);
}

// sum over the 256 threads of a block; result in every thread
__device__ __forceinline__ double block_sum(double v, double* sh /* [kWaves] */) {
    v = wave_sum(v);
    __syncthreads();   // protect sh against the previous use
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__device__ __forceinline__ double block_max(double v, double* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

// Sum an array of per-block partials written by the PREVIOUS kernel.  Every block of the
// consumer kernel does this redundantly in its prologue (<= 8 KiB, L2 resident), which
// replaces a separate 1-block "finalise" launch.
__device__ __forceinline__ double sum_partials(const double* __restrict__ p, int np, double* sh) {
    double s = 0.0;
    for (int k = threadIdx.x; k < np; k += kBlock) s += p[k];
    return block_sum(s, sh);
}

__device__ __forceinline__ double max_partials(const double* __restrict__ p, int np, double* sh) {
    double s = -DBL_MAX;
    for (int k = threadIdx.x; k < np; k += kBlock) s = fmax(s, p[k]);
    return block_max(s, sh);
}

// ---- exchange-stage accessors (ctx.hpp: XStage).  Layout [rank][problem a][array q][block].
// put: this rank's block partial.  sum/max: over all ranks and blocks in a fixed order
// (rank-major), identical on every rank.
template <int A>
__device__ __forceinline__ void xput(const Xch& x, int a, int q, double v) {
    x.base[(size_t)x.rank * x.payload + (size_t)(a * A + q) * x.npl + blockIdx.x] = v;
}

template <int A>
__device__ __forceinline__ double xsum(const Xch& x, int a, int q, double* sh) {
    double s = 0.0;
    for (int r = 0; r < x.world; ++r) {
        const double* p = x.base + (size_t)r * x.payload + (size_t)(a * A + q) * x.npl;
        for (int k = threadIdx.x; k < x.npl; k += kBlock) s += p[k];
    }
    return block_sum(s, sh);
}

// sum over the blocks of ONE rank's segment
template <int A>
__device__ __forceinline__ double xsum_rank(const Xch& x, int rk, int a, int q, double* sh) {
    double s = 0.0;
    const double* p = x.base + (size_t)rk * x.payload + (size_t)(a * A + q) * x.npl;
    for (int k = threadIdx.x; k < x.npl; k += kBlock) s += p[k];
    return block_sum(s, sh);
}

template <int A>
__device__ __forceinline__ double xmax(const Xch& x, int a, int q, double* sh) {
    double s = -DBL_MAX;
    for (int r = 0; r < x.world; ++r) {
        const double* p = x.base + (size_t)r * x.payload + (size_t)(a * A + q) * x.npl;
        for (int k = threadIdx.x; k < x.npl; k += kBlock) s = fmax(s, p[k]);
    }
    return block_max(s, sh);
}

// maximum over THIS rank's segment only (no exchange needed before it)
template <int A>
__device__ __forceinline__ double xmax_local(const Xch& x, int a, int q, double* sh) {
    double s = -DBL_MAX;
    const double* p = x.base + (size_t)x.rank * x.payload + (size_t)(a * A + q) * x.npl;
    for (int k = threadIdx.x; k < x.npl; k += kBlock) s = fmax(s, p[k]);
    return block_max(s, sh);
}

template <bool NT>
__device__ __forceinline__ d2 ldg2(const double* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const d2*>(p));
    return *reinterpret_cast<const d2*>(p);
}

constexpr int next_pow2(int v) { return v <= 1 ? 1 : (v <= 2 ? 2 : (v <= 4 ? 4 : 8)); }

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
                                                        int steps_per_tile, int total_steps) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // blockIdx.x = row block (fast index): consecutive blocks share the column tile, so the
    // tile's slice of v_a is fetched from HBM once per XCD and then served by L2
    const int tile = blockIdx.y;
    const int row0 = (blockIdx.x * kWaves + wave) * R;
    int s = tile * steps_per_tile;
    int s_end = s + steps_per_tile;
    if (s_end > total_steps) s_end = total_steps;

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

// reduce the column tiles of one (row, problem) per wave (fixed order): this rank's share of
// ybar, written into its segment of the X_YBAR stage (compact layout [row*K + a])
// WITH_EXP (log-weights rounds): block 0 of each problem also totals this rank's softmax partials
// and appends {sum e, sum e (x - G), m_r} to the rank's segment, so the normalisation needs no
// exchange of its own.
template <bool WITH_EXP>
__global__ __launch_bounds__(kBlock) void k_fwd_rows_local(const double* __restrict__ partial, int ctiles,
                                                           int mp, int K, Xch xo, Xch xe) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double* out = xo.base + (size_t)xo.rank * xo.payload;
    if (WITH_EXP && blockIdx.x == 0) {
        const double s = xsum_rank<3>(xe, xe.rank, a, 0, sh);
        const double pp = xsum_rank<3>(xe, xe.rank, a, 1, sh);
        if (threadIdx.x == 0) {
            double* tail = out + (size_t)mp * K + 3 * a;
            tail[0] = s;
            tail[1] = pp;
            tail[2] = xe.base[(size_t)xe.rank * xe.payload + (size_t)(a * 3 + 2) * xe.npl];
        }
    }
    for (int row = blockIdx.x * kWaves + wave; row < mp; row += gridDim.x * kWaves) {
        const double* p = partial + ((size_t)row * K + a) * ctiles;
        double s = 0.0;
        for (int k = lane; k < ctiles; k += 64) s += p[k];
        s = wave_sum(s);
        if (lane == 0) out[(size_t)row * K + a] = s;
    }
}

// add the ranks' shares (rank order) -> ybar, r = ybar - YT (compact) ; per-block partials of
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
                                                         double* __restrict__ ybar_c, double* __restrict__ r_c,
                                                         MVec8 part, Round rd) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    // LOGW: the shares are yTilde . e_r with e_r = exp(x - m_r) unnormalised; global shift
    // M = max_r m_r, S = sum_r e^{m_r - M} S_r, and rank r's share enters with e^{m_r - M} / S
    // (exactly 1 / S on one GPU).  _get_weights' normalisation (c_bioen_kernels_logw.c:84-90) is
    // thereby applied to the M sums instead of the N weights.
    double gmax = 0.0, invS = 1.0;
    if (LOGW) {
        gmax = -DBL_MAX;
        for (int r = 0; r < xi.world; ++r)
            gmax = fmax(gmax, xi.base[(size_t)r * xi.payload + (size_t)mp * K + 3 * a + 2]);
        double S = 0.0, PP = 0.0;
        for (int r = 0; r < xi.world; ++r) {
            const double* tail = xi.base + (size_t)r * xi.payload + (size_t)mp * K + 3 * a;
            const double fr = exp(tail[2] - gmax);
            S = fma(fr, tail[0], S);
            PP = fma(fr, tail[1], PP);
        }
        invS = 1.0 / S;
        if (threadIdx.x == 0) {
            double* sc = rd.scal[a];
            const double mown = xi.base[(size_t)xi.rank * xi.payload + (size_t)mp * K + 3 * a + 2];
            sc[S_LOGS] = gmax + log(S);
            sc[S_P] = PP * invS;
            sc[S_INV] = exp(mown - gmax) * invS;
        }
    }
    double chi = 0.0, cc = 0.0;
    for (int row = threadIdx.x; row < mp; row += kBlock) {
        double s = 0.0;
        for (int r = 0; r < xi.world; ++r) {
            const double v = xi.base[(size_t)r * xi.payload + (size_t)row * K + a];
            if (LOGW) s = fma(exp(xi.base[(size_t)r * xi.payload + (size_t)mp * K + 3 * a + 2] - gmax) * invS, v, s);
            else s += v;
        }
        const double sc = row_scale[row];
        const double eff = fma(sc, s, row_offset[row]);
        const double res = eff - YT[row];
        ybar_c[(size_t)row * K + a] = s;
        r_c[(size_t)row * K + a] = res * sc;
        chi = fma(res, res, chi);
        cc = fma(eff, res, cc);
    }
    chi = block_sum(chi, sh);
    cc = block_sum(cc, sh);
    if (threadIdx.x == 0) {
        if (LOGW) {
            double* sc = rd.scal[a];            // S_P, S_LOGS: written above by this same thread
            sc[S_CHI] = chi;
            sc[S_C] = cc;
            sc[S_KL] = sc[S_P] - sc[S_LOGS] + sc[S_LOGS0];     // theta's factor: KL(w || w0) in both methods
            sc[S_F] = rd.theta[a] * sc[S_KL] + 0.5 * chi;
        } else {
            double* pa = part.p[a];
            pa[(size_t)P_CHI * kMaxPartials] = chi;
            pa[(size_t)P_C * kMaxPartials] = cc;
        }
    }
}

// forces gradient (c_bioen_kernels_forces.c:330-338): the partials already hold the centred
// sums  sum_j (Y_ij - ybar_i) t_j ; only the column tiles remain to be added up.
__global__ __launch_bounds__(kBlock) void k_fwd_rows_forces_grad(const double* __restrict__ partial, int ctiles,
                                                                 int mp, int K, double* __restrict__ gm_c) {
    const int a = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * kWaves + wave; row < mp; row += gridDim.x * kWaves) {
        const double* p = partial + ((size_t)row * K + a) * ctiles;
        double s = 0.0;
        for (int k = lane; k < ctiles; k += 64) s += p[k];
        s = wave_sum(s);
        if (lane == 0) gm_c[(size_t)row * K + a] = s;
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
template <int U, int K, bool NT, bool CENTER>
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
    for (int i = 0; i < rows_per_wave; i += U) {
        d2 y[U];
#pragma unroll
        for (int q = 0; q < U; ++q) y[q] = ldg2<NT>(yp + (size_t)q * ld);
#pragma unroll
        for (int q = 0; q < U; q += 2) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const double u0 = up[(size_t)(i + q) * K + k], u1 = up[(size_t)(i + q + 1) * K + k];
                const double b0 = CENTER ? bp[(size_t)(i + q) * K + k] : 0.0;
                const double b1 = CENTER ? bp[(size_t)(i + q + 1) * K + k] : 0.0;
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

// ------------------------------------------------------------------------------
// log-weights N-vector kernels (blockIdx.y = position a in the round's batch)
// ------------------------------------------------------------------------------
// x = xp + stp * d ; block maxima of x
__global__ __launch_bounds__(kBlock) void k_trial(Round r, int n, Xch xo) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    double* __restrict__ x = r.x[a];
    const double* __restrict__ xp = r.xp[a];
    const double* __restrict__ d = r.d[a];
    const double stp = r.stp[a];
    double mx = -DBL_MAX;
    const int n2 = (n + 1) >> 1;   // 16-byte pairs; vectors are zero-padded to an even length
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < n2; p += gridDim.x * kBlock) {
        const int j = 2 * p;
        const d2 dv = *reinterpret_cast<const d2*>(d + j);
        const d2 pv = *reinterpret_cast<const d2*>(xp + j);
        d2 v = {fma(stp, dv.x, pv.x), fma(stp, dv.y, pv.y)};
        *reinterpret_cast<d2*>(x + j) = v;
        mx = fmax(mx, v.x);
        if (j + 1 < n) mx = fmax(mx, v.y);
    }
    mx = block_max(mx, sh);
    if (threadIdx.x == 0) xput<1>(xo, a, 0, mx);
}

__global__ __launch_bounds__(kBlock) void k_max(Round r, int n, Xch xo) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double* __restrict__ x = r.x[a];
    double mx = -DBL_MAX;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) mx = fmax(mx, x[j]);
    mx = block_max(mx, sh);
    if (threadIdx.x == 0) xput<1>(xo, a, 0, mx);
}

// _get_weights (c_bioen_kernels_logw.c:55-94) with a max shift, first half:
//   e_j = exp(x_j - m_r) ; partials of sum e and sum e (x - G)   (prior, :96-127)
// m_r is the maximum over THIS rank's structures (its own block maxima need no exchange); it
// travels with the sums (third array, entry 0) and k_logw_norm rescales by exp(m_r - max_r m_r),
// which is exactly 1 on a single GPU.
__global__ __launch_bounds__(kBlock) void k_logw_exp(Round r, const double* __restrict__ G, int n, Xch xmx,
                                                     Xch xo) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double* __restrict__ x = r.x[a];
    double* __restrict__ e = r.w[a];
    const double gmax = xmax_local<1>(xmx, a, 0, sh);
    double s = 0.0, pp = 0.0;
    const int n2 = (n + 1) >> 1;
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < n2; p += gridDim.x * kBlock) {
        const int j = 2 * p;
        const d2 xv = *reinterpret_cast<const d2*>(x + j);
        const d2 Gv = *reinterpret_cast<const d2*>(G + j);
        d2 ev;
        ev.x = exp(xv.x - gmax);
        ev.y = (j + 1 < n) ? exp(xv.y - gmax) : 0.0;
        *reinterpret_cast<d2*>(e + j) = ev;
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
        if (blockIdx.x == 0) xput<3>(xo, a, 2, gmax);
    }
}

// second half: w = e / S ; scal[S_LOGS] = max + log S ; scal[S_P] = sum e (x-G) / S
__global__ __launch_bounds__(kBlock) void k_logw_norm(Round r, int n, Xch xe) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    double* __restrict__ w = r.w[a];
    // global shift M = max_r m_r ; S = sum_r e^{m_r - M} S_r   (rank order; e^0 = 1 on one GPU)
    double gmax = -DBL_MAX;
    for (int rk = 0; rk < xe.world; ++rk)
        gmax = fmax(gmax, xe.base[(size_t)rk * xe.payload + (size_t)(a * 3 + 2) * xe.npl]);
    double S = 0.0;
    for (int rk = 0; rk < xe.world; ++rk) {
        const double mr = xe.base[(size_t)rk * xe.payload + (size_t)(a * 3 + 2) * xe.npl];
        S = fma(exp(mr - gmax), xsum_rank<3>(xe, rk, a, 0, sh), S);
    }
    const double mown = xe.base[(size_t)xe.rank * xe.payload + (size_t)(a * 3 + 2) * xe.npl];
    const double inv = exp(mown - gmax) / S;
    if (blockIdx.x == 0) {
        double PP = 0.0;
        for (int rk = 0; rk < xe.world; ++rk) {
            const double mr = xe.base[(size_t)rk * xe.payload + (size_t)(a * 3 + 2) * xe.npl];
            PP = fma(exp(mr - gmax), xsum_rank<3>(xe, rk, a, 1, sh), PP);
        }
        if (threadIdx.x == 0) {
            r.scal[a][S_LOGS] = gmax + log(S);
            r.scal[a][S_P] = PP * (1.0 / S);
        }
    }
    const int n2 = (n + 1) >> 1;
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < n2; p += gridDim.x * kBlock) {
        d2 v = *reinterpret_cast<const d2*>(w + 2 * p);
        v.x *= inv;
        v.y *= inv;
        *reinterpret_cast<d2*>(w + 2 * p) = v;
    }
}

// log s0 = log sum exp(G): constant per problem, computed once (the reference recomputes it
// at every evaluation, c_bioen_kernels_logw.c:122).  One block; every problem of the round
// gets the value.
__global__ __launch_bounds__(kBlock) void k_logsumexp1(const double* __restrict__ G, int n, Round r) {
    __shared__ double sh[kWaves];
    double mx = -DBL_MAX;
    for (int j = threadIdx.x; j < n; j += kBlock) mx = fmax(mx, G[j]);
    mx = block_max(mx, sh);
    double s = 0.0;
    for (int j = threadIdx.x; j < n; j += kBlock) s += exp(G[j] - mx);
    s = block_sum(s, sh);
    if (threadIdx.x == 0) {
        const double v = mx + log(s);
        for (int a = 0; a < r.n; ++a) r.scal[a][S_LOGS0] = v;
    }
}

// gradient epilogue (c_bioen_kernels_logw.c:207-218):
//   g_k = w_k [ theta (x_k - G_k - P) + a_k ],  a_k = sum_i r_i (yTilde_ik - ybar_i)  (centred adjoint)
// plus the three dot products the line search / convergence test needs.
__global__ __launch_bounds__(kBlock) void k_logw_grad(Round r, const double* __restrict__ G, int n, Xch xo) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double* __restrict__ x = r.x[a];
    const double* __restrict__ w = r.w[a];
    const double* __restrict__ av = r.a[a];
    const double* __restrict__ d = r.d[a];
    double* __restrict__ g = r.g[a];
    const double theta = r.theta[a];
    const double P = r.scal[a][S_P];
    const double inv = r.scal[a][S_INV];      // w = e * inv
    double dg = 0.0, gg = 0.0, xx = 0.0;
    const int n2 = (n + 1) >> 1;
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < n2; p += gridDim.x * kBlock) {
        const int j = 2 * p;
        const d2 xv = *reinterpret_cast<const d2*>(x + j);
        d2 wv = *reinterpret_cast<const d2*>(w + j);           // pad: e = 0  =>  g = 0
        wv.x *= inv;
        wv.y *= inv;
        const d2 Gv = *reinterpret_cast<const d2*>(G + j);
        const d2 aa = *reinterpret_cast<const d2*>(av + j);
        const d2 dv = *reinterpret_cast<const d2*>(d + j);
        d2 gv;
        gv.x = wv.x * (theta * ((xv.x - Gv.x) - P) + aa.x);
        gv.y = wv.y * (theta * ((xv.y - Gv.y) - P) + aa.y);
        *reinterpret_cast<d2*>(g + j) = gv;
        dg = fma(gv.x, dv.x, dg);
        gg = fma(gv.x, gv.x, gg);
        xx = fma(xv.x, xv.x, xx);
        dg = fma(gv.y, dv.y, dg);
        gg = fma(gv.y, gv.y, gg);
        xx = fma(xv.y, xv.y, xx);
    }
    dg = block_sum(dg, sh);
    gg = block_sum(gg, sh);
    xx = block_sum(xx, sh);
    if (threadIdx.x == 0) {
        xput<3>(xo, a, 0, dg);
        xput<3>(xo, a, 1, gg);
        xput<3>(xo, a, 2, xx);
    }
}

__global__ __launch_bounds__(kBlock) void k_finish_eval(Round r, Xch xg) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double dg = xsum<3>(xg, a, 0, sh);
    const double gg = xsum<3>(xg, a, 1, sh);
    const double xx = xsum<3>(xg, a, 2, sh);
    if (threadIdx.x == 0) {
        double* sc = r.scal[a];
        sc[S_DG] = dg;
        sc[S_GG] = gg;
        sc[S_XX] = xx;
    }
}

// gp . d of the freshly built direction -> scal[S_DGINIT] (the next line search's initial slope)
__global__ __launch_bounds__(kBlock) void k_store_dginit(MVec8 scal, Xch xd) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double di = xsum<1>(xd, a, 0, sh);
    if (threadIdx.x == 0) scal.p[a][S_DGINIT] = di;
}

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
__global__ __launch_bounds__(kBlock) void k_forces_t(ForcesRound r, const double* __restrict__ w0, int n) {
    const int a = blockIdx.y;
    const double* __restrict__ w = r.w[a];
    const double* __restrict__ b = r.a[a];
    double* __restrict__ t = r.t[a];
    const double theta = r.theta[a];
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const double wv = w[j], w0v = w0[j];
        double dd = 1.0;
        if (wv >= DBL_MIN && w0v >= DBL_MIN) dd += log(wv) - log(w0v);
        t[j] = (dd * theta + b[j]) * wv;
    }
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
constexpr int kStripCols = 16;

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
    constexpr int C = kStripCols;
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
    d2 pre[PIECES];
    auto fetch = [&](int strip) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int p = t + THREADS * i, row = p >> 3, part = p & 7;
            pre[i] = row < mp ? ldg2<NT>(Y + (size_t)row * ld + (size_t)strip * C + part * 2) : d2{0.0, 0.0};
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
    if (s < nstrips) fetch(s);
    for (; s < nstrips; s += gridDim.x) {
        __syncthreads();
        stash();
        __syncthreads();
        double w0v = 0.0;                                       // before the prefetch (vmcnt retires in order)
        const size_t col = (size_t)s * C + t / K;
        if (t < C * K) w0v = w0[col];
        if (s + (int)gridDim.x < nstrips) fetch(s + gridDim.x);
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
            fr.a[k][col] = valid ? x : 0.0;
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
        double* pa = fr.part[t];
        pa[(size_t)P_MAX * kMaxPartials + blockIdx.x] = m_run;   // thread t = (c 0, k t)
        pa[(size_t)P_SUM * kMaxPartials + blockIdx.x] = z;
        pa[(size_t)P_PP * kMaxPartials + blockIdx.x] = px;
    }
}

// Merge the blocks of k_forces_xy on THIS rank (one block per problem): m_r = max_b m_b,
// Z_r = sum_b e^{m_b - m_r} Z_b, likewise sum e x; P_MAX[b] <- e^{m_b - m_r}, the weight of block
// b's raw sums.  The rank totals {Z_r, sum e x, m_r} go to the tail of the rank's X_YBAR segment --
// the layout of the log-weights rounds -- so k_rows_combine<true> finishes both methods alike:
//   scal[S_LOGS] = M + log Z  (w_j = w0_j exp(x_j - S_LOGS)),  scal[S_P] = sum_j w_j x_j,
//   KL = sum_j w_j log(w_j / w0_j) = S_P - S_LOGS   (c_bioen_kernels_forces.c:246-258, with
//   log w_j - log w0_j = x_j - S_LOGS; the prior constant S_LOGS0 is zero for this method).
__global__ __launch_bounds__(kBlock) void k_forces_blockstats(ForcesRound fr, int nblk, int mp, int K, Xch xo) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    double* pa = fr.part[a];
    double* pm = pa + (size_t)P_MAX * kMaxPartials;
    const double mr = max_partials(pm, nblk, sh);
    double z = 0.0, px = 0.0;
    for (int b = threadIdx.x; b < nblk; b += kBlock) {
        const double fb = exp(pm[b] - mr);
        z = fma(fb, pa[(size_t)P_SUM * kMaxPartials + b], z);
        px = fma(fb, pa[(size_t)P_PP * kMaxPartials + b], px);
    }
    z = block_sum(z, sh);
    px = block_sum(px, sh);
    __syncthreads();
    for (int b = threadIdx.x; b < nblk; b += kBlock) pm[b] = exp(pm[b] - mr);
    if (threadIdx.x == 0) {
        double* tail = xo.base + (size_t)xo.rank * xo.payload + (size_t)mp * K + 3 * a;
        tail[0] = z;
        tail[1] = px;
        tail[2] = mr;
        fr.scal[a][S_LOGS0] = 0.0;
    }
}

// this rank's share of ybar: sum_b weight_b raw_i,b   (a wave per (row, problem), fixed order)
__global__ __launch_bounds__(kBlock) void k_forces_rows_weighted(const double* __restrict__ partial, int nblk, int mp,
                                                                 int K, ForcesRound fr, Xch xo) {
    const int a = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const double* __restrict__ wb = fr.part[a] + (size_t)P_MAX * kMaxPartials;
    double* out = xo.base + (size_t)xo.rank * xo.payload;
    for (int row = blockIdx.x * kWaves + wave; row < mp; row += gridDim.x * kWaves) {
        const double* p = partial + ((size_t)row * K + a) * nblk;
        double s = 0.0;
        for (int b = lane; b < nblk; b += 64) s = fma(wb[b], p[b], s);
        s = wave_sum(s);
        if (lane == 0) out[(size_t)row * K + a] = s;
    }
}

// w_j = w0_j exp(x_j - S_LOGS): the weights themselves, when a result is handed out
__global__ __launch_bounds__(kBlock) void k_forces_w_from_x(ForcesRound fr, const double* __restrict__ w0, int n) {
    const int a = blockIdx.y;
    const double* __restrict__ x = fr.a[a];
    double* __restrict__ w = fr.w[a];
    const double logz = fr.scal[a][S_LOGS];
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) w[j] = w0[j] * exp(x[j] - logz);
}

template <int K, bool NT, bool FROMX, int THREADS>
__global__ __launch_bounds__(THREADS, 2) void k_forces_bt(const double* __restrict__ Y, size_t ld, int mp, int nstrips,
                                                      const double* __restrict__ r_c,
                                                      const double* __restrict__ ybar_c, ForcesRound fr,
                                                      const double* __restrict__ w0,
                                                      double* __restrict__ partial, int nblk) {
    constexpr int C = kStripCols;
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
    d2 pre[PIECES];
    auto fetch = [&](int strip) {                               // global -> registers, 8 lanes per row segment
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int p = t + THREADS * i, row = p >> 3, part = p & 7;
            pre[i] = row < mp ? ldg2<NT>(Y + (size_t)row * ld + (size_t)strip * C + part * 2) : d2{0.0, 0.0};
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
    if (s < nstrips) fetch(s);
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
                lr = fr.a[t % K][col] - fr.scal[t % K][S_LOGS];
                wv = w0v * exp(lr);
            } else {
                wv = fr.w[t % K][col];
            }
        }
        if (s + (int)gridDim.x < nstrips) fetch(s + gridDim.x); // in flight during the three phases
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
            tv[c][k] = (dd * fr.theta[k] + b) * wv;
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

// ------------------------------------------------------------------------------
// L-BFGS vector kernels (liblbfgs lbfgs.c:543-615 with every scalar device-resident)
// ------------------------------------------------------------------------------
// s = x - xp, y = g - gp (lbfgs.c:549-551); partials of y.s and y.y (:559-561)
__global__ __launch_bounds__(kBlock) void k_update_sy(PairArgs p, int n, Xch xo) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double* __restrict__ x = p.x[a];
    const double* __restrict__ xp = p.xp[a];
    const double* __restrict__ g = p.g[a];
    const double* __restrict__ gp = p.gp[a];
    double* __restrict__ s = p.s[a];
    double* __restrict__ y = p.y[a];
    double ys = 0.0, yy = 0.0;
    const int n2 = (n + 1) >> 1;
    for (int q = blockIdx.x * kBlock + threadIdx.x; q < n2; q += gridDim.x * kBlock) {
        const int j = 2 * q;
        const d2 xa = *reinterpret_cast<const d2*>(x + j), xb = *reinterpret_cast<const d2*>(xp + j);
        const d2 ga = *reinterpret_cast<const d2*>(g + j), gb = *reinterpret_cast<const d2*>(gp + j);
        const d2 sv = {xa.x - xb.x, xa.y - xb.y};
        const d2 yv = {ga.x - gb.x, ga.y - gb.y};
        *reinterpret_cast<d2*>(s + j) = sv;
        *reinterpret_cast<d2*>(y + j) = yv;
        ys = fma(yv.x, sv.x, ys);
        yy = fma(yv.x, yv.x, yy);
        ys = fma(yv.y, sv.y, ys);
        yy = fma(yv.y, yv.y, yy);
    }
    ys = block_sum(ys, sh);
    yy = block_sum(yy, sh);
    if (threadIdx.x == 0) {
        xput<2>(xo, p.xpos[a], 0, ys);
        xput<2>(xo, p.xpos[a], 1, yy);
    }
}

// One fused step of the two-loop recursion (lbfgs.c:571-598).  The dot product a step needs
// was left as per-block partials by the previous step; every block re-reduces them (fixed
// order) in its prologue, so a step is ONE launch:
//   mode 0: d = -gp                                   [+ finalise y.s, y.y of slot `hist`]
//   mode 1: alpha_h = (S_h . d) / ys_h ; d -= alpha_h Y_h        (first loop)
//   mode 2: beta = (Y_h . d) / ys_h ; d += (alpha_h - beta) S_h  (second loop)
//   scale : d *= ys / yy   after the update (last step of the first loop)
// and in the same sweep  out_partials = vdot . d  for the next step (or gp . d, the next
// line search's initial slope).  mode -1: this problem has no step in this launch.
__global__ __launch_bounds__(kBlock) void k_recur(RecurArgs q, int n, Xch xin, Xch xsy, Xch xrec, Xch xdgi) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const int mode = q.mode[a];
    if (mode < 0) return;
    const int hist = q.hist[a];
    const int scale = q.scale[a];
    double* __restrict__ d = q.d[a];
    const double* __restrict__ gp = q.gp[a];
    const double* __restrict__ vaxpy = q.vaxpy[a];
    const double* __restrict__ vdot = q.vdot[a];
    double* scal = q.scal[a];
    double coef = 0.0, sc = 1.0;
    if (mode == 0) {
        if (q.finalize_sy[a] && blockIdx.x == 0) {
            const double ys = xsum<2>(xsy, a, 0, sh);
            const double yy = xsum<2>(xsy, a, 1, sh);
            if (threadIdx.x == 0) {
                scal[S_YSH + hist] = ys;
                scal[S_YS] = ys;
                scal[S_YY] = yy;
            }
        }
    } else {
        const double dot = xsum<1>(xin, a, 0, sh);
        const double ysh = scal[S_YSH + hist];
        if (mode == 1) {
            const double alpha = dot / ysh;
            coef = -alpha;
            if (blockIdx.x == 0 && threadIdx.x == 0) scal[S_ALPHA + hist] = alpha;
        } else {
            coef = scal[S_ALPHA + hist] - dot / ysh;
        }
        if (scale) sc = scal[S_YS] / scal[S_YY];
    }
    double acc = 0.0;
    const int n2 = (n + 1) >> 1;
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < n2; p += gridDim.x * kBlock) {
        const int j = 2 * p;
        d2 dv;
        if (mode == 0) {
            const d2 gv = *reinterpret_cast<const d2*>(gp + j);
            dv.x = -gv.x;
            dv.y = -gv.y;
        } else {
            const d2 av = *reinterpret_cast<const d2*>(vaxpy + j);
            const d2 old = *reinterpret_cast<const d2*>(d + j);
            dv.x = fma(coef, av.x, old.x);
            dv.y = fma(coef, av.y, old.y);
            if (scale) {
                dv.x *= sc;
                dv.y *= sc;
            }
        }
        *reinterpret_cast<d2*>(d + j) = dv;
        if (vdot) {
            const d2 vv = *reinterpret_cast<const d2*>(vdot + j);
            acc = fma(vv.x, dv.x, acc);
            acc = fma(vv.y, dv.y, acc);
        }
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) {
        if (q.to_dginit[a]) xput<1>(xdgi, a, 0, acc);
        else if (vdot) xput<1>(xrec, a, 0, acc);
    }
}

// ------------------------------------------------------------------------------
// direction from inner products ("compact" two-loop recursion)
// ------------------------------------------------------------------------------
// One sweep: s = xnew - xold, y = gnew - gold -> history slot `end`; and the 39 inner products
// of (s, y, gnew) with the basis B = {S_0..5, Y_0..5, gnew} (S_end = s, Y_end = y).
__global__ __launch_bounds__(kBlock) void k_gram(GramArgs q, int n, Xch xo) {
    __shared__ double sh[kWaves][64];
    const int a = blockIdx.y;
    const int e = q.end[a];
    const double* __restrict__ xn = q.xnew[a];
    const double* __restrict__ xo_ = q.xold[a];
    const double* __restrict__ gn = q.gnew[a];
    const double* __restrict__ go = q.gold[a];
    double acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.0;
    const int n2 = (n + 1) >> 1;
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < n2; p += gridDim.x * kBlock) {
        const int j = 2 * p;
        const d2 a0 = *reinterpret_cast<const d2*>(xn + j), a1 = *reinterpret_cast<const d2*>(xo_ + j);
        const d2 gv = *reinterpret_cast<const d2*>(gn + j), g1 = *reinterpret_cast<const d2*>(go + j);
        const d2 sv = {a0.x - a1.x, a0.y - a1.y};
        const d2 yv = {gv.x - g1.x, gv.y - g1.y};
        d2 B[kBasis];
#pragma unroll
        for (int k = 0; k < kHistory; ++k) {
            B[k] = (k == e) ? sv : *reinterpret_cast<const d2*>(q.S[a][k] + j);
            B[kHistory + k] = (k == e) ? yv : *reinterpret_cast<const d2*>(q.Y[a][k] + j);
        }
        B[2 * kHistory] = gv;
        *reinterpret_cast<d2*>(q.S[a][e] + j) = sv;
        *reinterpret_cast<d2*>(q.Y[a][e] + j) = yv;
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
        xo.base[(size_t)xo.rank * xo.payload + (size_t)(a * kGramDots + threadIdx.x) * xo.npl + blockIdx.x] = v;
    }
}

// Per problem (one block): finish the 39 sums, update the Gram matrix, run the two-loop recursion
// (lbfgs.c:571-598) on coefficients, leave them in gram[169..181] and gp.d in scal[S_DGINIT].
// Sharded contexts: one block per (sum, problem) totals THIS rank's block partials into the compact
// X_GRAMR stage, so that the all-gather ships 39 doubles per problem and rank instead of 39 x npl.
__global__ __launch_bounds__(kBlock) void k_gram_rank_reduce(Xch xi, Xch xo) {
    __shared__ double sh[kWaves];
    const int c = blockIdx.x, a = blockIdx.y;
    const double v = xsum_rank<kGramDots>(xi, xi.rank, a, c, sh);
    if (threadIdx.x == 0) xo.base[(size_t)xo.rank * xo.payload + (size_t)a * kGramDots + c] = v;
}

// One block per (sum, problem) finishes the 39 sums (gram[kGramSums + c]); a single block doing
// all of them walks 39 x npl partials as dependent L2 loads (66 us at N = 1e6, measured).
__global__ __launch_bounds__(kBlock) void k_gram_reduce(GramArgs q, Xch xi) {
    __shared__ double sh[kWaves];
    const int c = blockIdx.x, a = blockIdx.y;
    const double v = xsum<kGramDots>(xi, a, c, sh);
    if (threadIdx.x == 0) q.gram[a][kGramSums + c] = v;
}

// The 13x13 matrix lives in LDS while one thread walks the two loops.  FUSED: few partials per sum
// (small or sharded problems) -- the block also finishes the 39 sums, dealt to its 4 waves, which
// saves the k_gram_reduce launch.  The order of the additions differs between the two paths, so
// the choice depends on (npl, world) alone: every rank and every batch width takes the same one.
template <bool FUSED>
__global__ __launch_bounds__(kBlock) void k_gram_solve(GramArgs q, Xch xi) {
    __shared__ double dots[kGramDots];
    __shared__ double Gs[kBasis * kBasis];
    __shared__ double coef[kBasis];
    __shared__ double alpha[kHistory];
    const int a = blockIdx.y;
    double* G = q.gram[a];
    for (int i = threadIdx.x; i < kBasis * kBasis; i += kBlock) Gs[i] = G[i];
    if (FUSED) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int c = wave; c < kGramDots; c += kWaves) {
            double s = 0.0;
            for (int r = 0; r < xi.world; ++r) {
                const double* p = xi.base + (size_t)r * xi.payload + (size_t)(a * kGramDots + c) * xi.npl;
                for (int k = lane; k < xi.npl; k += 64) s += p[k];
            }
            s = wave_sum(s);
            if (lane == 0) dots[c] = s;
        }
    } else {
        for (int i = threadIdx.x; i < kGramDots; i += kBlock) dots[i] = G[kGramSums + i];
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int e = q.end[a], bound = q.bound[a];
    const int rs = e, ry = kHistory + e, rg = 2 * kHistory;
    for (int c = 0; c < kBasis; ++c) {
        Gs[rs * kBasis + c] = Gs[c * kBasis + rs] = dots[c];
        Gs[ry * kBasis + c] = Gs[c * kBasis + ry] = dots[kBasis + c];
    }
    for (int c = 0; c < kBasis; ++c) Gs[rg * kBasis + c] = Gs[c * kBasis + rg] = dots[2 * kBasis + c];
    for (int c = 0; c < kBasis; ++c) {        // the three rows/columns that changed go back to HBM
        G[rs * kBasis + c] = G[c * kBasis + rs] = Gs[rs * kBasis + c];
        G[ry * kBasis + c] = G[c * kBasis + ry] = Gs[ry * kBasis + c];
        G[rg * kBasis + c] = G[c * kBasis + rg] = Gs[rg * kBasis + c];
    }
    // q = -g as coefficients over {S, Y, g}
    for (int c = 0; c < kBasis; ++c) coef[c] = 0.0;
    coef[rg] = -1.0;
    for (int b = 0; b < bound; ++b) {         // first loop, newest -> oldest
        const int i = (e + kHistory - b) % kHistory;
        double sq = 0.0;
        for (int c = 0; c < kBasis; ++c) sq = fma(coef[c], Gs[i * kBasis + c], sq);
        const double al = sq / Gs[(kHistory + i) * kBasis + i];
        alpha[i] = al;
        coef[kHistory + i] -= al;
    }
    const double scale = Gs[ry * kBasis + rs] / Gs[ry * kBasis + ry];   // ys / yy of the newest pair
    for (int c = 0; c < kBasis; ++c) coef[c] *= scale;
    for (int b = bound - 1; b >= 0; --b) {    // second loop, oldest -> newest
        const int i = (e + kHistory - b) % kHistory;
        double yq = 0.0;
        for (int c = 0; c < kBasis; ++c) yq = fma(coef[c], Gs[(kHistory + i) * kBasis + c], yq);
        const double beta = yq / Gs[(kHistory + i) * kBasis + i];
        coef[i] += alpha[i] - beta;
    }
    double dg = 0.0;
    for (int c = 0; c < kBasis; ++c) dg = fma(coef[c], Gs[rg * kBasis + c], dg);
    for (int c = 0; c < kBasis; ++c) G[kBasis * kBasis + c] = coef[c];
    q.scal[a][S_DGINIT] = dg;
}

// d = sum_c coef_c B_c
__global__ __launch_bounds__(kBlock) void k_combine(GramArgs q, int n) {
    const int a = blockIdx.y;
    const double* coef = q.gram[a] + kBasis * kBasis;
    double cf[kBasis];
#pragma unroll
    for (int c = 0; c < kBasis; ++c) cf[c] = coef[c];
    double* __restrict__ d = q.d[a];
    const double* __restrict__ gn = q.gnew[a];
    const int n2 = (n + 1) >> 1;
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < n2; p += gridDim.x * kBlock) {
        const int j = 2 * p;
        const d2 gv = *reinterpret_cast<const d2*>(gn + j);
        d2 dv = {cf[2 * kHistory] * gv.x, cf[2 * kHistory] * gv.y};
#pragma unroll
        for (int k = 0; k < kHistory; ++k) {
            if (cf[k] != 0.0) {               // unused history slots may hold another problem's leftovers
                const d2 v = *reinterpret_cast<const d2*>(q.S[a][k] + j);
                dv.x = fma(cf[k], v.x, dv.x);
                dv.y = fma(cf[k], v.y, dv.y);
            }
            if (cf[kHistory + k] != 0.0) {
                const d2 v = *reinterpret_cast<const d2*>(q.Y[a][k] + j);
                dv.x = fma(cf[kHistory + k], v.x, dv.x);
                dv.y = fma(cf[kHistory + k], v.y, dv.y);
            }
        }
        *reinterpret_cast<d2*>(d + j) = dv;
    }
}

// ------------------------------------------------------------------------------
// level-1 algebra on resident N-vectors for the host-driven GSL-style minimizers
// (multimin.hpp).  Vectors may alias in the read-only positions, hence no __restrict__.
// All loops run over pairs up to ld/2: the padding is zero in every operand and stays zero.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_vaxpy(double a, const double* x, double* y, int n2) {
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < n2; p += gridDim.x * kBlock) {
        const d2 xv = *reinterpret_cast<const d2*>(x + 2 * p);
        d2 yv = *reinterpret_cast<d2*>(y + 2 * p);
        yv.x += a * xv.x;      // two roundings, as cblas_daxpy compiled without contraction
        yv.y += a * xv.y;
        *reinterpret_cast<d2*>(y + 2 * p) = yv;
    }
}

__global__ __launch_bounds__(kBlock) void k_vscal(double a, double* x, int n2) {
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < n2; p += gridDim.x * kBlock) {
        d2 v = *reinterpret_cast<d2*>(x + 2 * p);
        v.x *= a;
        v.y *= a;
        *reinterpret_cast<d2*>(x + 2 * p) = v;
    }
}

// dx = coef p ; x1 = x + dx      (directional_minimize.c: take_step)
__global__ __launch_bounds__(kBlock) void k_vstep(const double* x, const double* pv, double coef, double* x1,
                                                  double* dx, int n2) {
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < n2; p += gridDim.x * kBlock) {
        const d2 xv = *reinterpret_cast<const d2*>(x + 2 * p);
        const d2 dv = *reinterpret_cast<const d2*>(pv + 2 * p);
        const d2 s = {coef * dv.x, coef * dv.y};
        *reinterpret_cast<d2*>(dx + 2 * p) = s;
        const d2 o = {xv.x + s.x, xv.y + s.y};
        *reinterpret_cast<d2*>(x1 + 2 * p) = o;
    }
}

// up to 4 inner products in one pass; mode 1: [0] = #(x != y), [1] = max |x|
__global__ __launch_bounds__(kBlock) void k_vdots(VDotArgs q, int n2, double* part) {
    __shared__ double sh[kWaves];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < n2; p += gridDim.x * kBlock) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < q.k) {
                const d2 xv = *reinterpret_cast<const d2*>(q.x[k] + 2 * p);
                const d2 yv = *reinterpret_cast<const d2*>(q.y[k] + 2 * p);
                if (q.mode == 0) {
                    acc[k] = fma(xv.x, yv.x, acc[k]);
                    acc[k] = fma(xv.y, yv.y, acc[k]);
                } else if (k == 0) {
                    acc[0] += (xv.x != yv.x ? 1.0 : 0.0) + (xv.y != yv.y ? 1.0 : 0.0);
                } else {
                    acc[1] = fmax(acc[1], fmax(fabs(xv.x), fabs(xv.y)));
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < q.k) {
            const double v = (q.mode == 1 && k == 1) ? block_max(acc[k], sh) : block_sum(acc[k], sh);
            if (threadIdx.x == 0) part[(size_t)k * kMaxPartials + blockIdx.x] = v;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_vdots_finish(const double* part, int np, int k, int mode, double* out) {
    __shared__ double sh[kWaves];
    for (int q = 0; q < k; ++q) {
        const double* p = part + (size_t)q * kMaxPartials;
        double v;
        if (mode == 1 && q == 1) {
            double s = 0.0;
            for (int i = threadIdx.x; i < np; i += kBlock) s = fmax(s, p[i]);
            v = block_max(s, sh);
        } else {
            v = sum_partials(p, np, sh);
        }
        if (threadIdx.x == 0) out[q] = v;
    }
}

// ------------------------------------------------------------------------------
// assembly of yTilde = sim / sigma from raw simulated observables (observables.py:123-143 does
// this element by element in Python, then divides on the host)
// ------------------------------------------------------------------------------
// observables-major input already sits in Y: divide row i by sigma_i in place
__global__ __launch_bounds__(kBlock) void k_rows_div(double* __restrict__ Y, size_t ld, int m, int n,
                                                     const double* __restrict__ sigma) {
    const size_t total = (size_t)m * ld;
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (size_t)gridDim.x * kBlock) {
        const size_t i = idx / ld, j = idx - i * ld;
        if (j < (size_t)n) Y[idx] = Y[idx] / sigma[i];
    }
}

// structure-major chunk src[jc][i] (jc < ncols, i < m: one structure's observables contiguous) ->
// Y[i][col0 + jc] / sigma_i, through a 32 x 33 LDS tile so that both sides are coalesced
__global__ __launch_bounds__(kBlock) void k_transpose_div(const double* __restrict__ src, int ncols, int m,
                                                          double* __restrict__ Y, size_t ld, size_t col0,
                                                          const double* __restrict__ sigma) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    for (int r = ty; r < 32; r += 8) {
        const int jc = j0 + r, i = i0 + tx;
        tile[r][tx] = (jc < ncols && i < m) ? src[(size_t)jc * m + i] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, jc = j0 + tx;
        if (i < m && jc < ncols) Y[(size_t)i * ld + col0 + jc] = tile[tx][r] / sigma[i];
    }
}

// ------------------------------------------------------------------------------
// synthetic ensemble generated in HBM (bench): counter-based Box-Muller
// ------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(kBlock) void k_generate(double* __restrict__ Y, size_t ld, int m, int n, int mp,
                                                     unsigned long long col0, unsigned long long n_global,
                                                     const double* __restrict__ YTrue,
                                                     const double* __restrict__ sig_sim,
                                                     const double* __restrict__ sig_exp, unsigned long long seed) {
    const size_t half = ld / 2;
    const size_t total = (size_t)mp * half;
    for (size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (size_t)gridDim.x * kBlock) {
        const int i = (int)(idx / half);
        const size_t jp = idx - (size_t)i * half;
        const size_t j = jp * 2;
        d2 out = {0.0, 0.0};
        if (i < m && j < (size_t)n) {
            // counter = position of the column PAIR in the global (unsharded) matrix
            const unsigned long long ctr = (unsigned long long)i * ((n_global + 1) / 2) + (col0 / 2 + jp);
            const unsigned long long h1 = mix64(seed + 0x9E3779B97F4A7C15ULL * (2 * ctr + 1));
            const unsigned long long h2 = mix64(seed + 0x9E3779B97F4A7C15ULL * (2 * ctr + 2));
            const double u1 = ((double)(h1 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
            const double u2 = ((double)(h2 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
            const double rad = sqrt(-2.0 * log(u1));
            double sn, cs;
            sincos(6.283185307179586476925286766559 * u2, &sn, &cs);
            const double mu = YTrue[i], ss = sig_sim[i], inv = 1.0 / sig_exp[i];
            out.x = (mu + ss * rad * cs) * inv;
            if (j + 1 < (size_t)n) out.y = (mu + ss * rad * sn) * inv;
        }
        *reinterpret_cast<d2*>(Y + (size_t)i * ld + j) = out;
    }
}

// ==============================================================================
// host-side launchers
// ==============================================================================
int vec_grid(const bioen_hip_ctx* c) {
    // from ld (identical on every rank of a sharded context), 2 pairs (4 elements) per thread
    long long b = ((long long)c->ld + 4 * kBlock - 1) / (4 * kBlock);
    const long long cap = kMaxPartials / c->world;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

Xch make_xch(const bioen_hip_ctx* c, int stage, int payload) {
    Xch x;
    x.base = c->xbuf[stage];
    x.payload = payload;
    x.world = c->world;
    x.rank = c->rank;
    x.npl = vec_grid(c);
    return x;
}

int rows_grid(const bioen_hip_ctx* c) {
    int b = c->mp / kWaves;
    if (b > kMaxPartials) b = kMaxPartials;
    return b;
}

struct TimedLaunch {
    bioen_hip_ctx* c;
    KernelTimer::Pair pr;
    bool on;
    TimedLaunch(bioen_hip_ctx* ctx, int which, int k) : c(ctx), on(ctx->timer.enabled) {
        if (!on) return;
        KernelTimer& t = c->timer;
        if (!t.pool.empty()) {
            pr = t.pool.back();
            t.pool.pop_back();
        } else {
            (void)hipEventCreate(&pr.a);
            (void)hipEventCreate(&pr.b);
        }
        pr.which = which;
        pr.k = k;
        (void)hipEventRecord(pr.a, c->stream);
    }
    ~TimedLaunch() {
        if (!on) return;
        (void)hipEventRecord(pr.b, c->stream);
        c->timer.pending.push_back(pr);
    }
};

// ---- forward ---------------------------------------------------------------------------
template <int K, int STEPS, bool NT, bool CENTER>
static void fwd_launch(bioen_hip_ctx* c, const Vec8& v) {
    const int total_steps = (int)(c->ld / 128);
    dim3 grid(c->mp / kRowAlign, c->fwd_ctiles);
    hipLaunchKernelGGL((k_fwd_partial<8, K, STEPS, NT, CENTER>), grid, dim3(kBlock), 0, c->stream, c->Y, c->ld, v,
                       c->ybar_c, c->fwd_partial, c->fwd_ctiles, c->fwd_steps, total_steps);
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

void launch_fwd_rows_local(bioen_hip_ctx* c, int K, bool logw) {
    const Xch xo = make_xch(c, X_YBAR, ybar_payload(c, K, logw));
    const Xch xe = make_xch(c, X_EXP, 3 * K * vec_grid(c));
    if (logw)
        hipLaunchKernelGGL(k_fwd_rows_local<true>, dim3(rows_grid(c), K), dim3(kBlock), 0, c->stream, c->fwd_partial,
                           c->fwd_ctiles, c->mp, K, xo, xe);
    else
        hipLaunchKernelGGL(k_fwd_rows_local<false>, dim3(rows_grid(c), K), dim3(kBlock), 0, c->stream,
                           c->fwd_partial, c->fwd_ctiles, c->mp, K, xo, xe);
}

// w = e * scal[S_INV]: the weights themselves are only needed when a result is handed out
__global__ __launch_bounds__(kBlock) void k_scale_w(Round r, int n2) {
    const int a = blockIdx.y;
    const double inv = r.scal[a][S_INV];
    double* __restrict__ w = r.w[a];
    for (int p = blockIdx.x * kBlock + threadIdx.x; p < n2; p += gridDim.x * kBlock) {
        d2 v = *reinterpret_cast<d2*>(w + 2 * p);
        v.x *= inv;
        v.y *= inv;
        *reinterpret_cast<d2*>(w + 2 * p) = v;
    }
}

void launch_scale_w(bioen_hip_ctx* c, const Round& r) {
    hipLaunchKernelGGL(k_scale_w, dim3(vec_grid(c), r.n), dim3(kBlock), 0, c->stream, r, (int)(c->ld / 2));
}

int combine_grid(const bioen_hip_ctx*) { return 1; }

void launch_rows_combine(bioen_hip_ctx* c, const Round& r, bool logw) {
    MVec8 part;
    for (int a = 0; a < kMaxBatch; ++a) part.p[a] = a < r.n ? r.part[a] : nullptr;
    if (logw)
        hipLaunchKernelGGL(k_rows_combine<true>, dim3(1, r.n), dim3(kBlock), 0, c->stream,
                           make_xch(c, X_YBAR, ybar_payload(c, r.n, true)), c->mp, r.n, c->YT, c->row_offset, c->row_scale,
                           c->ybar_c, c->r_c, part, r);
    else
        hipLaunchKernelGGL(k_rows_combine<false>, dim3(1, r.n), dim3(kBlock), 0, c->stream,
                           make_xch(c, X_YBAR, c->mp * r.n), c->mp, r.n, c->YT, c->row_offset, c->row_scale,
                           c->ybar_c, c->r_c, part, r);
}

void launch_fwd_rows_forces_grad(bioen_hip_ctx* c, int K, int ctiles) {
    hipLaunchKernelGGL(k_fwd_rows_forces_grad, dim3(rows_grid(c), K), dim3(kBlock), 0, c->stream, c->fwd_partial,
                       ctiles, c->mp, K, c->gm);
}

// sharded: this rank's share of the forces gradient -> its X_YBAR segment; after the exchange
// k_sum_ranks adds the shares in rank order (identical on every rank) -> gm
void launch_fwd_rows_forces_grad_share(bioen_hip_ctx* c, int K, int ctiles) {
    const Xch xo = make_xch(c, X_YBAR, c->mp * K);
    hipLaunchKernelGGL(k_fwd_rows_forces_grad, dim3(rows_grid(c), K), dim3(kBlock), 0, c->stream, c->fwd_partial,
                       ctiles, c->mp, K, xo.base + (size_t)xo.rank * xo.payload);
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

// ---- forces evaluation over LDS-resident column strips (M <= 1024) --------------------------
static int strip_threads(const bioen_hip_ctx* c) { return c->mp <= 512 ? 256 : 512; }

int forces_fused_blocks(const bioen_hip_ctx* c) {      // 0: not applicable on this context
    if (c->mp > 1024) return 0;
    const int nstrips = (int)(c->ld / kStripCols);
    // 256 threads: two 70-KB blocks per CU; 512 threads: one 148-KB block per CU
    return std::min(strip_threads(c) == 256 ? kFusedBlocks : kFusedBlocks / 2, nstrips);
}

template <int K, bool NT, int THREADS>
static void forces_strip_launch(bioen_hip_ctx* c, const ForcesRound& fr, int nblk, int pass) {
    const int nstrips = (int)(c->ld / kStripCols);
    if (pass == 1)
        hipLaunchKernelGGL((k_forces_xy<K, NT, THREADS>), dim3(nblk), dim3(THREADS), 0, c->stream, c->Y, c->ld, c->mp,
                           nstrips, c->n, c->um, fr, c->fixed, c->fwd_partial, nblk);
    else
        hipLaunchKernelGGL((k_forces_bt<K, NT, true, THREADS>), dim3(nblk), dim3(THREADS), 0, c->stream, c->Y, c->ld,
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
    if (strip_threads(c) == 256) forces_strip_dispatch_k<NT, 256>(c, fr, nblk, pass);
    else forces_strip_dispatch_k<NT, 512>(c, fr, nblk, pass);
}

// pass 1: x = yTilde^T f, online softmax, raw ybar per block; then the block merge and ybar -> X_YBAR
void launch_forces_xy(bioen_hip_ctx* c, const ForcesRound& fr, int nblk) {
    {
        TimedLaunch tl(c, 1, fr.n);
        if (c->nontemporal) forces_strip_dispatch<true>(c, fr, nblk, 1); else forces_strip_dispatch<false>(c, fr, nblk, 1);
    }
    const Xch xo = make_xch(c, X_YBAR, ybar_payload(c, fr.n, true));
    hipLaunchKernelGGL(k_forces_blockstats, dim3(1, fr.n), dim3(kBlock), 0, c->stream, fr, nblk, c->mp, fr.n, xo);
    hipLaunchKernelGGL(k_forces_rows_weighted, dim3(rows_grid(c), fr.n), dim3(kBlock), 0, c->stream, c->fwd_partial,
                       nblk, c->mp, fr.n, fr, xo);
}

// pass 2: b = yTilde^T r, t, centred yTilde . t
void launch_forces_bt(bioen_hip_ctx* c, const ForcesRound& fr, int nblk) {
    TimedLaunch tl(c, 0, fr.n);
    if (c->nontemporal) forces_strip_dispatch<true>(c, fr, nblk, 2); else forces_strip_dispatch<false>(c, fr, nblk, 2);
}

void launch_forces_w_from_x(bioen_hip_ctx* c, const ForcesRound& fr) {
    hipLaunchKernelGGL(k_forces_w_from_x, dim3(vec_grid(c), fr.n), dim3(kBlock), 0, c->stream, fr, c->fixed, c->n);
}



// ---- adjoint ---------------------------------------------------------------------------
template <int K, bool NT, bool CENTER>
static void adj_launch(bioen_hip_ctx* c, const double* u_c, const MVec8& out) {
    dim3 grid((unsigned)(c->ld / 128));
    hipLaunchKernelGGL((k_adj<8, K, NT, CENTER>), grid, dim3(kBlock), 0, c->stream, c->Y, c->ld, c->mp / kWaves,
                       u_c, c->ybar_c, out);
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

// ---- log-weights vector kernels -----------------------------------------------------------
void launch_trial(bioen_hip_ctx* c, const Round& r) {
    hipLaunchKernelGGL(k_trial, dim3(vec_grid(c), r.n), dim3(kBlock), 0, c->stream, r, c->n,
                       make_xch(c, X_MAX, r.n * vec_grid(c)));
}

void launch_max(bioen_hip_ctx* c, const Round& r) {
    hipLaunchKernelGGL(k_max, dim3(vec_grid(c), r.n), dim3(kBlock), 0, c->stream, r, c->n,
                       make_xch(c, X_MAX, r.n * vec_grid(c)));
}

void launch_logw_exp(bioen_hip_ctx* c, const Round& r) {
    hipLaunchKernelGGL(k_logw_exp, dim3(vec_grid(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                       make_xch(c, X_MAX, r.n * vec_grid(c)), make_xch(c, X_EXP, 3 * r.n * vec_grid(c)));
}

void launch_logw_norm(bioen_hip_ctx* c, const Round& r) {
    hipLaunchKernelGGL(k_logw_norm, dim3(vec_grid(c), r.n), dim3(kBlock), 0, c->stream, r, c->n,
                       make_xch(c, X_EXP, 3 * r.n * vec_grid(c)));
}

void launch_logw_logs0(bioen_hip_ctx* c, const Round& r) {
    hipLaunchKernelGGL(k_logsumexp1, dim3(1), dim3(kBlock), 0, c->stream, c->fixed, c->n, r);
}

void launch_logw_grad(bioen_hip_ctx* c, const Round& r) {
    hipLaunchKernelGGL(k_logw_grad, dim3(vec_grid(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                       make_xch(c, X_GRAD, 3 * r.n * vec_grid(c)));
}

void launch_finish_eval(bioen_hip_ctx* c, const Round& r) {
    hipLaunchKernelGGL(k_finish_eval, dim3(1, r.n), dim3(kBlock), 0, c->stream, r,
                       make_xch(c, X_GRAD, 3 * r.n * vec_grid(c)));
}

void launch_store_dginit(bioen_hip_ctx* c, int k, const MVec8& scal) {
    hipLaunchKernelGGL(k_store_dginit, dim3(1, k), dim3(kBlock), 0, c->stream, scal,
                       make_xch(c, X_DGI, k * vec_grid(c)));
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

void launch_forces_t(bioen_hip_ctx* c, const ForcesRound& r) {
    hipLaunchKernelGGL(k_forces_t, dim3(vec_grid(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n);
}

void launch_forces_scalars(bioen_hip_ctx* c, const ForcesRound& r) {
    hipLaunchKernelGGL(k_forces_scalars, dim3(1, r.n), dim3(kBlock), 0, c->stream, r, combine_grid(c), vec_grid(c));
}

// ---- L-BFGS vector kernels ----------------------------------------------------------------------
void launch_update_sy(bioen_hip_ctx* c, const PairArgs& a, int kdir) {
    hipLaunchKernelGGL(k_update_sy, dim3(vec_grid(c), a.n), dim3(kBlock), 0, c->stream, a, c->n,
                       make_xch(c, X_SY, 2 * kdir * vec_grid(c)));
}

void launch_recur(bioen_hip_ctx* c, const RecurArgs& a, int step) {
    const int k = a.n, g = vec_grid(c);
    // step s reads the running dot its predecessor left in REC[(s-1)&1] and writes REC[s&1]
    hipLaunchKernelGGL(k_recur, dim3(g, k), dim3(kBlock), 0, c->stream, a, c->n,
                       make_xch(c, ((step - 1) & 1) ? X_REC1 : X_REC0, k * g), make_xch(c, X_SY, 2 * k * g),
                       make_xch(c, (step & 1) ? X_REC1 : X_REC0, k * g), make_xch(c, X_DGI, k * g));
}

void launch_gram(bioen_hip_ctx* c, const GramArgs& a) {
    hipLaunchKernelGGL(k_gram, dim3(vec_grid(c), a.n), dim3(kBlock), 0, c->stream, a, c->n,
                       make_xch(c, X_GRAM, kGramDots * a.n * vec_grid(c)));
}

static Xch gram_rank_view(const bioen_hip_ctx* c, int k) {      // X_GRAMR: one value per (rank, problem, sum)
    Xch x = make_xch(c, X_GRAMR, kGramDots * k);
    x.npl = 1;
    return x;
}

void launch_gram_rank_reduce(bioen_hip_ctx* c, int k) {
    hipLaunchKernelGGL(k_gram_rank_reduce, dim3(kGramDots, k), dim3(kBlock), 0, c->stream,
                       make_xch(c, X_GRAM, kGramDots * k * vec_grid(c)), gram_rank_view(c, k));
}

void launch_gram_solve(bioen_hip_ctx* c, const GramArgs& a) {
    if (c->world > 1) {                                  // the ranks' totals (after the X_GRAMR exchange)
        hipLaunchKernelGGL(k_gram_solve<true>, dim3(1, a.n), dim3(kBlock), 0, c->stream, a, gram_rank_view(c, a.n));
        return;
    }
    const Xch xi = make_xch(c, X_GRAM, kGramDots * a.n * vec_grid(c));
    if ((long long)vec_grid(c) * c->world <= 256) {      // <= 4 dependent loads per lane and sum
        hipLaunchKernelGGL(k_gram_solve<true>, dim3(1, a.n), dim3(kBlock), 0, c->stream, a, xi);
    } else {
        hipLaunchKernelGGL(k_gram_reduce, dim3(kGramDots, a.n), dim3(kBlock), 0, c->stream, a, xi);
        hipLaunchKernelGGL(k_gram_solve<false>, dim3(1, a.n), dim3(kBlock), 0, c->stream, a, xi);
    }
}

void launch_combine(bioen_hip_ctx* c, const GramArgs& a) {
    hipLaunchKernelGGL(k_combine, dim3(vec_grid(c), a.n), dim3(kBlock), 0, c->stream, a, c->n);
}

// ---- level-1 algebra (multimin) -------------------------------------------------------------
void launch_vaxpy(bioen_hip_ctx* c, double a, const double* x, double* y) {
    hipLaunchKernelGGL(k_vaxpy, dim3(vec_grid(c)), dim3(kBlock), 0, c->stream, a, x, y, (int)(c->ld / 2));
}
void launch_vscal(bioen_hip_ctx* c, double a, double* x) {
    hipLaunchKernelGGL(k_vscal, dim3(vec_grid(c)), dim3(kBlock), 0, c->stream, a, x, (int)(c->ld / 2));
}
void launch_vstep(bioen_hip_ctx* c, const double* x, const double* p, double coef, double* x1, double* dx) {
    hipLaunchKernelGGL(k_vstep, dim3(vec_grid(c)), dim3(kBlock), 0, c->stream, x, p, coef, x1, dx, (int)(c->ld / 2));
}
void launch_vdots(bioen_hip_ctx* c, const VDotArgs& q, double* part, double* out) {
    const int g = vec_grid(c);
    hipLaunchKernelGGL(k_vdots, dim3(g), dim3(kBlock), 0, c->stream, q, (int)(c->ld / 2), part);
    hipLaunchKernelGGL(k_vdots_finish, dim3(1), dim3(kBlock), 0, c->stream, part, g, q.k, q.mode, out);
}

void launch_rows_div(bioen_hip_ctx* c, const double* sigma) {
    hipLaunchKernelGGL(k_rows_div, dim3(256 * 16), dim3(kBlock), 0, c->stream, c->Y, c->ld, c->m, c->n, sigma);
}

void launch_transpose_div(bioen_hip_ctx* c, const double* src, int ncols, size_t col0, const double* sigma) {
    dim3 grid((c->m + 31) / 32, (ncols + 31) / 32);
    hipLaunchKernelGGL(k_transpose_div, grid, dim3(kBlock), 0, c->stream, src, ncols, c->m, c->Y, c->ld, col0, sigma);
}

void launch_generate(bioen_hip_ctx* c, const double* YTrue, const double* sig_sim, const double* sig_exp,
                     unsigned long long seed) {
    hipLaunchKernelGGL(k_generate, dim3(256 * 16), dim3(kBlock), 0, c->stream, c->Y, c->ld, c->m, c->n, c->mp,
                       (unsigned long long)c->col0, (unsigned long long)c->n_global, YTrue, sig_sim, sig_exp, seed);
}

}  // namespace bioen
