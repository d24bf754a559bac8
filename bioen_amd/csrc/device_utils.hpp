// Device-side building blocks shared by the kernel files: wave / block reductions with fixed
// shapes, accessors of the exchange stages (ctx.hpp: XStage), 16-byte loads, launch timing.
#pragma once

#include "kernels.hpp"

#include <hip/hip_ext.h>

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

// ---- consumers of TRANSPOSED partials (the strip kernels): P[set * n + idx], idx = row K + a ------------------------
struct TermAdd {
    __device__ __forceinline__ double operator()(int, double v, double s) const { return s + v; }
};

// ---- r05: consumers of the strip kernels' sets, per SEGMENT.  P -> entry `idx` of the segment's first set; set b at
// P[b n].  THE share of an entry in a segment: its sets' terms are dealt to EIGHT running sums (part p: sets p, p + 8, ...
// in turn, from +0.0; fold > 1: "set" b is a GROUP of `fold` consecutive sets -- the chunks the forward kernel has not
// folded itself -- whose value is their sum in turn from +0.0), which meet as ((s0 + s4) + (s2 + s6)) + ((s1 + s5) +
// (s3 + s7)).  A block = 8 parts x 32 entries: every load of a wave is 32 consecutive entries of one set (256 B).  The
// classic wave-per-entry tree (tiles_sum16, r02-r04) had half its lanes idle on a segment's 32 groups and cost 8 x its
// time over the eight segments.  Returns the share in threads < 32 (entry blockIdx.x * 32 + threadIdx.x).
template <class Term>
__device__ __forceinline__ double sets_sum8(const double* __restrict__ P, size_t n, int nsets, int fold,
                                            double (*lds)[32] /* [8][32] */, Term term) {
    const int p = threadIdx.x >> 5, el = threadIdx.x & 31;
    double s = 0.0;
    if (fold <= 1) {
        int b = p;
        for (; b + 24 < nsets; b += 32) {                      // four of the part's sets at a time: their loads in flight together
            const double v0 = P[(size_t)b * n], v1 = P[(size_t)(b + 8) * n], v2 = P[(size_t)(b + 16) * n], v3 = P[(size_t)(b + 24) * n];
            s = term(b, v0, s);
            s = term(b + 8, v1, s);
            s = term(b + 16, v2, s);
            s = term(b + 24, v3, s);
        }
        for (; b < nsets; b += 8) s = term(b, P[(size_t)b * n], s);
    } else if (fold <= 8) {
        // (StripSets: at most eight chunks per group) all chunk loads of TWO of the part's groups in flight together -- one
        // after the other (the trip count is a kernel argument) a rank of eight spent 10 us per round here; the adds in turn
        int b = p;
        for (; b + 8 < nsets; b += 16) {
            double u[8], v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                u[i] = i < fold ? P[((size_t)b * fold + i) * n] : 0.0;
                v[i] = i < fold ? P[((size_t)(b + 8) * fold + i) * n] : 0.0;
            }
            double gu = 0.0, gv = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < fold) {
                    gu += u[i];
                    gv += v[i];
                }
            s = term(b, gu, s);
            s = term(b + 8, gv, s);
        }
        for (; b < nsets; b += 8) {
            double u[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) u[i] = i < fold ? P[((size_t)b * fold + i) * n] : 0.0;
            double gu = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < fold) gu += u[i];
            s = term(b, gu, s);
        }
    } else {
        for (int b = p; b < nsets; b += 8) {
            double gv = 0.0;
            for (int i = 0; i < fold; ++i) gv += P[((size_t)b * fold + i) * n];
            s = term(b, gv, s);
        }
    }
    __syncthreads();      // protect lds against its previous use
    lds[p][el] = s;
    __syncthreads();
    return ((lds[0][el] + lds[4][el]) + (lds[2][el] + lds[6][el])) + ((lds[1][el] + lds[5][el]) + (lds[3][el] + lds[7][el]));
}

// ---- canonical segments (ctx.hpp: XStage) ---------------------------------------------------------------------
// An N-vector kernel is launched with npl blocks per local segment (grid.x = npl * vr): block blockIdx.x works on
// segment v = blockIdx.x / npl as its block b = blockIdx.x % npl -- exactly the columns, in exactly the order, that
// block b of a GPU holding nothing but that segment would work on.  Local columns of the segment: [j0, jend).
constexpr int kShRed = 128;            // LDS doubles of the stage-sum helpers below (>= nseg)

struct SegPos {
    int v, b;                          // local segment, block within it
    int j0, jend;                      // first column of the segment, end of its VALID columns (jend <= j0: empty)
};
__device__ __forceinline__ SegPos seg_pos(int npl, int segcols, int n) {
    SegPos s;
    s.v = blockIdx.x / npl;
    s.b = blockIdx.x - s.v * npl;
    s.j0 = s.v * segcols;
    const int e = s.j0 + segcols;
    s.jend = n < e ? n : e;
    return s;
}
// the block's walk over its segment, in 16-byte pairs: for (j = seg_first(...); j < sp.jend; j += seg_step(npl))
__device__ __forceinline__ int seg_first(const SegPos& s) { return s.j0 + 2 * (s.b * kBlock + (int)threadIdx.x); }
__device__ __forceinline__ int seg_step(int npl) { return 2 * npl * kBlock; }

// ---- exchange-stage accessors.  Layout [segment][problem a][array q][block].
// put: this block's partial (its segment, its block number there).
template <int A>
__device__ __forceinline__ const double* xseg_ptr(const Xch& x, int seg, int a, int q) {
    return x.base + (size_t)seg * x.payload + (size_t)(a * A + q) * x.npl;
}
template <int A>
__device__ __forceinline__ void xput(const Xch& x, int a, int q, double v) {
    const int vloc = blockIdx.x / x.npl;
    x.base[(size_t)(x.rank + vloc) * x.payload + (size_t)(a * A + q) * x.npl + (blockIdx.x - vloc * x.npl)] = v;
}

// THE sum of one segment's block partials (every consumer forms it this way): by one wave, lane l adding the partials
// l, l + 64, ... in turn, the 64 lane sums meeting in wave_sum's pair order.  Result in every lane.
__device__ __forceinline__ double wave_seg_total(const double* __restrict__ p, int npl) {
    double s = 0.0;
    for (int k = threadIdx.x & 63; k < npl; k += 64) s += p[k];
    return wave_sum(s);
}
__device__ __forceinline__ double wave_seg_max(const double* __restrict__ p, int npl) {
    double s = -DBL_MAX;
    for (int k = threadIdx.x & 63; k < npl; k += 64) s = fmax(s, p[k]);
    return wave_max(s);
}

// THE sum over structures: the segments' totals added in segment order, from +0.0 -- whichever GPU computed which.
// (A block of four waves: wave w totals the segments w, w + 4, ...; every thread then adds them up.)
template <int A>
__device__ __forceinline__ double xsum(const Xch& x, int a, int q, double* sh /* [kShRed] */) {
    const int wave = threadIdx.x >> 6;
    __syncthreads();   // protect sh against its previous use
    for (int seg = wave; seg < x.world; seg += kWaves) {
        const double t = wave_seg_total(xseg_ptr<A>(x, seg, a, q), x.npl);
        if ((threadIdx.x & 63) == 0) sh[seg] = t;
    }
    __syncthreads();
    double s = 0.0;
    for (int seg = 0; seg < x.world; ++seg) s += sh[seg];
    return s;
}
// NQ arrays of one problem at once (one barrier pair): out[q] = xsum<A>(x, a, q0 + q)
template <int A, int NQ>
__device__ __forceinline__ void xsum_multi(const Xch& x, int a, int q0, double* sh /* [NQ * kShRed] */, double (&out)[NQ]) {
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    for (int task = wave; task < NQ * x.world; task += kWaves) {
        const int q = task / x.world, seg = task - q * x.world;
        const double t = wave_seg_total(xseg_ptr<A>(x, seg, a, q0 + q), x.npl);
        if ((threadIdx.x & 63) == 0) sh[q * kShRed + seg] = t;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double s = 0.0;
        for (int seg = 0; seg < x.world; ++seg) s += sh[q * kShRed + seg];
        out[q] = s;
    }
}

// total of ONE segment (every wave of the block forms it for itself: no LDS, no barrier)
template <int A>
__device__ __forceinline__ double xsum_seg(const Xch& x, int seg, int a, int q) {
    return wave_seg_total(xseg_ptr<A>(x, seg, a, q), x.npl);
}

// maximum over the segment THIS block works on (its own block maxima need no exchange before it)
template <int A>
__device__ __forceinline__ double xmax_local(const Xch& x, int a, int q) {
    return wave_seg_max(xseg_ptr<A>(x, x.rank + (int)(blockIdx.x / x.npl), a, q), x.npl);
}

// finished per-segment totals (stages with npl = 1: X_GRAMR, the tail of X_YBAR): added in segment order from +0.0
__device__ __forceinline__ double seg_order_sum(const double* __restrict__ first, size_t stride, int nseg) {
    double s = 0.0;
    if (nseg <= kMaxSeg) {      // the loads together, the adds in order (a trip count from a kernel argument sends them one by one)
        double v[kMaxSeg];
#pragma unroll
        for (int seg = 0; seg < kMaxSeg; ++seg) v[seg] = first[(size_t)(seg < nseg ? seg : 0) * stride];
#pragma unroll
        for (int seg = 0; seg < kMaxSeg; ++seg)
            if (seg < nseg) s += v[seg];
        return s;
    }
    for (int seg = 0; seg < nseg; ++seg) s += first[(size_t)seg * stride];
    return s;
}

// The scalar part of the Gram-form direction (kernels_logw.hip: k_gram_solve; kernels_devls.hip: k_dev_decide), run by
// ONE thread: the 39 finished sums `dots` update rows / columns s_e, y_e, g of the 13 x 13 Gram matrix (G in HBM, Gs
// its LDS image), then the two-loop recursion (lbfgs.c:571-598) runs on 13 coefficients over {S_0..5, Y_0..5, g};
// the coefficients go to G[169..181], gp . d to scal[S_DGINIT].
__device__ __forceinline__ void gram_solve_thread0(double* G, double* Gs, const double* dots, double* alpha, int e,
                                                   int bound, double* scal) {
    const int rs = e, ry = kHistory + e, rg = 2 * kHistory;
    for (int c = 0; c < kBasis; ++c) {
        Gs[rs * kBasis + c] = Gs[c * kBasis + rs] = dots[c];
        Gs[ry * kBasis + c] = Gs[c * kBasis + ry] = dots[kBasis + c];
    }
    for (int c = 0; c < kBasis; ++c) Gs[rg * kBasis + c] = Gs[c * kBasis + rg] = dots[2 * kBasis + c];
    if (G != Gs)                              // (the decision kernel updates the LDS image in place and writes it back whole:
        for (int c = 0; c < kBasis; ++c) {    //  as a self-copy these 39 dependent load -> store pairs cost one lane 4 k cycles)
            G[rs * kBasis + c] = G[c * kBasis + rs] = Gs[rs * kBasis + c];     // the three rows/columns that changed go back to HBM
            G[ry * kBasis + c] = G[c * kBasis + ry] = Gs[ry * kBasis + c];
            G[rg * kBasis + c] = G[c * kBasis + rg] = Gs[rg * kBasis + c];
        }
    // q = -g as coefficients over {S, Y, g}.  The coefficients stay in registers; slot numbers are run-time values, so
    // the one entry a step changes is picked by comparison.  r03: the rows a loop needs are fetched from LDS up front,
    // all at once (the lane used to wait for an LDS round trip, an integer modulo and thirteen selects in every one of
    // the up to twelve dependent steps: 12.5 k cycles of the decision kernel's 33 k), the per-step quotients stay in
    // registers.  Same operations on the same numbers in the same order as before.
    double cf[kBasis];
#pragma unroll
    for (int c = 0; c < kBasis; ++c) cf[c] = c == rg ? -1.0 : 0.0;
    int idx[kHistory];                       // ring order, newest -> oldest: (e + kHistory - b) % kHistory
#pragma unroll
    for (int b = 0; b < kHistory; ++b) {
        idx[b] = e - b;
        if (idx[b] < 0) idx[b] += kHistory;
    }
    double R[kHistory][kBasis], D[kHistory], A[kHistory];
#pragma unroll
    for (int b = 0; b < kHistory; ++b) {
        D[b] = 1.0;
        A[b] = 0.0;
#pragma unroll
        for (int c = 0; c < kBasis; ++c) R[b][c] = 0.0;
        if (b < bound) {
#pragma unroll
            for (int c = 0; c < kBasis; ++c) R[b][c] = Gs[idx[b] * kBasis + c];
            D[b] = Gs[(kHistory + idx[b]) * kBasis + idx[b]];
        }
    }
#pragma unroll
    for (int b = 0; b < kHistory; ++b) {      // first loop, newest -> oldest
        if (b < bound) {
            double sq = 0.0;
#pragma unroll
            for (int c = 0; c < kBasis; ++c) sq = fma(cf[c], R[b][c], sq);
            const double al = sq / D[b];
            A[b] = al;
            alpha[idx[b]] = al;
#pragma unroll
            for (int c = kHistory; c < 2 * kHistory; ++c)
                if (c == kHistory + idx[b]) cf[c] -= al;
        }
    }
    const double scale = Gs[ry * kBasis + rs] / Gs[ry * kBasis + ry];   // ys / yy of the newest pair
#pragma unroll
    for (int c = 0; c < kBasis; ++c) cf[c] *= scale;
#pragma unroll
    for (int b = 0; b < kHistory; ++b)
        if (b < bound) {
#pragma unroll
            for (int c = 0; c < kBasis; ++c) R[b][c] = Gs[(kHistory + idx[b]) * kBasis + c];
        }
#pragma unroll
    for (int b = kHistory - 1; b >= 0; --b) {    // second loop, oldest -> newest
        if (b < bound) {
            double yq = 0.0;
#pragma unroll
            for (int c = 0; c < kBasis; ++c) yq = fma(cf[c], R[b][c], yq);
            const double beta = yq / D[b];
#pragma unroll
            for (int c = 0; c < kHistory; ++c)
                if (c == idx[b]) cf[c] += A[b] - beta;
        }
    }
    double dg = 0.0;
#pragma unroll
    for (int c = 0; c < kBasis; ++c) dg = fma(cf[c], Gs[rg * kBasis + c], dg);
#pragma unroll
    for (int c = 0; c < kBasis; ++c) G[kBasis * kBasis + c] = cf[c];
    scal[S_DGINIT] = dg;
}

template <bool NT>
__device__ __forceinline__ d2 ldg2(const double* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const d2*>(p));
    return *reinterpret_cast<const d2*>(p);
}

// Cache policy of the N-vector streams of the round (r04).  POLICY = true (contexts whose N-vectors are far larger than
// the caches, ctx->nvec_nt): the L-BFGS history vectors -- read once per kernel, 96 MB per problem at N = 1e6 -- come
// through nontemporal loads and every output leaves through nontemporal stores, so that neither evicts the operands the
// NEXT kernel of the round reads again (x, e, a, d, g: left on plain loads).  Values are untouched, so are the bits.
// Measured at the headline, same box, processes alternating (profiles/r04_nvec_nt_ab.txt): 2755-2758 -> 2726-2728 us per
// round (-1.1 %); all loads nontemporal as well: 2738-2744; history loads alone: -0.9 %; x and d (whose reader is the very
// next kernel) kept on plain stores: no gain, e as well: +1 %.
template <bool POLICY>
__device__ __forceinline__ d2 ld_hist(const double* p) {
    if (POLICY) return __builtin_nontemporal_load(reinterpret_cast<const d2*>(p));
    return *reinterpret_cast<const d2*>(p);
}
__device__ __forceinline__ d2 ld_vec(const double* p) { return *reinterpret_cast<const d2*>(p); }
template <bool POLICY>
__device__ __forceinline__ void st_vec(double* p, d2 v) {
    if (POLICY) __builtin_nontemporal_store(v, reinterpret_cast<d2*>(p));
    else *reinterpret_cast<d2*>(p) = v;
}

constexpr int next_pow2(int v) { return v <= 1 ? 1 : (v <= 2 ? 2 : (v <= 4 ? 4 : 8)); }


// Kernel durations for bioen_hip_kernel_stats: the two events ride on the kernel's own dispatch packet
// (hipExtLaunchKernelGGL) instead of being recorded around it -- an event record is a barrier packet of its own, and
// the pair cost ~6 us of idle queue per launch (rocprofv3: 4 x 2.4 ms over the headline sweep).  A launch inside the
// scope of a TimedLaunch goes through BIOEN_LAUNCH_TIMED; with the stats off both events are NULL = a plain launch.
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
        t.cur_a = pr.a;
        t.cur_b = pr.b;
    }
    ~TimedLaunch() {
        if (!on) return;
        c->timer.cur_a = c->timer.cur_b = nullptr;
        c->timer.pending.push_back(pr);
    }
};

#define BIOEN_LAUNCH_TIMED(ctx, kern, grid, block, lds, ...) \
    hipExtLaunchKernelGGL(kern, grid, block, lds, (ctx)->stream, (ctx)->timer.cur_a, (ctx)->timer.cur_b, 0, __VA_ARGS__)

}  // namespace bioen
