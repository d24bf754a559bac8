// Forces method: N-vector kernels of the four-pass paths (gfx950; M > 1024).  Two families: the r01 kernels of the
// streaming fallback (no row panels: BIOEN_HIP_PANELS=0 or no memory for them; unsharded contexts only -- their partial
// sums are plain per-block arrays), and the canonical-segment kernels of the row-panel path (r05, below), whose sums
// are those of ctx.hpp's eight segments on any number of ranks.
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
    if (threadIdx.x == 0) r.part[a][(size_t)P_MAX * kPartStride + blockIdx.x] = mx;
}

// _get_weights_from_forces (c_bioen_kernels_forces.c:152-176), first half
__global__ __launch_bounds__(kBlock) void k_forces_exp(ForcesRound r, const double* __restrict__ w0, int n, int np) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double* __restrict__ xj = r.a[a];
    double* __restrict__ w = r.w[a];
    double* pa = r.part[a];
    const double xmax = max_partials(pa + (size_t)P_MAX * kPartStride, np, sh);
    double s = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const double ev = w0[j] * exp(xj[j] - xmax);
        w[j] = ev;
        s += ev;
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) pa[(size_t)P_SUM * kPartStride + blockIdx.x] = s;
}

// second half + relative entropy terms (c_bioen_kernels_forces.c:246-258)
__global__ __launch_bounds__(kBlock) void k_forces_norm(ForcesRound r, const double* __restrict__ w0, int n, int np) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    double* __restrict__ w = r.w[a];
    double* pa = r.part[a];
    const double inv = 1.0 / sum_partials(pa + (size_t)P_SUM * kPartStride, np, sh);
    double kl = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const double wv = inv * w[j];
        w[j] = wv;
        const double w0v = w0[j];
        if (wv >= DBL_MIN && w0v >= DBL_MIN) kl = fma(log(wv) - log(w0v), wv, kl);
    }
    kl = block_sum(kl, sh);
    if (threadIdx.x == 0) pa[(size_t)P_KL * kPartStride + blockIdx.x] = kl;
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

__global__ __launch_bounds__(kBlock) void k_forces_scalars(ForcesRound r, int npchi, int npkl) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double* pa = r.part[a];
    const double chi = sum_partials(pa + (size_t)P_CHI * kPartStride, npchi, sh);
    const double kl = sum_partials(pa + (size_t)P_KL * kPartStride, npkl, sh);
    if (threadIdx.x == 0) {
        double* sc = r.scal[a];
        sc[S_CHI] = chi;
        sc[S_KL] = kl;
        sc[S_F] = kl * r.theta[a] + 0.5 * chi;
    }
}


// ---- canonical segments (r05): the four passes over row panels (M > 1024) on any number of ranks --------------------
// The softmax over all structures is merged segment by segment exactly as in the two-pass strip path
// (kernels_strip.hip: k_forces_blockstats) and in the log-weights rounds: e_j = w0_j exp(x_j - m_v) with the maximum
// m_v of the block's own SEGMENT, the block's shares of sum e and sum e x, and m_v go to the X_EXP stage in
// k_logw_exp's layout, ride on the segment's part of X_YBAR behind the row sums of yTilde . e (k_fwd_rows_local_t) and
// k_rows_combine<true> finishes: S_LOGS = M + log sum, S_P = sum_j w_j x_j, KL = S_P - S_LOGS, f.  One exchange.
__global__ __launch_bounds__(kBlock) void k_forces_seg_exp(ForcesRound r, const double* __restrict__ w0, int n, Xch xmx, Xch xo) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double* __restrict__ x = r.a[a];
    double* __restrict__ e = r.w[a];
    const double gmax = xmax_local<1>(xmx, a, 0);
    double s = 0.0, pp = 0.0;
    const SegPos sp = seg_pos(xo.npl, xo.segcols, n);
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(xo.npl)) {
        const d2 xv = ld_vec(x + j);
        const d2 wv = ld_vec(w0 + j);
        const bool two = j + 1 < sp.jend;
        d2 ev;
        ev.x = wv.x * exp(xv.x - gmax);
        ev.y = two ? wv.y * exp(xv.y - gmax) : 0.0;        // the padding of e stays zero (k_scale_w)
        *reinterpret_cast<d2*>(e + j) = ev;
        s += ev.x;
        pp = fma(ev.x, xv.x, pp);
        s += ev.y;
        pp = fma(ev.y, two ? xv.y : 0.0, pp);
    }
    s = block_sum(s, sh);
    pp = block_sum(pp, sh);
    if (threadIdx.x == 0) {
        xput<3>(xo, a, 0, s);
        xput<3>(xo, a, 1, pp);
        if (sp.b == 0) xput<3>(xo, a, 2, gmax);
        if (blockIdx.x == 0) r.scal[a][S_LOGS0] = 0.0;     // no prior constant in this method (k_rows_combine adds it)
    }
}

// k_forces_t on the NORMALISED weights (k_scale_w has run), with the block's share of T = sum_j t_j into a one-array stage: the
// last matrix pass forms sum_j (Y_ij - c_i) t_j on the centred copy and the gradient takes (ybar_i - c_i) T off, as the
// two-pass strip kernels do
__global__ __launch_bounds__(kBlock) void k_forces_seg_t(ForcesRound r, const double* __restrict__ w0, int n, Xch xo) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double* __restrict__ w = r.w[a];
    const double* __restrict__ b = r.a[a];
    double* __restrict__ t = r.t[a];
    const double theta = r.theta[a];
    double s = 0.0;
    const SegPos sp = seg_pos(xo.npl, xo.segcols, n);
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(xo.npl)) {
        const d2 wv = ld_vec(w + j), w0v = ld_vec(w0 + j), bv = ld_vec(b + j);
        double d0 = 1.0, d1 = 1.0;
        if (wv.x >= DBL_MIN && w0v.x >= DBL_MIN) d0 += log(wv.x) - log(w0v.x);
        if (wv.y >= DBL_MIN && w0v.y >= DBL_MIN) d1 += log(wv.y) - log(w0v.y);
        d2 tv;
        tv.x = (d0 * theta + bv.x) * wv.x;
        tv.y = (j + 1 < sp.jend) ? (d1 * theta + bv.y) * wv.y : 0.0;
        *reinterpret_cast<d2*>(t + j) = tv;
        s += tv.x;
        s += tv.y;
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) xput<1>(xo, a, 0, s);
}

// T_v, the total of a local segment's shares -> share 0 of that segment's `seg_sets` P_KL shares, the others zero
// (k_fwd_rows_forces_grad_t totals them per segment); grid (problems, local segments)
__global__ __launch_bounds__(kBlock) void k_forces_seg_tsum(Xch xi, ForcesRound r, int seg_sets) {
    const int a = blockIdx.x, v = blockIdx.y;
    const double s = xsum_seg<1>(xi, xi.rank + v, a, 0);
    double* share = r.part[a] + (size_t)P_KL * kPartStride + (size_t)v * seg_sets;
    for (int b = threadIdx.x; b < seg_sets; b += kBlock) share[b] = b == 0 ? s : 0.0;
}

void launch_forces_seg_exp(bioen_hip_ctx* c, const ForcesRound& r) {
    hipLaunchKernelGGL(k_forces_seg_exp, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                       make_xch(c, X_MAX, r.n * vec_grid(c)), make_xch(c, X_EXP, 3 * r.n * vec_grid(c)));
}

void launch_forces_seg_t(bioen_hip_ctx* c, const ForcesRound& r, int seg_sets) {
    const Xch xo = make_xch(c, X_MAX, r.n * vec_grid(c));       // the block maxima have been consumed by k_forces_seg_exp
    hipLaunchKernelGGL(k_forces_seg_t, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n, xo);
    hipLaunchKernelGGL(k_forces_seg_tsum, dim3(r.n, c->vr), dim3(kBlock), 0, c->stream, xo, r, seg_sets);
}

// ---- forces ------------------------------------------------------------------------------------
void launch_forces_max(bioen_hip_ctx* c, const ForcesRound& r) {
    hipLaunchKernelGGL(k_forces_max, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->n);
}

void launch_forces_exp(bioen_hip_ctx* c, const ForcesRound& r) {
    hipLaunchKernelGGL(k_forces_exp, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                       vec_blocks(c));
}

void launch_forces_norm(bioen_hip_ctx* c, const ForcesRound& r) {
    hipLaunchKernelGGL(k_forces_norm, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                       vec_blocks(c));
}

void launch_forces_t(bioen_hip_ctx* c, const ForcesRound& r) {
    hipLaunchKernelGGL(k_forces_t, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n);
}

void launch_forces_scalars(bioen_hip_ctx* c, const ForcesRound& r) {
    hipLaunchKernelGGL(k_forces_scalars, dim3(1, r.n), dim3(kBlock), 0, c->stream, r, combine_grid(c), vec_blocks(c));
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
