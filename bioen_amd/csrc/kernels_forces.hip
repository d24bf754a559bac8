// Forces method: N-vector kernels of the streaming path (gfx950; M > 1024 without row panels, bioen_hip_forces_weights --
// unsharded contexts only: their partial sums are plain per-block arrays, not canonical segments).
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
    double* share = r.part[a] + (size_t)P_KL * kPartStride;
    for (int b = threadIdx.x; b < sets; b += kBlock) share[b] = b == 0 ? s : 0.0;
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

void launch_forces_t(bioen_hip_ctx* c, const ForcesRound& r, int tsum_sets) {
    if (tsum_sets > 0) {
        const int g = vec_blocks(c);
        double* tpart = c->xbuf[X_GRAM];            // kGramDots values per block and problem: room for one; idle in the forces method
        hipLaunchKernelGGL(k_forces_t, dim3(g, r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n, tpart);
        hipLaunchKernelGGL(k_forces_tsum, dim3(r.n), dim3(kBlock), 0, c->stream, tpart, g, r, tsum_sets);
        return;
    }
    hipLaunchKernelGGL(k_forces_t, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                       static_cast<double*>(nullptr));
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
