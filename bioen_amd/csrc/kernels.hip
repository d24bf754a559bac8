// gfx950 (CDNA4, wave64) kernels of the BioEn log-weights / forces hot path.
//
// Two kernels touch the M x N matrix and carry >99 % of the bytes:
//   k_fwd_partial : ybar = yTilde . v      (replaces _bioen_chi_squared's GEMV,
//                                           c_bioen_common.c:76-86, and _getAve,
//                                           c_bioen_kernels_forces.c:93-109)
//   k_adj         : a    = yTilde^T . u    (replaces the transposed-cache walks of
//                                           c_bioen_kernels_logw.c:185-205 and
//                                           c_bioen_kernels_forces.c:127-150,300-320)
// Both stream the row-major matrix exactly once with 16-byte-per-lane loads
// (one aligned KiB per wave instruction) straight into registers -- the operand
// is read once and not shared across waves, so an LDS round trip would be pure
// overhead -- and reduce in a fixed order (bitwise reproducible run to run).
// Everything else is O(N) or O(M) glue that keeps all vectors and scalars in
// HBM so that only line-search decisions cross PCIe.
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

// Sum an array of per-block partials written by the PREVIOUS kernel.  Every
// block of the consumer kernel does this redundantly in its prologue (<= 8 KiB,
// L2 resident), which replaces a separate 1-block "finalise" launch.
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

template <bool NT>
__device__ __forceinline__ d2 ldg2(const double* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const d2*>(p));
    return *reinterpret_cast<const d2*>(p);
}

// ------------------------------------------------------------------------------
// forward pass: partial[row * ctiles + tile] = sum_{j in tile} Y[row][j] v[j]
//   block = 4 waves stacked over rows, R rows per wave; a wave walks its column
//   tile in 128-column (1 KiB) steps, two steps in flight (2*R loads of 1 KiB).
// ------------------------------------------------------------------------------
// CENTER: sum_j (Y[row][j] - ybar[row]) v[j] -- the centred form the reference uses for the
// forces gradient (c_bioen_kernels_forces.c:330-338); avoids the cancellation of
// (Y v) - ybar (1.v) when a gradient component is small against its two terms.
template <int R, bool NT, bool CENTER>
__global__ __launch_bounds__(kBlock) void k_fwd_partial(const double* __restrict__ Y, size_t ld,
                                                        const double* __restrict__ v,
                                                        const double* __restrict__ ybar,
                                                        double* __restrict__ partial, int ctiles,
                                                        int steps_per_tile, int total_steps) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x;
    const int row0 = (blockIdx.y * kWaves + wave) * R;
    int s = tile * steps_per_tile;
    int s_end = s + steps_per_tile;
    if (s_end > total_steps) s_end = total_steps;

    const double* yp = Y + (size_t)row0 * ld + (size_t)s * 128 + lane * 2;
    const double* vp = v + (size_t)s * 128 + lane * 2;

    double acc[R], yb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        acc[r] = 0.0;
        yb[r] = CENTER ? ybar[row0 + r] : 0.0;
    }

    for (; s + 2 <= s_end; s += 2) {
        d2 y0[R], y1[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            y0[r] = ldg2<NT>(yp + (size_t)r * ld);
            y1[r] = ldg2<NT>(yp + (size_t)r * ld + 128);
        }
        const d2 v0 = *reinterpret_cast<const d2*>(vp);
        const d2 v1 = *reinterpret_cast<const d2*>(vp + 128);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (CENTER) {
                acc[r] = fma(y0[r].x - yb[r], v0.x, acc[r]);
                acc[r] = fma(y0[r].y - yb[r], v0.y, acc[r]);
                acc[r] = fma(y1[r].x - yb[r], v1.x, acc[r]);
                acc[r] = fma(y1[r].y - yb[r], v1.y, acc[r]);
            } else {
                acc[r] = fma(y0[r].x, v0.x, acc[r]);
                acc[r] = fma(y0[r].y, v0.y, acc[r]);
                acc[r] = fma(y1[r].x, v1.x, acc[r]);
                acc[r] = fma(y1[r].y, v1.y, acc[r]);
            }
        }
        yp += 256;
        vp += 256;
    }
    if (s < s_end) {
        d2 y0[R];
#pragma unroll
        for (int r = 0; r < R; ++r) y0[r] = ldg2<NT>(yp + (size_t)r * ld);
        const d2 v0 = *reinterpret_cast<const d2*>(vp);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (CENTER) {
                acc[r] = fma(y0[r].x - yb[r], v0.x, acc[r]);
                acc[r] = fma(y0[r].y - yb[r], v0.y, acc[r]);
            } else {
                acc[r] = fma(y0[r].x, v0.x, acc[r]);
                acc[r] = fma(y0[r].y, v0.y, acc[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const double tot = wave_sum(acc[r]);
        if (lane == r) partial[(size_t)(row0 + r) * ctiles + tile] = tot;
    }
}

// reduce the column tiles of one row per wave (fixed order), mode 0:
//   ybar_i, r_i = ybar_i - YT_i ; per-block partials of sum r^2 and sum ybar r
__global__ __launch_bounds__(kBlock) void k_fwd_rows_residual(const double* __restrict__ partial, int ctiles,
                                                              int mp, const double* __restrict__ YT,
                                                              double* __restrict__ ybar, double* __restrict__ r,
                                                              double* __restrict__ pchi, double* __restrict__ pc) {
    __shared__ double sh[2][kWaves];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double chi = 0.0, cc = 0.0;
    for (int row = blockIdx.x * kWaves + wave; row < mp; row += gridDim.x * kWaves) {
        const double* p = partial + (size_t)row * ctiles;
        double s = 0.0;
        for (int k = lane; k < ctiles; k += 64) s += p[k];
        s = wave_sum(s);
        const double res = s - YT[row];
        if (lane == 0) {
            ybar[row] = s;
            r[row] = res;
        }
        chi += res * res;
        cc += s * res;
    }
    if (lane == 0) {
        sh[0][wave] = chi;
        sh[1][wave] = cc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        pchi[blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        pc[blockIdx.x] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}

// mode 1 (forces gradient, c_bioen_kernels_forces.c:330-338): the partials already hold the
// centred sums  sum_j (Y_ij - ybar_i) t_j ; only the column tiles remain to be added up.
__global__ __launch_bounds__(kBlock) void k_fwd_rows_forces_grad(const double* __restrict__ partial, int ctiles,
                                                                 int mp, double* __restrict__ gm) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * kWaves + wave; row < mp; row += gridDim.x * kWaves) {
        const double* p = partial + (size_t)row * ctiles;
        double s = 0.0;
        for (int k = lane; k < ctiles; k += 64) s += p[k];
        s = wave_sum(s);
        if (lane == 0) gm[row] = s;
    }
}

// ------------------------------------------------------------------------------
// adjoint pass: out[j] = sum_i Y[i][j] u[i]
//   block = one 128-column strip (a lane owns 2 adjacent columns = 16 B), the 4
//   waves split the rows; U rows (U KiB) in flight per wave; u[i] is wave-uniform
//   and comes through the scalar cache.
// ------------------------------------------------------------------------------
// CENTER: out[j] = sum_i u[i] (Y[i][j] - ybar[i]) -- the reference's centred gradient sum
// (c_bioen_kernels_logw.c:185-195); padded columns then hold -u.ybar, which nobody reads.
template <int U, bool NT, bool CENTER>
__global__ __launch_bounds__(kBlock) void k_adj(const double* __restrict__ Y, size_t ld, int rows_per_wave,
                                                const double* __restrict__ u, const double* __restrict__ ybar,
                                                double* __restrict__ out) {
    __shared__ d2 red[kWaves][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t col = (size_t)blockIdx.x * 128 + lane * 2;
    const int r0 = wave * rows_per_wave;
    const double* yp = Y + (size_t)r0 * ld + col;
    const double* up = u + r0;
    const double* bp = ybar + r0;

    d2 acc0 = {0.0, 0.0}, acc1 = {0.0, 0.0};
    for (int i = 0; i < rows_per_wave; i += U) {
        d2 y[U];
#pragma unroll
        for (int k = 0; k < U; ++k) y[k] = ldg2<NT>(yp + (size_t)k * ld);
#pragma unroll
        for (int k = 0; k < U; k += 2) {
            const double u0 = up[i + k], u1 = up[i + k + 1];
            if (CENTER) {
                const double b0 = bp[i + k], b1 = bp[i + k + 1];
                acc0.x = fma(y[k].x - b0, u0, acc0.x);
                acc0.y = fma(y[k].y - b0, u0, acc0.y);
                acc1.x = fma(y[k + 1].x - b1, u1, acc1.x);
                acc1.y = fma(y[k + 1].y - b1, u1, acc1.y);
            } else {
                acc0.x = fma(y[k].x, u0, acc0.x);
                acc0.y = fma(y[k].y, u0, acc0.y);
                acc1.x = fma(y[k + 1].x, u1, acc1.x);
                acc1.y = fma(y[k + 1].y, u1, acc1.y);
            }
        }
        yp += (size_t)U * ld;
    }
    d2 acc = {acc0.x + acc1.x, acc0.y + acc1.y};
    red[wave][lane] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        const d2 a0 = red[0][lane], a1 = red[1][lane], a2 = red[2][lane], a3 = red[3][lane];
        d2 o = {(a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y)};
        *reinterpret_cast<d2*>(out + col) = o;
    }
}

// ------------------------------------------------------------------------------
// log-weights N-vector kernels
// ------------------------------------------------------------------------------
// x = xp + stp * d ; block maxima of x
__global__ __launch_bounds__(kBlock) void k_trial(double* __restrict__ x, const double* __restrict__ xp,
                                                  const double* __restrict__ d, double stp, int n,
                                                  double* __restrict__ pmax) {
    __shared__ double sh[kWaves];
    double mx = -DBL_MAX;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const double v = fma(stp, d[j], xp[j]);
        x[j] = v;
        mx = fmax(mx, v);
    }
    mx = block_max(mx, sh);
    if (threadIdx.x == 0) pmax[blockIdx.x] = mx;
}

__global__ __launch_bounds__(kBlock) void k_max(const double* __restrict__ v, int n, double* __restrict__ pmax) {
    __shared__ double sh[kWaves];
    double mx = -DBL_MAX;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) mx = fmax(mx, v[j]);
    mx = block_max(mx, sh);
    if (threadIdx.x == 0) pmax[blockIdx.x] = mx;
}

// _get_weights (c_bioen_kernels_logw.c:55-94) with a max shift, first half:
//   e_j = exp(x_j - max) ; partials of sum e and sum e (x - G)   (prior, :96-127)
__global__ __launch_bounds__(kBlock) void k_logw_exp(const double* __restrict__ x, const double* __restrict__ G,
                                                     int n, const double* __restrict__ pmax, int np,
                                                     double* __restrict__ e, double* __restrict__ psum,
                                                     double* __restrict__ ppp) {
    __shared__ double sh[kWaves];
    const double gmax = max_partials(pmax, np, sh);
    double s = 0.0, pp = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const double xv = x[j];
        const double ev = exp(xv - gmax);
        e[j] = ev;
        s += ev;
        pp = fma(ev, xv - G[j], pp);
    }
    s = block_sum(s, sh);
    pp = block_sum(pp, sh);
    if (threadIdx.x == 0) {
        psum[blockIdx.x] = s;
        ppp[blockIdx.x] = pp;
    }
}

// second half: w = e / S ; scal[S_LOGS] = max + log S ; scal[S_P] = sum e (x-G) / S
__global__ __launch_bounds__(kBlock) void k_logw_norm(double* __restrict__ w, int n, const double* __restrict__ pmax,
                                                      const double* __restrict__ psum,
                                                      const double* __restrict__ ppp, int np,
                                                      double* __restrict__ scal) {
    __shared__ double sh[kWaves];
    const double S = sum_partials(psum, np, sh);
    const double inv = 1.0 / S;
    if (blockIdx.x == 0) {
        const double gmax = max_partials(pmax, np, sh);
        const double PP = sum_partials(ppp, np, sh);
        if (threadIdx.x == 0) {
            scal[S_LOGS] = gmax + log(S);
            scal[S_P] = PP * inv;
        }
    }
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) w[j] *= inv;
}

// log s0 = log sum exp(G): constant per problem, computed once (the reference
// recomputes it at every evaluation, c_bioen_kernels_logw.c:122)
__global__ __launch_bounds__(kBlock) void k_logsumexp1(const double* __restrict__ G, int n, double* __restrict__ scal,
                                                       int slot) {
    __shared__ double sh[kWaves];
    double mx = -DBL_MAX;
    for (int j = threadIdx.x; j < n; j += kBlock) mx = fmax(mx, G[j]);
    mx = block_max(mx, sh);
    double s = 0.0;
    for (int j = threadIdx.x; j < n; j += kBlock) s += exp(G[j] - mx);
    s = block_sum(s, sh);
    if (threadIdx.x == 0) scal[slot] = mx + log(s);
}

// f = theta (P - log s + log s0) + 0.5 sum r^2       (c_bioen_kernels_logw.c:124-147)
__global__ __launch_bounds__(kBlock) void k_logw_scalars(const double* __restrict__ pchi,
                                                         const double* __restrict__ pc, int np, double theta,
                                                         double* __restrict__ scal) {
    __shared__ double sh[kWaves];
    const double chi = sum_partials(pchi, np, sh);
    const double c = sum_partials(pc, np, sh);
    if (threadIdx.x == 0) {
        scal[S_CHI] = chi;
        scal[S_C] = c;
        scal[S_F] = theta * (scal[S_P] - scal[S_LOGS] + scal[S_LOGS0]) + 0.5 * chi;
    }
}

// gradient epilogue (c_bioen_kernels_logw.c:207-218):
//   g_k = w_k [ theta (x_k - G_k - P) + a_k ],  a_k = sum_i r_i (yTilde_ik - ybar_i)  (centred adjoint)
// plus the three dot products the line search / convergence test needs.
__global__ __launch_bounds__(kBlock) void k_logw_grad(const double* __restrict__ x, const double* __restrict__ G,
                                                      const double* __restrict__ w, const double* __restrict__ a,
                                                      const double* __restrict__ d, double theta,
                                                      const double* __restrict__ scal, int n,
                                                      double* __restrict__ g, double* __restrict__ pdg,
                                                      double* __restrict__ pgg, double* __restrict__ pxx) {
    __shared__ double sh[kWaves];
    const double P = scal[S_P];
    double dg = 0.0, gg = 0.0, xx = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const double xv = x[j];
        const double gv = w[j] * (theta * ((xv - G[j]) - P) + a[j]);
        g[j] = gv;
        dg = fma(gv, d[j], dg);
        gg = fma(gv, gv, gg);
        xx = fma(xv, xv, xx);
    }
    dg = block_sum(dg, sh);
    gg = block_sum(gg, sh);
    xx = block_sum(xx, sh);
    if (threadIdx.x == 0) {
        pdg[blockIdx.x] = dg;
        pgg[blockIdx.x] = gg;
        pxx[blockIdx.x] = xx;
    }
}

__global__ __launch_bounds__(kBlock) void k_finish_eval(const double* __restrict__ pdg, const double* __restrict__ pgg,
                                                        const double* __restrict__ pxx,
                                                        const double* __restrict__ pdginit, int np,
                                                        double* __restrict__ scal) {
    __shared__ double sh[kWaves];
    const double dg = sum_partials(pdg, np, sh);
    const double gg = sum_partials(pgg, np, sh);
    const double xx = sum_partials(pxx, np, sh);
    const double di = sum_partials(pdginit, np, sh);
    if (threadIdx.x == 0) {
        scal[S_DG] = dg;
        scal[S_GG] = gg;
        scal[S_XX] = xx;
        scal[S_DGINIT] = di;
    }
}

// ------------------------------------------------------------------------------
// forces N-vector kernels
// ------------------------------------------------------------------------------
// _get_weights_from_forces (c_bioen_kernels_forces.c:152-176), first half
__global__ __launch_bounds__(kBlock) void k_forces_exp(const double* __restrict__ xj, const double* __restrict__ w0,
                                                       int n, const double* __restrict__ pmax, int np,
                                                       double* __restrict__ w, double* __restrict__ psum) {
    __shared__ double sh[kWaves];
    const double xmax = max_partials(pmax, np, sh);
    double s = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const double ev = w0[j] * exp(xj[j] - xmax);
        w[j] = ev;
        s += ev;
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) psum[blockIdx.x] = s;
}

// second half + relative entropy terms (c_bioen_kernels_forces.c:246-258)
__global__ __launch_bounds__(kBlock) void k_forces_norm(double* __restrict__ w, const double* __restrict__ w0, int n,
                                                        const double* __restrict__ psum, int np,
                                                        double* __restrict__ pkl) {
    __shared__ double sh[kWaves];
    const double inv = 1.0 / sum_partials(psum, np, sh);
    double kl = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const double wv = inv * w[j];
        w[j] = wv;
        const double w0v = w0[j];
        if (wv >= DBL_MIN && w0v >= DBL_MIN) kl = fma(log(wv) - log(w0v), wv, kl);
    }
    kl = block_sum(kl, sh);
    if (threadIdx.x == 0) pkl[blockIdx.x] = kl;
}

// t_j = (theta (1 + log w_j - log w0_j) + b_j) w_j     (c_bioen_kernels_forces.c:320-328)
__global__ __launch_bounds__(kBlock) void k_forces_t(const double* __restrict__ w, const double* __restrict__ w0,
                                                     const double* __restrict__ b, double theta, int n,
                                                     double* __restrict__ t, double* __restrict__ ptsum) {
    __shared__ double sh[kWaves];
    double ts = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const double wv = w[j], w0v = w0[j];
        double dd = 1.0;
        if (wv >= DBL_MIN && w0v >= DBL_MIN) dd += log(wv) - log(w0v);
        const double tv = (dd * theta + b[j]) * wv;
        t[j] = tv;
        ts += tv;
    }
    ts = block_sum(ts, sh);
    if (threadIdx.x == 0) ptsum[blockIdx.x] = ts;
}

__global__ __launch_bounds__(kBlock) void k_forces_scalars(const double* __restrict__ pchi, int npchi,
                                                           const double* __restrict__ pkl, int npkl, double theta,
                                                           double* __restrict__ scal) {
    __shared__ double sh[kWaves];
    const double chi = sum_partials(pchi, npchi, sh);
    const double kl = sum_partials(pkl, npkl, sh);
    if (threadIdx.x == 0) {
        scal[S_CHI] = chi;
        scal[S_KL] = kl;
        scal[S_F] = kl * theta + 0.5 * chi;
    }
}

// ------------------------------------------------------------------------------
// L-BFGS vector kernels (liblbfgs lbfgs.c:543-615 with every scalar device-resident)
// ------------------------------------------------------------------------------
// s = x - xp, y = g - gp (lbfgs.c:549-551); partials of y.s and y.y (:559-561)
__global__ __launch_bounds__(kBlock) void k_update_sy(const double* __restrict__ x, const double* __restrict__ xp,
                                                      const double* __restrict__ g, const double* __restrict__ gp,
                                                      int n, double* __restrict__ s, double* __restrict__ y,
                                                      double* __restrict__ pys, double* __restrict__ pyy) {
    __shared__ double sh[kWaves];
    double ys = 0.0, yy = 0.0;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        const double sv = x[j] - xp[j];
        const double yv = g[j] - gp[j];
        s[j] = sv;
        y[j] = yv;
        ys = fma(yv, sv, ys);
        yy = fma(yv, yv, yy);
    }
    ys = block_sum(ys, sh);
    yy = block_sum(yy, sh);
    if (threadIdx.x == 0) {
        pys[blockIdx.x] = ys;
        pyy[blockIdx.x] = yy;
    }
}

// One fused step of the two-loop recursion (lbfgs.c:571-598).  The dot product a
// step needs was left as per-block partials by the previous step; every block
// re-reduces them (fixed order) in its prologue, so a step is ONE launch:
//   mode 0: d = -gp                                   [+ finalise y.s, y.y of slot `hist`]
//   mode 1: alpha_h = (S_h . d) / ys_h ; d -= alpha_h Y_h        (first loop)
//   mode 2: beta = (Y_h . d) / ys_h ; d += (alpha_h - beta) S_h  (second loop)
//   scale : d *= ys / yy   after the update (last step of the first loop)
// and in the same sweep  out_partials = vdot . d  for the next step (or gp . d,
// the next line search's initial slope).
__global__ __launch_bounds__(kBlock) void k_recur(int mode, int hist, int scale, int finalize_sy,
                                                  double* __restrict__ d, const double* __restrict__ gp,
                                                  const double* __restrict__ vaxpy, const double* __restrict__ vdot,
                                                  const double* __restrict__ pin, const double* __restrict__ pys,
                                                  const double* __restrict__ pyy, int np, int n,
                                                  double* __restrict__ scal, double* __restrict__ pout) {
    __shared__ double sh[kWaves];
    double coef = 0.0, sc = 1.0;
    if (mode == 0) {
        if (finalize_sy && blockIdx.x == 0) {
            const double ys = sum_partials(pys, np, sh);
            const double yy = sum_partials(pyy, np, sh);
            if (threadIdx.x == 0) {
                scal[S_YSH + hist] = ys;
                scal[S_YS] = ys;
                scal[S_YY] = yy;
            }
        }
    } else {
        const double dot = sum_partials(pin, np, sh);
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
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {
        double dv;
        if (mode == 0) {
            dv = -gp[j];
        } else {
            dv = fma(coef, vaxpy[j], d[j]);
            if (scale) dv *= sc;
        }
        d[j] = dv;
        if (vdot) acc = fma(vdot[j], dv, acc);
    }
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0 && pout) pout[blockIdx.x] = acc;
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
            const unsigned long long ctr = (unsigned long long)i * (unsigned long long)((n + 1) / 2) + jp;
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
    long long b = ((long long)c->n + 4 * kBlock - 1) / (4 * kBlock);
    if (b < 1) b = 1;
    if (b > kMaxPartials) b = kMaxPartials;
    return (int)b;
}

static int rows_grid(const bioen_hip_ctx* c) {
    int b = c->mp / kWaves;
    if (b > kMaxPartials) b = kMaxPartials;
    return b;
}

struct TimedLaunch {
    bioen_hip_ctx* c;
    KernelTimer::Pair pr;
    bool on;
    TimedLaunch(bioen_hip_ctx* ctx, int which) : c(ctx), on(ctx->timer.enabled) {
        if (!on) return;
        KernelTimer& t = c->timer;
        if (!t.pool.empty()) {
            pr = t.pool.back();
            t.pool.pop_back();
        } else {
            hipEventCreate(&pr.a);
            hipEventCreate(&pr.b);
        }
        pr.which = which;
        hipEventRecord(pr.a, c->stream);
    }
    ~TimedLaunch() {
        if (!on) return;
        hipEventRecord(pr.b, c->stream);
        c->timer.pending.push_back(pr);
    }
};

template <bool NT, bool CENTER>
static void launch_fwd_t(bioen_hip_ctx* c, const double* v) {
    const int total_steps = (int)(c->ld / 128);
    dim3 grid(c->fwd_ctiles, c->mp / kRowAlign);
    hipLaunchKernelGGL((k_fwd_partial<8, NT, CENTER>), grid, dim3(kBlock), 0, c->stream, c->Y, c->ld, v, c->ybar,
                       c->fwd_partial, c->fwd_ctiles, c->fwd_steps, total_steps);
}

void launch_fwd_partial(bioen_hip_ctx* c, const double* v, bool centred) {
    TimedLaunch tl(c, 0);
    if (c->nontemporal) {
        if (centred) launch_fwd_t<true, true>(c, v); else launch_fwd_t<true, false>(c, v);
    } else {
        if (centred) launch_fwd_t<false, true>(c, v); else launch_fwd_t<false, false>(c, v);
    }
}

void launch_fwd_rows_residual(bioen_hip_ctx* c) {
    hipLaunchKernelGGL(k_fwd_rows_residual, dim3(rows_grid(c)), dim3(kBlock), 0, c->stream, c->fwd_partial,
                       c->fwd_ctiles, c->mp, c->YT, c->ybar, c->r, part(c, P_CHI), part(c, P_C));
}

void launch_fwd_rows_forces_grad(bioen_hip_ctx* c) {
    hipLaunchKernelGGL(k_fwd_rows_forces_grad, dim3(rows_grid(c)), dim3(kBlock), 0, c->stream, c->fwd_partial,
                       c->fwd_ctiles, c->mp, c->gm);
}

template <bool NT, bool CENTER>
static void launch_adj_t(bioen_hip_ctx* c, const double* u, double* out) {
    dim3 grid((unsigned)(c->ld / 128));
    hipLaunchKernelGGL((k_adj<8, NT, CENTER>), grid, dim3(kBlock), 0, c->stream, c->Y, c->ld, c->mp / kWaves, u,
                       c->ybar, out);
}

void launch_adj(bioen_hip_ctx* c, const double* u, double* out, bool centred) {
    TimedLaunch tl(c, 1);
    if (c->nontemporal) {
        if (centred) launch_adj_t<true, true>(c, u, out); else launch_adj_t<true, false>(c, u, out);
    } else {
        if (centred) launch_adj_t<false, true>(c, u, out); else launch_adj_t<false, false>(c, u, out);
    }
}

void launch_trial(bioen_hip_ctx* c, double stp) {
    hipLaunchKernelGGL(k_trial, dim3(vec_grid(c)), dim3(kBlock), 0, c->stream, c->x, c->xp, c->d, stp, c->n,
                       part(c, P_MAX));
}

void launch_max(bioen_hip_ctx* c, const double* v) {
    hipLaunchKernelGGL(k_max, dim3(vec_grid(c)), dim3(kBlock), 0, c->stream, v, c->n, part(c, P_MAX));
}

void launch_logw_exp(bioen_hip_ctx* c) {
    hipLaunchKernelGGL(k_logw_exp, dim3(vec_grid(c)), dim3(kBlock), 0, c->stream, c->x, c->fixed, c->n,
                       part(c, P_MAX), vec_grid(c), c->w, part(c, P_SUM), part(c, P_PP));
}

void launch_logw_norm(bioen_hip_ctx* c) {
    hipLaunchKernelGGL(k_logw_norm, dim3(vec_grid(c)), dim3(kBlock), 0, c->stream, c->w, c->n, part(c, P_MAX),
                       part(c, P_SUM), part(c, P_PP), vec_grid(c), c->scal);
}

void launch_logw_logs0(bioen_hip_ctx* c) {
    hipLaunchKernelGGL(k_logsumexp1, dim3(1), dim3(kBlock), 0, c->stream, c->fixed, c->n, c->scal, (int)S_LOGS0);
}

void launch_logw_scalars(bioen_hip_ctx* c, double theta) {
    hipLaunchKernelGGL(k_logw_scalars, dim3(1), dim3(kBlock), 0, c->stream, part(c, P_CHI), part(c, P_C),
                       rows_grid(c), theta, c->scal);
}

void launch_logw_grad(bioen_hip_ctx* c, double theta) {
    hipLaunchKernelGGL(k_logw_grad, dim3(vec_grid(c)), dim3(kBlock), 0, c->stream, c->x, c->fixed, c->w, c->a,
                       c->d, theta, c->scal, c->n, c->g, part(c, P_DG), part(c, P_GG), part(c, P_XX));
}

void launch_finish_eval(bioen_hip_ctx* c) {
    hipLaunchKernelGGL(k_finish_eval, dim3(1), dim3(kBlock), 0, c->stream, part(c, P_DG), part(c, P_GG),
                       part(c, P_XX), part(c, P_DGINIT), vec_grid(c), c->scal);
}

void launch_forces_exp(bioen_hip_ctx* c, const double* xj) {
    hipLaunchKernelGGL(k_forces_exp, dim3(vec_grid(c)), dim3(kBlock), 0, c->stream, xj, c->fixed, c->n,
                       part(c, P_MAX), vec_grid(c), c->w, part(c, P_SUM));
}

void launch_forces_norm(bioen_hip_ctx* c) {
    hipLaunchKernelGGL(k_forces_norm, dim3(vec_grid(c)), dim3(kBlock), 0, c->stream, c->w, c->fixed, c->n,
                       part(c, P_SUM), vec_grid(c), part(c, P_KL));
}

void launch_forces_t(bioen_hip_ctx* c, double theta) {
    hipLaunchKernelGGL(k_forces_t, dim3(vec_grid(c)), dim3(kBlock), 0, c->stream, c->w, c->fixed, c->a, theta,
                       c->n, c->t, part(c, P_TSUM));
}

void launch_forces_scalars(bioen_hip_ctx* c, double theta) {
    hipLaunchKernelGGL(k_forces_scalars, dim3(1), dim3(kBlock), 0, c->stream, part(c, P_CHI), rows_grid(c),
                       part(c, P_KL), vec_grid(c), theta, c->scal);
}

void launch_update_sy(bioen_hip_ctx* c, double* s, double* y) {
    hipLaunchKernelGGL(k_update_sy, dim3(vec_grid(c)), dim3(kBlock), 0, c->stream, c->x, c->xp, c->g, c->gp,
                       c->n, s, y, part(c, P_YS), part(c, P_YY));
}

void launch_recur(bioen_hip_ctx* c, const RecurArgs& a) {
    // The running dot product ping-pongs between two partial arrays: a step reads the
    // partials of its predecessor in its prologue while its own blocks already write new ones.
    const double* pin = part(c, c->rec_flip ? P_REC2 : P_REC);
    double* pout = nullptr;
    if (a.vdot) {
        if (a.out_slot == P_DGINIT) {
            pout = part(c, P_DGINIT);
        } else {
            c->rec_flip ^= 1;
            pout = part(c, c->rec_flip ? P_REC2 : P_REC);
        }
    }
    hipLaunchKernelGGL(k_recur, dim3(vec_grid(c)), dim3(kBlock), 0, c->stream, a.mode, a.hist, a.scale,
                       a.finalize_sy, c->d, c->gp, a.vaxpy, a.vdot, pin, part(c, P_YS), part(c, P_YY), vec_grid(c),
                       c->n, c->scal, pout);
}

void launch_generate(bioen_hip_ctx* c, const double* YTrue, const double* sig_sim, const double* sig_exp,
                     unsigned long long seed) {
    hipLaunchKernelGGL(k_generate, dim3(256 * 16), dim3(kBlock), 0, c->stream, c->Y, c->ld, c->m, c->n, c->mp, YTrue,
                       sig_sim, sig_exp, seed);
}

}  // namespace bioen
