// N-vector kernels of the log-weights evaluation and of the L-BFGS direction (gfx950).
#include "device_utils.hpp"

namespace bioen {

// (cache policy of the N-vector streams: device_utils.hpp, ld_hist / st_vec<POLICY>)
// (k_logw_grad reads the adjoint output a and e for the last time in the round: nontemporal too under the policy, -0.4 ...
// -0.7 % of the headline sweep; x, xp, g, gp of k_gram as well: no further gain)

// ------------------------------------------------------------------------------
// log-weights N-vector kernels (blockIdx.y = position a in the round's batch)
// ------------------------------------------------------------------------------
// x = xp + stp * d ; block maxima of x
template <bool POLICY>
__global__ __launch_bounds__(kBlock) void k_trial(Round r, int n, Xch xo) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    double* __restrict__ x = r.x[a];
    const double* __restrict__ xp = r.xp[a];
    const double* __restrict__ d = r.d[a];
    const double stp = r.stp[a];
    double mx = -DBL_MAX;
    const SegPos sp = seg_pos(xo.npl, xo.segcols, n);   // 16-byte pairs; vectors are zero-padded to an even length
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(xo.npl)) {
        const d2 dv = ld_vec(d + j);
        const d2 pv = ld_vec(xp + j);
        d2 v = {fma(stp, dv.x, pv.x), fma(stp, dv.y, pv.y)};
        st_vec<POLICY>(x + j, v);
        mx = fmax(mx, v.x);
        if (j + 1 < sp.jend) mx = fmax(mx, v.y);
    }
    mx = block_max(mx, sh);
    if (threadIdx.x == 0) xput<1>(xo, a, 0, mx);
}

__global__ __launch_bounds__(kBlock) void k_max(Round r, int n, Xch xo) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double* __restrict__ x = r.x[a];
    double mx = -DBL_MAX;
    const SegPos sp = seg_pos(xo.npl, xo.segcols, n);
    for (int j = sp.j0 + sp.b * kBlock + threadIdx.x; j < sp.jend; j += xo.npl * kBlock) mx = fmax(mx, x[j]);
    mx = block_max(mx, sh);
    if (threadIdx.x == 0) xput<1>(xo, a, 0, mx);
}

// _get_weights (c_bioen_kernels_logw.c:55-94) with a max shift, first half:
//   e_j = exp(x_j - m_r) ; partials of sum e and sum e (x - G)   (prior, :96-127)
// m_v is the maximum over the structures of the block's SEGMENT (its own block maxima need no exchange); it
// travels with the sums (third array, entry 0) and the consumers rescale by exp(m_v - max_v m_v).
template <bool POLICY>
__global__ __launch_bounds__(kBlock) void k_logw_exp(Round r, const double* __restrict__ G, int n, Xch xmx,
                                                     Xch xo) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y;
    const double* __restrict__ x = r.x[a];
    double* __restrict__ e = r.w[a];
    const double gmax = xmax_local<1>(xmx, a, 0);
    double s = 0.0, pp = 0.0;
    const SegPos sp = seg_pos(xo.npl, xo.segcols, n);
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(xo.npl)) {
        const d2 xv = ld_vec(x + j);
        const d2 Gv = ld_vec(G + j);
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

// second half: w = e / S ; scal[S_LOGS] = max + log S ; scal[S_P] = sum e (x-G) / S
__global__ __launch_bounds__(kBlock) void k_logw_norm(Round r, int n, Xch xe) {
    const int a = blockIdx.y;
    double* __restrict__ w = r.w[a];
    const SegPos sp = seg_pos(xe.npl, xe.segcols, n);
    // global shift M = max_v m_v ; S = sum_v e^{m_v - M} S_v   (segment order)
    double gmax = -DBL_MAX;
    for (int sg = 0; sg < xe.world; ++sg) gmax = fmax(gmax, xseg_ptr<3>(xe, sg, a, 2)[0]);
    double S = 0.0, PP = 0.0;
    for (int sg = 0; sg < xe.world; ++sg) {
        const double fr = exp(xseg_ptr<3>(xe, sg, a, 2)[0] - gmax);
        S = fma(fr, xsum_seg<3>(xe, sg, a, 0), S);
        PP = fma(fr, xsum_seg<3>(xe, sg, a, 1), PP);
    }
    const double mown = xseg_ptr<3>(xe, xe.rank + sp.v, a, 2)[0];
    const double inv = exp(mown - gmax) / S;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        r.scal[a][S_LOGS] = gmax + log(S);
        r.scal[a][S_P] = PP * (1.0 / S);
    }
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(xe.npl)) {
        d2 v = *reinterpret_cast<const d2*>(w + j);
        v.x *= inv;
        v.y = (j + 1 < sp.jend) ? v.y * inv : 0.0;      // the padding stays zero whatever the factor (see k_scale_w)
        *reinterpret_cast<d2*>(w + j) = v;
    }
}

// log s0 = log sum exp(G): constant per problem, computed once (the reference recomputes it
// at every evaluation, c_bioen_kernels_logw.c:122).  Two stages (until r03 one block walked the whole vector: 1.2 ms
// at N = 1e6, 0.6 ms per run of a K = 1 series at N = 5e5): every block of the N-vector grid leaves {max, sum of
// exp(G - max)} of its share, one block merges them in block order -- a fixed order, and for a uniform prior
// (G = 0: every term is exactly 1, the sums are integers) the same bits as any other order.
__global__ __launch_bounds__(kBlock) void k_logsumexp_part(const double* __restrict__ G, int n, SegMap sm, double* __restrict__ part) {
    __shared__ double sh[kWaves];
    // block b of a segment takes the b-th run of ceil(segcols / npl) of its columns: part[seg * npl + b] on every GPU count
    const SegPos sp = seg_pos(sm.npl, sm.segcols, n);
    const int per = (sm.segcols + sm.npl - 1) / sm.npl;
    const int j0 = sp.j0 + sp.b * per, j1 = min(sp.jend, j0 + per);
    double mx = -DBL_MAX;
    for (int j = j0 + threadIdx.x; j < j1; j += kBlock) mx = fmax(mx, G[j]);
    mx = block_max(mx, sh);
    double s = 0.0;
    for (int j = j0 + threadIdx.x; j < j1; j += kBlock) s += exp(G[j] - mx);
    s = block_sum(s, sh);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = mx;
        part[2 * blockIdx.x + 1] = s;          // 0 for a block without elements
    }
}

__global__ __launch_bounds__(kBlock) void k_logsumexp_merge(const double* __restrict__ part, int nblk, Round r) {
    __shared__ double sh[kWaves];
    double mx = -DBL_MAX;
    for (int b = threadIdx.x; b < nblk; b += kBlock) mx = fmax(mx, part[2 * b]);
    mx = block_max(mx, sh);
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += kBlock) {
        const double sb = part[2 * b + 1];
        if (sb > 0.0) s += sb * exp(part[2 * b] - mx);
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) {
        const double v = mx + log(s);
        for (int a = 0; a < r.n; ++a) r.scal[a][S_LOGS0] = v;
    }
}

// gradient epilogue (c_bioen_kernels_logw.c:207-218):
//   g_k = w_k [ theta (x_k - G_k - P) + a_k ],  a_k = sum_i r_i (yTilde_ik - ybar_i)  (centred adjoint)
// plus the three dot products the line search / convergence test needs.
template <bool POLICY>
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
    const SegPos sp = seg_pos(xo.npl, xo.segcols, n);
    const double inv = r.scal[a][S_INV + sp.v];      // w = e * inv (the segment's own shift)
    double dg = 0.0, gg = 0.0, xx = 0.0;
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(xo.npl)) {
        const d2 xv = ld_vec(x + j);
        d2 wv = ld_hist<POLICY>(w + j);                        // pad: e = 0  =>  g = 0
        wv.x *= inv;
        wv.y *= inv;
        const d2 Gv = ld_vec(G + j);
        const d2 aa = ld_hist<POLICY>(av + j);
        const d2 dv = ld_vec(d + j);
        d2 gv;
        gv.x = wv.x * (theta * ((xv.x - Gv.x) - P) + aa.x);
        gv.y = wv.y * (theta * ((xv.y - Gv.y) - P) + aa.y);
        st_vec<POLICY>(g + j, gv);
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

// live != NULL: the problem's whole scalar slot also goes to the host-mapped page (ctx.hpp: live), then -- after a
// system-scope fence -- the round number into the problem's flag, which the host is spinning on.
__global__ __launch_bounds__(kBlock) void k_finish_eval(Round r, Xch xg, double* __restrict__ live,
                                                        unsigned long long round) {
    __shared__ double sh[3 * kShRed];
    const int a = blockIdx.y;
    double sums[3];
    xsum_multi<3, 3>(xg, a, 0, sh, sums);
    const double dg = sums[0], gg = sums[1], xx = sums[2];
    double* sc = r.scal[a];
    if (threadIdx.x == 0) {
        sc[S_DG] = dg;
        sc[S_GG] = gg;
        sc[S_XX] = xx;
    }
    static_assert(kScalStride <= 64, "the slot is published by ONE wave: its fence covers the stores of all its lanes");
    if (live && threadIdx.x < kScalStride) {      // one wave: the fence below is a wave-wide wait for ALL its lanes' stores
                                                  // (s_waitcnt vmcnt(0) + write-back), after which lane 0 raises the flag
        const int t = threadIdx.x;
        double v = sc[t];                         // the other entries were written by earlier kernels
        if (t == S_DG) v = dg;
        if (t == S_GG) v = gg;
        if (t == S_XX) v = xx;
        live[(size_t)a * kScalStride + t] = v;
        __threadfence_system();
        if (t == 0)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(live + (size_t)kMaxBatch * kScalStride) + a, round,
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// gp . d of the freshly built direction -> scal[S_DGINIT] (the next line search's initial slope)
__global__ __launch_bounds__(kBlock) void k_store_dginit(MVec8 scal, Xch xd) {
    __shared__ double sh[kShRed];
    const int a = blockIdx.y;
    const double di = xsum<1>(xd, a, 0, sh);
    if (threadIdx.x == 0) scal.p[a][S_DGINIT] = di;
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
    const SegPos sp = seg_pos(xo.npl, xo.segcols, n);
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(xo.npl)) {
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
    __shared__ double sh[kShRed];
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
    const SegPos sp = seg_pos(xin.npl, xin.segcols, n);
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(xin.npl)) {
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
template <bool POLICY>
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
    const SegPos sp = seg_pos(xo.npl, xo.segcols, n);
    for (int j = seg_first(sp); j < sp.jend; j += seg_step(xo.npl)) {
        const d2 a0 = ld_vec(xn + j), a1 = ld_vec(xo_ + j);
        const d2 gv = ld_vec(gn + j), g1 = ld_vec(go + j);
        const d2 sv = {a0.x - a1.x, a0.y - a1.y};
        const d2 yv = {gv.x - g1.x, gv.y - g1.y};
        d2 B[kBasis];
#pragma unroll
        for (int k = 0; k < kHistory; ++k) {
            B[k] = (k == e) ? sv : ld_hist<POLICY>(q.S[a][k] + j);
            B[kHistory + k] = (k == e) ? yv : ld_hist<POLICY>(q.Y[a][k] + j);
        }
        B[2 * kHistory] = gv;
        st_vec<POLICY>(q.S[a][e] + j, sv);
        st_vec<POLICY>(q.Y[a][e] + j, yv);
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

// One wave per (sum, problem, local segment) totals the segment's block partials into the compact X_GRAMR stage
// (39 doubles per problem and segment): what sharded contexts exchange, and what every context -- one GPU or eight --
// finishes the sums from.
__global__ void k_gram_rank_reduce(Xch xi, Xch xo) {
    const int c = blockIdx.x, a = blockIdx.y, v = threadIdx.x >> 6;
    const double t = xsum_seg<kGramDots>(xi, xi.rank + v, a, c);
    if ((threadIdx.x & 63) == 0) xo.base[(size_t)(xo.rank + v) * xo.payload + (size_t)a * kGramDots + c] = t;
}

// Per problem (one block): finish the 39 sums from the segments' totals (segment order), update the Gram matrix, run
// the two-loop recursion (lbfgs.c:571-598) on coefficients, leave them in gram[169..181] and gp.d in scal[S_DGINIT].
// The 13x13 matrix lives in LDS while one thread walks the two loops.
__global__ __launch_bounds__(kBlock) void k_gram_solve(GramArgs q, Xch xi) {
    __shared__ double dots[kGramDots];
    __shared__ double Gs[kBasis * kBasis];
    __shared__ double alpha[kHistory];
    const int a = blockIdx.y;
    double* G = q.gram[a];
    for (int i = threadIdx.x; i < kBasis * kBasis; i += kBlock) Gs[i] = G[i];
    for (int i = threadIdx.x; i < kGramDots; i += kBlock)
        dots[i] = seg_order_sum(xi.base + (size_t)a * kGramDots + i, (size_t)xi.payload, xi.world);
    __syncthreads();
    if (threadIdx.x != 0) return;
    gram_solve_thread0(G, Gs, dots, alpha, q.end[a], q.bound[a], q.scal[a]);
}

// d = sum_c coef_c B_c
template <bool POLICY>
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
        const d2 gv = ld_vec(gn + j);
        d2 dv = {cf[2 * kHistory] * gv.x, cf[2 * kHistory] * gv.y};
#pragma unroll
        for (int k = 0; k < kHistory; ++k) {
            if (cf[k] != 0.0) {               // unused history slots may hold another problem's leftovers
                const d2 v = ld_hist<POLICY>(q.S[a][k] + j);
                dv.x = fma(cf[k], v.x, dv.x);
                dv.y = fma(cf[k], v.y, dv.y);
            }
            if (cf[kHistory + k] != 0.0) {
                const d2 v = ld_hist<POLICY>(q.Y[a][k] + j);
                dv.x = fma(cf[kHistory + k], v.x, dv.x);
                dv.y = fma(cf[kHistory + k], v.y, dv.y);
            }
        }
        st_vec<POLICY>(d + j, dv);
    }
}


// ---- log-weights vector kernels -----------------------------------------------------------
void launch_trial(bioen_hip_ctx* c, const Round& r) {
    if (c->nvec_nt) hipLaunchKernelGGL(k_trial<true>, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->n,
                                       make_xch(c, X_MAX, r.n * vec_grid(c)));
    else hipLaunchKernelGGL(k_trial<false>, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->n,
                            make_xch(c, X_MAX, r.n * vec_grid(c)));
}

void launch_max(bioen_hip_ctx* c, const Round& r) {
    hipLaunchKernelGGL(k_max, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->n,
                       make_xch(c, X_MAX, r.n * vec_grid(c)));
}

void launch_logw_exp(bioen_hip_ctx* c, const Round& r) {
    if (c->nvec_nt) hipLaunchKernelGGL(k_logw_exp<true>, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                                       make_xch(c, X_MAX, r.n * vec_grid(c)), make_xch(c, X_EXP, 3 * r.n * vec_grid(c)));
    else hipLaunchKernelGGL(k_logw_exp<false>, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                            make_xch(c, X_MAX, r.n * vec_grid(c)), make_xch(c, X_EXP, 3 * r.n * vec_grid(c)));
}

void launch_logw_norm(bioen_hip_ctx* c, const Round& r) {
    hipLaunchKernelGGL(k_logw_norm, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->n,
                       make_xch(c, X_EXP, 3 * r.n * vec_grid(c)));
}

// the per-block {max, sum} pairs go into this rank's segment of the X_GRAD stage (3 K values per block: room for 2), idle
// before a run's first evaluation; sharded contexts exchange the segments between the two launches (api.hip: enqueue_logs0)
void launch_logw_logs0_part(bioen_hip_ctx* c) {
    const int g = vec_grid(c);
    double* part = c->xbuf[X_GRAD] + (size_t)c->seg0 * 2 * g;
    hipLaunchKernelGGL(k_logsumexp_part, dim3(vec_blocks(c)), dim3(kBlock), 0, c->stream, c->fixed, c->n, seg_map(c), part);
}
void launch_logw_logs0_merge(bioen_hip_ctx* c, const Round& r) {
    hipLaunchKernelGGL(k_logsumexp_merge, dim3(1), dim3(kBlock), 0, c->stream, c->xbuf[X_GRAD], vec_grid(c) * c->nseg, r);
}

void launch_logw_grad(bioen_hip_ctx* c, const Round& r) {
    if (c->nvec_nt) hipLaunchKernelGGL(k_logw_grad<true>, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                                       make_xch(c, X_GRAD, 3 * r.n * vec_grid(c)));
    else hipLaunchKernelGGL(k_logw_grad<false>, dim3(vec_blocks(c), r.n), dim3(kBlock), 0, c->stream, r, c->fixed, c->n,
                            make_xch(c, X_GRAD, 3 * r.n * vec_grid(c)));
}

void launch_finish_eval(bioen_hip_ctx* c, const Round& r) {
    hipLaunchKernelGGL(k_finish_eval, dim3(1, r.n), dim3(kBlock), 0, c->stream, r,
                       make_xch(c, X_GRAD, 3 * r.n * vec_grid(c)), c->live_round ? c->live : nullptr, c->live_round);
    c->live_round = 0;
}

void launch_store_dginit(bioen_hip_ctx* c, int k, const MVec8& scal) {
    hipLaunchKernelGGL(k_store_dginit, dim3(1, k), dim3(kBlock), 0, c->stream, scal,
                       make_xch(c, X_DGI, k * vec_grid(c)));
}


// ---- L-BFGS vector kernels ----------------------------------------------------------------------
void launch_update_sy(bioen_hip_ctx* c, const PairArgs& a, int kdir) {
    hipLaunchKernelGGL(k_update_sy, dim3(vec_blocks(c), a.n), dim3(kBlock), 0, c->stream, a, c->n,
                       make_xch(c, X_SY, 2 * kdir * vec_grid(c)));
}

void launch_recur(bioen_hip_ctx* c, const RecurArgs& a, int step) {
    const int k = a.n, g = vec_grid(c);
    // step s reads the running dot its predecessor left in REC[(s-1)&1] and writes REC[s&1]
    hipLaunchKernelGGL(k_recur, dim3(vec_blocks(c), k), dim3(kBlock), 0, c->stream, a, c->n,
                       make_xch(c, ((step - 1) & 1) ? X_REC1 : X_REC0, k * g), make_xch(c, X_SY, 2 * k * g),
                       make_xch(c, (step & 1) ? X_REC1 : X_REC0, k * g), make_xch(c, X_DGI, k * g));
}

void launch_gram(bioen_hip_ctx* c, const GramArgs& a) {
    if (c->nvec_nt) hipLaunchKernelGGL(k_gram<true>, dim3(vec_blocks(c), a.n), dim3(kBlock), 0, c->stream, a, c->n,
                                       make_xch(c, X_GRAM, kGramDots * a.n * vec_grid(c)));
    else hipLaunchKernelGGL(k_gram<false>, dim3(vec_blocks(c), a.n), dim3(kBlock), 0, c->stream, a, c->n,
                            make_xch(c, X_GRAM, kGramDots * a.n * vec_grid(c)));
}

static Xch gram_rank_view(const bioen_hip_ctx* c, int k) {      // X_GRAMR: one value per (segment, problem, sum)
    Xch x = make_xch(c, X_GRAMR, kGramDots * k);
    x.npl = 1;
    return x;
}

void launch_gram_rank_reduce(bioen_hip_ctx* c, int k) {
    hipLaunchKernelGGL(k_gram_rank_reduce, dim3(kGramDots, k), dim3(64 * c->vr), 0, c->stream,
                       make_xch(c, X_GRAM, kGramDots * k * vec_grid(c)), gram_rank_view(c, k));
}

void launch_gram_solve(bioen_hip_ctx* c, const GramArgs& a) {      // the segments' totals (after the X_GRAMR exchange)
    hipLaunchKernelGGL(k_gram_solve, dim3(1, a.n), dim3(kBlock), 0, c->stream, a, gram_rank_view(c, a.n));
}

void launch_combine(bioen_hip_ctx* c, const GramArgs& a) {
    if (c->nvec_nt) hipLaunchKernelGGL(k_combine<true>, dim3(vec_blocks(c), a.n), dim3(kBlock), 0, c->stream, a, c->n);
    else hipLaunchKernelGGL(k_combine<false>, dim3(vec_blocks(c), a.n), dim3(kBlock), 0, c->stream, a, c->n);
}


}  // namespace bioen
