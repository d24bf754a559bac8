// Host-side launch API of the gfx950 kernels (definitions: kernels.hip).
// Every function enqueues on ctx->stream and returns without synchronising.
#pragma once

#include "ctx.hpp"

namespace bioen {

// partial-reduction arrays inside ctx->part (each kMaxPartials doubles)
enum PartSlot : int {
    P_MAX = 0, P_SUM, P_PP, P_CHI, P_C, P_DG, P_GG, P_XX, P_DGINIT, P_REC, P_REC2, P_YS, P_YY, P_KL, P_TSUM,
    P_COUNT = 16
};

inline double* part(bioen_hip_ctx* c, int slot) { return c->part + (size_t)slot * kMaxPartials; }

int vec_grid(const bioen_hip_ctx* c);   // blocks used by every N-vector kernel of this context

// ---- matrix streaming kernels ------------------------------------------------
// forward:  partial[row][ctile] = sum_{j in tile} (Y[row][j] - [centred] ybar[row]) * v[j]
void launch_fwd_partial(bioen_hip_ctx* c, const double* v, bool centred = false);
// reduce the column tiles; mode 0: ybar,r,chi/c partials; mode 1: gm = rowsum - ybar * tsum
void launch_fwd_rows_residual(bioen_hip_ctx* c);
void launch_fwd_rows_forces_grad(bioen_hip_ctx* c);
// adjoint:  out[j] = sum_i (Y[i][j] - [centred] ybar[i]) * u[i]
void launch_adj(bioen_hip_ctx* c, const double* u, double* out, bool centred = false);

// ---- log-weights N-vector kernels ------------------------------------------------
// x = xp + stp * d ; block maxima of x -> P_MAX
void launch_trial(bioen_hip_ctx* c, double stp);
// e = exp(x - max) -> w (unnormalised), partial sums -> P_SUM, P_PP
void launch_logw_exp(bioen_hip_ctx* c);
// w /= S ; scal[S_LOGS], scal[S_P]
void launch_logw_norm(bioen_hip_ctx* c);
// scal[S_LOGS0] = log sum exp(fixed)     (once per problem)
void launch_logw_logs0(bioen_hip_ctx* c);
// scal[S_CHI], scal[S_C], scal[S_F] for log-weights
void launch_logw_scalars(bioen_hip_ctx* c, double theta);
// g = w (theta (x - G - P) + a - c) ; partials of g.d, g.g, x.x
void launch_logw_grad(bioen_hip_ctx* c, double theta);
// scal[S_DG], scal[S_GG], scal[S_XX], scal[S_DGINIT] from their partials
void launch_finish_eval(bioen_hip_ctx* c);

// ---- forces N-vector kernels ---------------------------------------------------
void launch_max(bioen_hip_ctx* c, const double* v);                 // block maxima of v -> P_MAX
void launch_forces_exp(bioen_hip_ctx* c, const double* xj);          // w = w0 exp(xj - max), P_SUM
void launch_forces_norm(bioen_hip_ctx* c);                           // w /= S ; KL partials
void launch_forces_t(bioen_hip_ctx* c, double theta);                // t, tsum partials
void launch_forces_scalars(bioen_hip_ctx* c, double theta);          // scal[S_F], S_KL, S_CHI

// ---- L-BFGS vector kernels (device-resident scalars) -------------------------------
// s = x - xp ; y = g - gp ; partials y.s -> P_YS, y.y -> P_YY
void launch_update_sy(bioen_hip_ctx* c, double* s, double* y);
// One fused step of the two-loop recursion on d (see kernels.hip: k_recur).
struct RecurArgs {
    int mode;            // 0 init (d = -gp), 1 first loop, 2 second loop
    int hist;            // history slot whose alpha / ys this step uses (modes 1,2)
    const double* vaxpy; // Y[hist] (mode 1) or S[hist] (mode 2)
    const double* vdot;  // vector dotted with the updated d for the NEXT step (may be null)
    int scale;           // 1: multiply the updated d by ys/yy (end of first loop / bound==0 never)
    int finalize_sy;     // 1 (mode 0 only): block 0 finalises y.s,y.y of slot `hist` from partials
    int out_slot;        // partial slot receiving vdot . d (P_REC or P_DGINIT)
};
void launch_recur(bioen_hip_ctx* c, const RecurArgs& a);

// ---- misc ---------------------------------------------------------------------------
void launch_generate(bioen_hip_ctx* c, const double* YTrue, const double* sig_sim, const double* sig_exp,
                     unsigned long long seed);

}  // namespace bioen
