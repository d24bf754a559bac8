// Host-side launch API of the gfx950 kernels (definitions: kernels_strip.hip -- the matrix passes for M <= 1024 and, over
// row panels, beyond --, kernels_matrix.hip, kernels_logw.hip, kernels_forces.hip, kernels_devls.hip, kernels_misc.hip,
// kernels_p2p.hip -- the peer-to-peer stage exchange; shared device helpers: device_utils.hpp).
// Every function enqueues on ctx->stream and returns without synchronising.
//
// All kernels are batched: a launch serves the K (<= kMaxBatch) problems listed in a
// `Round`, in "batch order" a = 0..K-1.  Each problem performs exactly the arithmetic,
// in exactly the order, it would perform alone (K = 1), so a batched run is bitwise
// identical to K separate runs.
#pragma once

#include "ctx.hpp"
#include "lbfgs_state.hpp"

namespace bioen {

// Per-round, per-batch-position device pointers and scalars (passed by value).
struct Round {
    int n;                        // K: problems in this round
    double* x[kMaxBatch];         // trial point
    double* xp[kMaxBatch];        // accepted point
    double* g[kMaxBatch];         // trial gradient
    double* gp[kMaxBatch];        // accepted gradient
    double* d[kMaxBatch];         // search direction
    double* w[kMaxBatch];         // weights
    double* a[kMaxBatch];         // adjoint output
    double* scal[kMaxBatch];
    double* part[kMaxBatch];
    double stp[kMaxBatch];
    double theta[kMaxBatch];
};

struct Vec8 {
    const double* p[kMaxBatch];
};
struct MVec8 {
    double* p[kMaxBatch];
};

Xch make_xch(const bioen_hip_ctx* c, int stage, int payload);   // stage view for a launch

// peer-to-peer stage exchange (kernels_p2p.hip): one kernel per all-gather, stores into the peers' mailboxes + flags
size_t p2p_mailbox_doubles(int world, size_t cap);
void launch_p2p_exchange(bioen_hip_ctx* c, int stage, size_t payload);
void launch_xch_mirror(bioen_hip_ctx* c, int stage, size_t payload);        // measurement aid: this rank's part over every other rank's
void launch_xch_fill(bioen_hip_ctx* c, int stage, int payload, int rep);    // self-test of a transport: pattern in, ...
void launch_xch_check(bioen_hip_ctx* c, int stage, int payload, int rep, unsigned long long* bad);   // ... every segment checked

int vec_grid(const bioen_hip_ctx* c);   // blocks PER SEGMENT of every N-vector kernel of this context (a function of segcols and
                                        // nseg alone: the same on 1, 2, 4 and 8 GPUs)
int vec_blocks(const bioen_hip_ctx* c); // ... and their grid: vec_grid x the segments this context holds
struct SegMap {                         // what an N-vector kernel without a stage view needs to walk its segment
    int npl, segcols;
};
SegMap seg_map(const bioen_hip_ctx* c);
void rank_columns(const bioen_hip_ctx* c, int rank, long long* col0, long long* n_local);   // columns of `rank` in this context's decomposition
int rows_grid(const bioen_hip_ctx* c);

// ---- matrix streaming kernels ------------------------------------------------
// forward: fwd_partial[(row*K + a)*ctiles + tile] = sum_{j in tile} (Y[row][j] - [centred] ybar_c[row*K+a]) v_a[j]
void launch_fwd_partial(bioen_hip_ctx* c, int K, const Vec8& v, bool centred = false);
// reduce the column tiles -> this rank's share of ybar in its X_YBAR segment   [exchange X_YBAR]
void launch_fwd_rows_local(bioen_hip_ctx* c, int K, bool logw, int ctiles = 0, bool tposed = false);   // logw: + {sum e, sum e (x-G), m_r} per problem; ctiles: partials per (row, problem), default the streaming kernel's
int ybar_payload(const bioen_hip_ctx* c, int K, bool logw);        // doubles per rank in the X_YBAR stage
void launch_scale_w(bioen_hip_ctx* c, const Round& r);              // w = e * scal[S_INV] (when a result is handed out)
// add the ranks' shares -> ybar_c, r_c (compact), chi^2 / ybar.r partials per problem
// center != NULL: the shares are sums over the CENTRED copy (kernels_strip.hip): ybar_raw = share + center;
// ybar_c keeps ybar_raw (store_raw: what k_adj's centring and the callers expect) or the centred share (the
// forces strip pass 2, whose gradient correction is written in terms of it)
struct DevGate {                      // device-resident engine: skip the problems the device has finished (kernels.hpp: DevSlot)
    const struct DevSlot* tab;        // NULL: no gating
    int slot[kMaxBatch];              // position -> slot of its OWNER
    int cand[kMaxBatch];              // != 0: a speculative trial (needs a line search in progress)
};
void launch_rows_combine(bioen_hip_ctx* c, const Round& r, bool logw,   // logw: also chi^2, c, f -> scal
                         const double* center = nullptr, bool store_raw = true, const DevGate* gate = nullptr);
int combine_grid(const bioen_hip_ctx* c);
// forces gradient: gm_c[row*K + a] = reduced centred sums
// tsum = true (strip passes on the centred copy): gm = sum of partials - ybar_c * T, T = sum of the blocks' P_KL shares
void launch_fwd_rows_forces_grad(bioen_hip_ctx* c, int K, int ctiles);
// forces evaluation in TWO matrix passes over LDS-resident column strips (M <= 1024):
//   xy: x = yTilde^T f (-> slot a), online softmax per block, raw ybar partials; block merge; ybar -> X_YBAR
//   bt: b = yTilde^T r, t, and the centred yTilde . t together; partials -> fwd_partial[.. * nblk + block]
constexpr int kFusedBlocks = 1024;
// Canonical partial sets of the row-sum strip passes (the log-weights forward pass, both forces passes).  Per SEGMENT
// (ctx.hpp): its sps strips are dealt to gs groups (group g: strips g, g + gs, ...), a group's strips are cut into nch
// chunks of tc consecutive ones, and a SET is the matrix-core accumulation chain over one chunk, from zero, in strip
// order -- a function of (segcols, M) alone.  The log-weights sums are then formed as
//     segment share = fixed tree over its groups of [ the group's chunks added in turn from +0.0 ],
// the forces sums as the online-softmax merge over the segment's sets in set order.  How many sets one block runs is
// a matter of the GPU count, not of the result: on 8 GPUs a block runs one chunk (gs x nch = a full grid per segment);
// a context that holds all 8 segments lets the forward kernel run whole groups and add up their chunks in registers
// (fold: one set per group reaches memory, and 256 blocks sweep the matrix exactly as before r05).
struct StripSets {
    int sps, gs, tc, nch;
    int fold;           // 1: one physical slot per group, chunks folded in the kernel; 0: one per chunk
    int slots;          // physical slots per segment = gs * (fold ? 1 : nch)
    int sets;           // sets per segment that reach memory = slots
};
StripSets strip_sets(const bioen_hip_ctx* c);      // log-weights forward pass
StripSets forces_sets(const bioen_hip_ctx* c);     // forces passes (never folded); gs = 0: the strip passes do not apply
constexpr int kPartStride = 8192;                  // entries per local partial array (ctx.hpp: PartSlot): sets of a context
void launch_read_probe(bioen_hip_ctx* c, const double* p, size_t doubles, double* out);   // bench: measured read ceiling
// forces engine: the round's gradients and scalars straight into the host's page, then the round number into its flag
void launch_forces_publish(bioen_hip_ctx* c, int ngrad, unsigned long long round);
bool strip_panels(const bioen_hip_ctx* c);             // M > 1024: the strip kernels run over row panels
int forces_fused_blocks(const bioen_hip_ctx* c);       // sets per segment of the forces strip passes; 0 when the context does not qualify
void launch_forces_xy(bioen_hip_ctx* c, const struct ForcesRound& fr, int nblk);
void launch_forces_bt(bioen_hip_ctx* c, const struct ForcesRound& fr, int nblk);
void launch_forces_w_from_x(bioen_hip_ctx* c, const struct ForcesRound& fr);   // w = w0 exp(x - S_LOGS)
// every local segment's share of the forces gradient -> its part of X_YBAR.  tposed: the two-pass strip kernels' sets (seg_sets per
// segment); else the log-weights forward kernel's sets (row panels, M > 1024)
void launch_fwd_rows_forces_grad_share(bioen_hip_ctx* c, int K, int seg_sets, const struct ForcesRound* tsum = nullptr, bool tposed = false);
// builds ctx->Ys on first use; method 0 / 1: the log-weights / the forces passes come next and want the copy in their
// layout (kernels_strip.hip: strip_phys; an existing copy is moved, best effort), -1: whatever is there
int ensure_strip_copy(bioen_hip_ctx* c, int method = -1);
int set_storage_format(bioen_hip_ctx* c, int fmt);     // reduced-byte storage experiment of the log-weights passes (0 = FP64)
int fwd_strip_blocks(const bioen_hip_ctx* c);          // > 0: the log-weights forward pass runs on the strip copy
void launch_fwd_strip(bioen_hip_ctx* c, int K, const Vec8& v, int nblk, bool plain = false);
int ensure_rowmajor(bioen_hip_ctx* c);                 // the row-major matrix back from the strip copy (it is freed once that exists)
int gather_block(bioen_hip_ctx* c, int row0, int rows, size_t col0, int cols, double* out);   // -> device out[rows][cols]
int ensure_strip_copy_colsum(bioen_hip_ctx* c);        // builds ctx->Ys1 (column-sum operand order) on first use
void launch_adj_strip(bioen_hip_ctx* c, int K, const double* u_c, const MVec8& out, const MVec8& scal, int nblk, bool plain = false);
void launch_forces_blockmerge(bioen_hip_ctx* c, const struct ForcesRound& fr, int seg_sets, bool tposed = true);
void launch_forces_grad_sum_ranks(bioen_hip_ctx* c, int K);                     //          shares -> gm
// adjoint: out_a[j] = sum_i (Y[i][j] - [centred] ybar_c[i*K+a]) u_c[i*K+a]
void launch_adj(bioen_hip_ctx* c, int K, const double* u_c, const MVec8& out, bool centred = false);

// ---- log-weights N-vector kernels (blockIdx.y = batch position) ------------------------
void launch_trial(bioen_hip_ctx* c, const Round& r);        // x = xp + stp d ; block maxima
void launch_max(bioen_hip_ctx* c, const Round& r);          // block maxima of x only
void launch_logw_exp(bioen_hip_ctx* c, const Round& r);     // e = exp(x - max) -> w ; partial sums
void launch_logw_norm(bioen_hip_ctx* c, const Round& r);    // w /= S ; log s, P
void launch_logw_logs0_part(bioen_hip_ctx* c);               // {max, sum exp(fixed - max)} per block -> X_GRAD segment of this rank
void launch_logw_logs0_merge(bioen_hip_ctx* c, const Round& r);   // scal[S_LOGS0] = log sum exp(fixed), all ranks' pairs merged
void launch_logw_grad(bioen_hip_ctx* c, const Round& r);    // gradient epilogue + g.d, g.g, x.x
void launch_finish_eval(bioen_hip_ctx* c, const Round& r);  // scal[S_DG], S_GG, S_XX
void launch_store_dginit(bioen_hip_ctx* c, int k, const MVec8& scal);   // scal[S_DGINIT] <- X_DGI

// ---- forces N-vector kernels (blockIdx.y = batch position) ---------------------------------
struct ForcesRound {
    int n;
    double* a[kMaxBatch];      // xj = yTilde^T f, later b = yTilde^T r
    double* w[kMaxBatch];
    double* t[kMaxBatch];
    double* scal[kMaxBatch];
    double* part[kMaxBatch];
    double theta[kMaxBatch];
};
void launch_forces_max(bioen_hip_ctx* c, const ForcesRound& r);
void launch_forces_exp(bioen_hip_ctx* c, const ForcesRound& r);
void launch_forces_norm(bioen_hip_ctx* c, const ForcesRound& r);
void launch_forces_t(bioen_hip_ctx* c, const ForcesRound& r);
void launch_forces_scalars(bioen_hip_ctx* c, const ForcesRound& r);
// canonical-segment forms of the row-panel path (M > 1024; any number of ranks)
void launch_forces_seg_exp(bioen_hip_ctx* c, const ForcesRound& r);                 // e = w0 exp(x - m_v) -> w ; X_EXP shares (needs launch_max on x)
void launch_forces_seg_t(bioen_hip_ctx* c, const ForcesRound& r, int seg_sets);     // t ; T_v -> share 0 of the segment's seg_sets P_KL shares

// ---- assembly of yTilde = sim / sigma on the device ----------------------------------------
void launch_rows_div(bioen_hip_ctx* c, const double* sigma);          // Y[i][:] /= sigma_i (device pointer)
void launch_transpose_div(bioen_hip_ctx* c, const double* src, int ncols, size_t col0, const double* sigma);

// ---- level-1 algebra on resident N-vectors (GSL-style minimizers, multimin.hpp) ----------
struct VDotArgs {
    int k;                 // number of (x, y) pairs, <= 4
    int mode;              // 0: inner products ; 1: out[0] = #(x0 != y0), out[1] = max |x1|
    const double* x[4];
    const double* y[4];
};
void launch_vaxpy(bioen_hip_ctx* c, double a, const double* x, double* y);                  // y += a x
void launch_vscal(bioen_hip_ctx* c, double a, double* x);
void launch_vstep(bioen_hip_ctx* c, const double* x, const double* p, double coef, double* x1, double* dx);
void launch_vdots_part(bioen_hip_ctx* c, const VDotArgs& q);                                // block partials -> X_GRAD stage   [exchange]
void launch_vdots_finish(bioen_hip_ctx* c, const VDotArgs& q, double* out);                 // out: device, k doubles

// ---- L-BFGS vector kernels (device-resident scalars) -------------------------------
struct PairArgs {      // s = x - xp ; y = g - gp for the accepting problems
    int n;
    const double* x[kMaxBatch];
    const double* xp[kMaxBatch];
    const double* g[kMaxBatch];
    const double* gp[kMaxBatch];
    double* s[kMaxBatch];
    double* y[kMaxBatch];
    int xpos[kMaxBatch];   // position of the problem in the direction batch (X_SY index)
};
void launch_update_sy(bioen_hip_ctx* c, const PairArgs& a, int kdir);

// One fused step of the two-loop recursion per problem (see kernels_logw.hip: k_recur).
struct RecurArgs {
    int n;
    int mode[kMaxBatch];          // -1 idle, 0 init (d = -gp), 1 first loop, 2 second loop
    int hist[kMaxBatch];          // history slot whose alpha / ys this step uses
    int scale[kMaxBatch];         // multiply the updated d by ys/yy (last step of the first loop)
    int finalize_sy[kMaxBatch];   // mode 0: block 0 finalises y.s, y.y of slot `hist` from partials
    double* d[kMaxBatch];
    const double* gp[kMaxBatch];
    const double* vaxpy[kMaxBatch];
    const double* vdot[kMaxBatch];
    int to_dginit[kMaxBatch];     // this step's dot is gp . d (goes to X_DGI instead of X_REC)
    double* scal[kMaxBatch];
};
void launch_recur(bioen_hip_ctx* c, const RecurArgs& a, int step);

// Direction from inner products (kernels_logw.hip: k_gram / k_gram_solve / k_combine): the same
// two-loop recursion carried out on coefficients over the basis {S_0..5, Y_0..5, g}.  One sweep
// commits the new (s, y) pair and produces every inner product the recursion needs, so a direction
// costs 3 launches and ONE stage exchange instead of 14 + 14.
struct GramArgs {
    int n;
    const double* xnew[kMaxBatch];   // accepted (former trial) point
    const double* xold[kMaxBatch];
    const double* gnew[kMaxBatch];   // gradient at the accepted point
    const double* gold[kMaxBatch];
    double* S[kMaxBatch][kHistory];
    double* Y[kMaxBatch][kHistory];
    double* d[kMaxBatch];
    double* gram[kMaxBatch];
    double* scal[kMaxBatch];
    int end[kMaxBatch];              // history slot of the new pair
    int bound[kMaxBatch];            // pairs in use (including the new one)
};
void launch_gram(bioen_hip_ctx* c, const GramArgs& a);          // [exchange X_GRAM]
void launch_gram_rank_reduce(bioen_hip_ctx* c, int k);   // sharded: X_GRAM block partials -> X_GRAMR rank totals
void launch_gram_solve(bioen_hip_ctx* c, const GramArgs& a);
void launch_combine(bioen_hip_ctx* c, const GramArgs& a);

// ---- device-resident line-search decisions (kernels_devls.hip, engine_devls.inl) ---------------------------------
// The log-weights batch engine keeps each problem's L-BFGS state machine, the ROLES of its vectors (which buffer is
// the trial point, which the accepted one, which history buffer the next pair goes to) and its status in HBM; a
// one-block kernel per problem takes the line-search decision a round ends with.  The host composes rounds, enqueues
// them ahead of the results and watches a host-mapped page for finished problems.
constexpr int kMaxPast = 64;          // lbfgs `past` entries kept on the device (yaml default 10); beyond: host engine
enum DevStatus : int { DS_IDLE = 0, DS_INITIAL = 1, DS_RUNNING = 2, DS_DONE = 3 };

struct DevSlot {                      // one per problem slot, device memory (ctx->dev_tab)
    double *x, *xp, *g, *gp;          // trial point / accepted point, their gradients
    double* S[kHistory];              // history ring
    double* Y[kHistory];
    double *Ssp, *Ysp;                // spare pair: (s, y) of the PENDING trial, written by the gradient sweep before
                                      //   the decision is known; an accepted trial swaps it into the ring
    LbfgsState m;
    double pf[kMaxPast];
    int status;                       // DevStatus
    int combine;                      // 1: the next step kernel forms d from the Gram coefficients first (a step was accepted);
                                      // 2: ... whose Gram products are still to be formed (an adopted shadow: k_dev_late_*)
    int code, keep_trial;             // DS_DONE: liblbfgs status / result is the trial point
    int was_initial;                  // DS_DONE: ... reached at the evaluation of the start point
    int late_end, late_bound;         // combine == 2: history slot / pairs in use of the accepted step
    int rej_dec, rej_inc;             // rejected trials so far that asked for a shorter / a longer step next
    int pad;
};

constexpr int kLiveRec = 56;          // doubles per published record
struct DevRecord {                    // what a decision publishes per position and round (host-mapped page)
    double scal[kScalStride];         // the problem's scalar slot (chi^2, KL pieces, ...) as it stands
    double* x; double* xp; double* g; double* gp;   // the roles after the decision
    double* w;                        // buffer holding e of the evaluation the problem stands on (own, or an adopted shadow's)
    double fx, stp;
    int status, code, keep_trial, was_initial;
    int iterations, evaluations;
    int adopted, evalpos;             // the evaluation the problem stands on was a shadow's / its position in the round
    int rej_dec, rej_inc;
};
static_assert(sizeof(DevRecord) <= kLiveRec * sizeof(double), "live record");

struct DevRound {                     // a round as the kernels see it (by value)
    int n;                            // positions: owners first (0 .. nown - 1), then shadows
    int nown;
    int slot[kMaxBatch];              // position -> problem slot whose buffers / scalars the evaluation uses
    int owner[kMaxBatch];             // position -> position of its owner (itself for owners)
    int cand[kMaxBatch];              // 0: the owner's own trial; 1: stp * 0.5; 2: stp * 2.1 (speculative trials)
    int shadow[kMaxBatch][2];         // owners: positions of their shadows with cand 1 / 2 (-1: none)
    double theta[kMaxBatch];
    double* d[kMaxBatch];             // fixed per-slot buffers of the position's OWNER: direction
    double* gram[kMaxBatch];          //   ... Gram matrix + coefficients (owner), finished sums (position)
    double* w[kMaxBatch];             // fixed per-slot buffers of the position: e = exp(x - m)
    double* a[kMaxBatch];             //   adjoint output
    double* scal[kMaxBatch];          //   scalar slot
    DevSlot* tab;                     // ctx->dev_tab
    int sgram;                        // shadows sweep their own (s, y) pair and its 39 products in the main pass (sharded
                                      // contexts: no late Gram pass, no third all-gather); they own a spare pair then
};

struct DevStart {                     // (re)start of problems: by value to k_dev_start
    int n;
    int slot[kMaxBatch];
    const double* g0[kMaxBatch];      // device start vectors
    double* d[kMaxBatch];
    double* gram[kMaxBatch];
    DevSlot* tab;
};

void launch_dev_table_init(bioen_hip_ctx* c, int nslots);                // roles <- the slots' own buffers
void launch_dev_start(bioen_hip_ctx* c, const DevStart& s, const bioen_lbfgs_config& cfg);
void launch_dev_step(bioen_hip_ctx* c, const DevRound& r);               // [d = sum cf B ;] x = xp + stp d ; block maxima
void launch_dev_exp(bioen_hip_ctx* c, const DevRound& r);                // e = exp(x - m) ; prior partials
void launch_dev_grad_gram(bioen_hip_ctx* c, const DevRound& r);          // gradient + 3 dots ; (s, y) -> spare ; 39 Gram dots
void launch_dev_decide(bioen_hip_ctx* c, const DevRound& r, const bioen_lbfgs_config& cfg, unsigned long long round);
void launch_dev_first_direction(bioen_hip_ctx* c, const DevRound& r, int mask);   // d = -gp, partials of gp.d for the owners in `mask`
void launch_dev_store_dginit(bioen_hip_ctx* c, const DevRound& r, int mask);      //   ... finished -> scal[S_DGINIT]   [after the X_DGI exchange]
// An ADOPTED shadow evaluation that is accepted has no Gram products yet (shadows skip that part of the sweep): the pair
// and its 39 products are formed now (k_gram through the role table), then the recursion (k_gram_solve).  Gated on the
// problems' `combine` word: both return at once in the usual round.
void launch_dev_late_gram(bioen_hip_ctx* c, const DevRound& r);     // [exchange X_GRAM / X_GRAMR]
void launch_dev_late_solve(bioen_hip_ctx* c, const DevRound& r);
int dev_all_fused(const bioen_hip_ctx* c);                               // the decision kernel finishes the Gram sums itself
void launch_dev_rank_reduce(bioen_hip_ctx* c, const DevRound& r);        // sharded: this rank's totals of the 39 + 3 sums -> X_GRAMR
constexpr int kDevRankSums = kGramDots + 3;

// ---- misc ---------------------------------------------------------------------------
void launch_generate(bioen_hip_ctx* c, const double* YTrue, const double* sig_sim, const double* sig_exp,
                     unsigned long long seed);

}  // namespace bioen
