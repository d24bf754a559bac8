// Context and device-buffer layout of the MI355X BioEn hot path.
//
// HBM layout
//   Y      : yTilde, row-major, padded to  mp x ld  doubles
//              ld = round_up(n, 128)   (every row starts on a 1 KiB boundary, so a
//                                       wave's 64 x 16 B load is one aligned KiB)
//              mp = round_up(m, 32)    (4 waves x 8 rows per forward block; 4 x 8
//                                       unrolled rows per adjoint block)
//            padding is zero, so no matrix kernel needs an edge branch.
//   slots  : up to kMaxBatch optimisation problems (thetas) live side by side and share
//            every pass over Y ("lock-step batch").  Per slot: N-vectors x, xp, g, gp,
//            d, w, a and the L-BFGS history S[6], Y[6] (ld doubles each, pad = 0), 32
//            device-resident scalars and 16 x 1024 reduction partials.
//   M-side : YT (targets), and per round the COMPACT interleaved arrays
//            ybar_c[row*K + a], r_c[row*K + a] (a = position in the round's batch), so
//            the adjoint kernel fetches the K wave-uniform operands of a row with one
//            scalar load.
//   Only line-search decisions ever cross PCIe (kMaxBatch x 32 doubles per round).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/bioen_hip.h"

namespace bioen {

constexpr int kHistory = 6;        // liblbfgs default m (lbfgs.c:113), never overridden by BioEn
constexpr int kColAlign = 128;     // doubles: one wave x 16 B
constexpr int kRowAlign = 32;
constexpr int kMaxPartials = 1024; // upper bound on any reduction grid
constexpr int kMaxBatch = 8;       // thetas sharing one matrix pass
constexpr int kMaxSeg = 8;         // canonical column segments a context can hold (device_utils.hpp: segments)
constexpr int kScalStride = 40;    // doubles per slot in `scal`
constexpr int kLiveRing = 4;       // pages of the device engine's live ring (rounds in flight + being read)
constexpr int kBasis = 2 * kHistory + 1;   // S[6], Y[6], g
constexpr int kGramDots = 3 * kBasis;      // new s, new y, new g against the basis
constexpr int kGramStride = 256;           // doubles per slot: 13x13 Gram matrix + 13 coefficients + 39 sums
constexpr int kGramSums = kBasis * kBasis + kBasis;   // offset of the finished sums
static_assert(kGramSums + kGramDots <= kGramStride, "gram slot");

// device-resident scalar slots (per problem slot)
enum ScalarSlot : int {
    S_F = 0,      // objective
    S_DG,         // grad . d   at the trial point
    S_GG,         // grad . grad
    S_XX,         // x . x
    S_DGINIT,     // gp . d     at the accepted point (for the next line search)
    S_LOGS,       // log sum exp(x)
    S_P,          // sum_j w_j (x_j - G_j)
    S_CHI,        // sum_i r_i^2
    S_C,          // sum_i ybar_i r_i
    S_LOGS0,      // log sum exp(G)
    S_KL,         // forces: sum_j w_j log(w_j / w0_j)
    S_TSUM,       // forces: sum_j t_j
    S_YS,         // y.s of the newest pair
    S_YY,         // y.y of the newest pair
    S_SPARE0,
    S_SPARE1,
    S_YSH,                         // [kHistory] y.s per history slot
    S_ALPHA = S_YSH + kHistory,    // [kHistory]
    S_INV = S_ALPHA + kHistory,    // [kMaxSeg] w_j = e_j * S_INV[v] in local segment v (deferred softmax normalisation;
                                   //           every segment shifts its exponentials by its own maximum)
    S_B0 = S_INV + kMaxSeg,        // sum_i center_i r_i   (strip passes with centred operands: the adjoint's constant)
    S_UY,                          // sum_i ybar_raw_i r_i
    S_COUNT
};
static_assert(S_COUNT <= kScalStride, "scalar slots");

// local (never exchanged) partial-reduction arrays, each kPartStride doubles (kernels.hpp), per problem slot
enum PartSlot : int {
    P_MAX = 0, P_SUM, P_PP, P_CHI, P_C, P_KL,
    P_COUNT = 8
};

// Exchange stages.  A reduction over the N structures is produced as per-block partials and
// consumed by the NEXT kernel, which re-sums them in a fixed order in its prologue.
//
// CANONICAL SEGMENTS (r05).  The N columns are cut into `nseg` segments of `segcols` columns (nseg = 8 whenever the
// number of ranks divides 8 -- 1, 2, 4, 8 GPUs --, else nseg = world; segcols = ceil(N / nseg) rounded up to 128), and
// EVERY reduction over structures has the shape  sum over segments (in index order) of [a fixed tree inside the
// segment]:  the shape never depends on how many GPUs hold the segments.  A rank holds the vr = nseg / world
// consecutive segments [seg0, seg0 + vr); a stage buffer is laid out [segment][problem a][array q][block]; a rank fills
// its vr segments and ONE in-place all-gather per stage (vr x payload doubles per rank) makes all nseg visible
// everywhere.  One GPU holds all eight segments and runs no collective -- and produces, bit for bit, what 2, 4 or 8
// GPUs produce: the reference's fast_openmp = 0 guarantee (c_bioen_common.c:46-55, c_bioen_kernels_logw.c:58-93:
// the same sums whatever the thread count), carried over to the rank count.
enum XStage : int {
    X_MAX = 0,   // 1 array : block maxima of the trial point (consumed by the same rank: never exchanged)
    X_EXP,       // 3 arrays: sum e, sum e (x - G), [0] = this rank's shift m_r (rank-local; its per-rank
                 //           totals ride on X_YBAR -- exchanged only by bioen_hip_logw_weights)
    X_YBAR,      // mp values per problem: this rank's share of yTilde . v ; log-weights rounds append
                 //           {sum e, sum e (x - G), m_r} per problem (v = e, normalised by the consumer)
    X_GRAD,      // 3 arrays: g.d, g.g, x.x
    X_SY,        // 2 arrays: y.s, y.y
    X_REC0,      // 1 array : running dot of the two-loop recursion (ping)
    X_REC1,      //                                                  (pong)
    X_DGI,       // 1 array : gp . d
    X_VEC,       // ld values per problem: result vectors at the end
    X_GRAM,      // kGramDots arrays: inner products of (s, y, g) with the 13 basis vectors (block partials)
    X_GRAMR,     // kGramDots values per problem: the rank's totals of X_GRAM -- what sharded contexts exchange
    X_COUNT
};

struct Xch {              // one stage, as the kernels see it
    double* base;         // [nseg][payload]
    int payload;          // doubles per SEGMENT in this launch
    int world;            // nseg: segments in the stage (all ranks')
    int rank;             // seg0: first segment this context fills
    int npl;              // blocks per array and segment (= blocks per segment of the producing N-vector kernel)
    int vr;               // segments this context fills (and holds the columns of)
    int segcols;          // columns per segment
};

struct KernelTimer {
    bool enabled = false;
    double total_ms[2] = {0.0, 0.0};
    long long launches[2] = {0, 0};
    long long problem_passes[2] = {0, 0};   // sum over launches of the batch width K
    struct Pair { hipEvent_t a, b; int which; int k; };
    std::vector<Pair> pending;
    std::vector<Pair> pool;
    hipEvent_t cur_a = nullptr, cur_b = nullptr;   // events of the enclosing TimedLaunch (device_utils.hpp), else NULL
};

struct ProblemSlot {
    bool allocated = false;
    bool history = false;
    double *xa = nullptr, *xb = nullptr, *ga = nullptr, *gb = nullptr;   // storage
    double *x = nullptr, *xp = nullptr, *g = nullptr, *gp = nullptr;     // current roles
    double *d = nullptr, *w = nullptr, *a = nullptr;
    double* S[kHistory] = {};
    double* Yh[kHistory] = {};
    double* scal = nullptr;   // kScalStride doubles inside ctx->scal
    double *Ssp = nullptr, *Ysp = nullptr;   // spare (s, y) pair of the device-resident engine (kernels.hpp: DevSlot)
    double* part = nullptr;   // P_COUNT * kPartStride doubles inside ctx->part (local partials)
    double* gram = nullptr;   // kGramStride doubles inside ctx->gram
};

}  // namespace bioen

struct bioen_hip_ctx {
    int device = 0;
    int m = 0, n = 0;          // n = structures held by THIS rank
    int mp = 0;
    size_t ld = 0;
    // structure (column) sharding over GPUs: this context holds columns [col0, col0 + n) of n_global
    int rank = 0, world = 1;
    long long n_global = 0, col0 = 0;
    // canonical segments (above): nseg in all, this context holds [seg0, seg0 + vr), each segcols columns; ld = vr * segcols
    int nseg = 8, vr = 8, seg0 = 0, segcols = 0;
    double* xbuf[bioen::X_COUNT] = {};   // exchange stage buffers, each world * capacity doubles
    size_t xcap[bioen::X_COUNT] = {};    // capacity (doubles) per RANK (= vr segments)
    // host-staged exchange for processes that cannot share an RCCL communicator (tests)
    int (*exchange_cb)(void* user, double* host_buf, size_t count_per_rank) = nullptr;
    void* exchange_user = nullptr;
    double* exchange_host = nullptr;
    char* stage_host = nullptr;      // pinned, 8 MB, on first need: uploads of caller buffers the runtime refuses to pin (api.hip: h2d_staged)
    size_t exchange_host_count = 0;
    int exchange_error = 0;
    int force_exchange = 0;              // world == 1: run the stage exchanges all the same (through the communicator or the
                                         // callback, if there is one) -- puts the RCCL stage path under single-GPU tests
    int mirror_exchange = 0;             // measurement aid (bioen_hip_ctx_set_mirror_exchange): an exchange copies this rank's part over the others'
    long long n_rccl_exchanges = 0, n_host_exchanges = 0;   // stage all-gathers executed so far, by transport
    long long n_p2p_exchanges = 0;
    // Peer-to-peer stage exchange (r04; kernels_p2p.hip): every rank owns a MAILBOX in its HBM -- two halves (by the
    // parity of the exchange number) of `world` slots of p2p_cap doubles, behind 2 x world flags -- which every peer
    // maps through hipIpc.  One small kernel per exchange: block p stores this rank's segment into peer p's mailbox,
    // releases the exchange number into p's flag for this rank (system scope), waits for p's flag in its OWN mailbox and
    // copies p's segment into the stage buffer.  No collective launch, no host.  A wait is bounded (wait_timeout_s): on
    // expiry the kernel records the failure in p2p_err / p2p_dev_err and every later exchange publishes an ABORT flag.
    double* p2p_box = nullptr;               // this rank's mailbox (uncached / fine-grained device memory)
    size_t p2p_cap = 0;                      // doubles per slot
    size_t p2p_bytes = 0;
    double** p2p_peers = nullptr;            // device array [world]: every rank's mailbox as mapped here ([rank] = p2p_box)
    void* p2p_mapped[128] = {};              // what hipIpcOpenMemHandle returned per peer (closed by p2p_detach)
    unsigned long long p2p_seq = 0;          // exchanges issued so far
    int p2p_on = 0;                          // attached: exchange() uses this transport
    unsigned long long* p2p_err = nullptr;   // host-mapped: first failure of an exchange kernel (0 = none)
    unsigned long long* p2p_dev_err = nullptr;   // the same word in device memory (read by the later kernels)
    unsigned int* p2p_cnt = nullptr;         // [world] arrival counters of the multi-block form (large segments)
    double wait_timeout_s = 60.0;            // BIOEN_HIP_WAIT_TIMEOUT: bound of every host and device wait on a round
    int failed = 0;                          // a wait expired or a transport failed: every later call returns at once
    int failed_p2p = 0;                      // ... and it was the peer-to-peer transport's own failure (detaching it clears it)
    std::string fail_msg;                    // ... with what happened first
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // results of finished problems leave on this one (engine_logw.inl: deliveries)

    double* Y = nullptr;       // mp x ld, row-major: the form data arrive in; M <= 1024: freed once the strip copy Ys
                               // exists (kernels_strip.hip: ensure_strip_copy), back on demand (ensure_rowmajor);
                               // (M > 1024: the copies are row panels, Yp / Y1p)
    int keep_rowmajor = 0;     // BIOEN_HIP_KEEP_ROWMAJOR=1: never free it (A/B)
    int rowmajor_rebuilt = 0;  // it was freed and has been re-created since
    double* zero_center = nullptr;   // mp zeros: "no centring" for the strip kernels (bioen_hip_chi_squared)
    // M <= 1024 (kernels_strip.hip): strip-major copies of the RAW matrix, built on first use; the kernels centre the
    // operands on strip_center on the fly
    double* Ys = nullptr;            // [ld / 16][strip rows][16], row-sum operand order (forces, log-weights forward)
    double* Ys1 = nullptr;           // the same strips in column-sum operand order (log-weights adjoint)
    double* strip_center = nullptr;  // mp: YTilde at the time of the copy
    // reduced-byte storage EXPERIMENT (kernels_strip.hip; bioen_hip_ctx_set_storage): the log-weights passes stream
    // CENTRED copies of 6 (fp32 + bf16 residual) or 4 (fp32) bytes per element; the row-major FP64 matrix stays resident
    int storage = 0;                 // 0 FP64 (default, the graded path) | 1 fp32 + bf16 split | 2 fp32
    void* Yr = nullptr;              // row-sum operand order
    void* Yr1 = nullptr;             // column-sum operand order
    // M > 1024 (r03): the matrix passes of both methods run the same kernels over PANELS of <= 1024 rows, each with its
    // own pair of strip copies; the row-major matrix is freed once the row-sum panels exist, as for M <= 1024
    static constexpr int kMaxPanels = 16;
    double* Yp[kMaxPanels] = {};     // row-sum order copy of rows [1024 p, 1024 (p + 1))
    double* Y1p[kMaxPanels] = {};    // column-sum order copy
    int panel_off = 0;               // BIOEN_HIP_PANELS=0: the r01 streaming kernels for M > 1024 (A/B)
    double* strip_stamps = nullptr;  // diagnostic builds only: [block][16 waves][8] phase-cycle sums of the last strip launch
    int fwd_stream = 0;              // BIOEN_HIP_FWD_STREAM=1: log-weights forward pass by k_fwd_partial (A/B)
    int strips_unavailable = 0;      // a strip copy could not be allocated: the streaming kernels serve this context
    // r05: the log-weights method on ONE strip copy (the row-sum order one; the adjoint through the forces kernels' LDS image,
    // kernels_strip.hip: k_strip<.., ADJ>): 1 x the matrix resident instead of 2 x.  wanted: BIOEN_HIP_ONE_COPY=1; taken
    // by itself when the column-sum order copy cannot be allocated.  M <= 1024, FP64 storage.
    int one_copy = 0, one_copy_wanted = -1;      // wanted: 1 / 0 asked for / refused, -1 (r06 default): by the matrix's size -- ONE copy above 1 GiB
    int strip_ilv = 0;               // r06: segments interleaved in the row-sum order FP64 copies (kernels_strip.hip: strip_phys); 0: none built yet
    int strip_relayouts = 0;         // times the copies were moved to another method's layout
    int strip_allocs = 0;            // strip-copy allocations attempted on this context (tests: BIOEN_HIP_TEST_FAIL_STRIP_ALLOC=k fails the k-th)
    double* YT = nullptr;      // mp   experimental targets (YTilde)
    // affine observable model: yTilde_eff[i][j] = row_offset[i] + row_scale[i] * Y[i][j]
    // (default 0, 1).  DEER / SAXS nuisance parameters enter exactly like this, so a refit never
    // touches Y.
    double* row_offset = nullptr;   // mp
    double* row_scale = nullptr;    // mp
    bool affine = false;            // anything but (0, 1)
    double* ybar_c = nullptr;  // mp * kMaxBatch, compact per round
    int last_width = 1, last_pos = 0;   // width of the round that wrote ybar_c last / column of the problem
                                        // bioen_hip_last_average hands out (a finished problem's, else 0);
                                        // width 0: nothing to hand out (a multi-problem call ran last)
    bool last_centered = false;         // ybar_c holds ybar - strip_center (forces strip passes), not the raw average
    double* r_c = nullptr;     // mp * kMaxBatch
    double* um = nullptr;      // mp * kMaxBatch  forces of the round's problems, compact [row*K + a]
    double* gm = nullptr;      // mp * kMaxBatch  forces gradients, compact
    double* fixed = nullptr;   // ld   G (log-weights) or w0 (forces), shared by all slots
    double* t = nullptr;       // ld   forces scratch
    double* g0 = nullptr;      // ld   shared start vector of a batch run (lazy)

    bioen::ProblemSlot slot[bioen::kMaxBatch];
    // device-resident line-search decisions (kernels.hpp: DevSlot; engine_devls.inl)
    void* dev_tab = nullptr;             // DevSlot[kMaxBatch], device memory
    double* live2 = nullptr;             // host-mapped: kLiveRing pages of kMaxBatch records | kLiveRing x kMaxBatch flags
    unsigned long long dev_round = 0;    // last round number handed out

    double* fwd_partial = nullptr;   // kMaxBatch * mp * fwd_ctiles, compact per round
    int fwd_ctiles = 0;              // column tiles of the forward pass
    int fwd_steps = 0;               // 128-column steps per tile
    double* part = nullptr;          // kMaxBatch * P_COUNT * kPartStride
    double* scal = nullptr;          // kMaxBatch * kScalStride
    double* gram = nullptr;          // kMaxBatch * kGramStride
    int direction_mode = 0;          // 0 auto (= Gram form), 1 two-loop on the vectors, 2 Gram form
    double* host_scal = nullptr;     // pinned mirror
    // Live hand-off of a round's scalars (log-weights batch engine): the round's last kernel stores each problem's
    // slot straight into this host-mapped page and then the round number into its flag; the host spins on the flags
    // instead of queueing a copy and sleeping on the stream (saves the copy kernel and the wake-up, ~15 us a round).
    double* live = nullptr;                        // pinned, coherent: kMaxBatch * kScalStride values | kMaxBatch flags
    double* live_f = nullptr;                      // forces engine, pinned, coherent: gradients mp x kMaxBatch | scalars kMaxBatch x kScalStride | 1 flag
    unsigned long long forces_round = 0;           // last round number published there
    unsigned long long live_seq = 0;               // last round number handed out
    unsigned long long live_round = 0;             // != 0: the next launch_finish_eval publishes under this number
    int live_off = 0;                              // BIOEN_HIP_LIVE=0: copy + stream synchronisation as before (A/B)
    double* host_m = nullptr;        // pinned, 2 x mp*kMaxBatch: forces up / gradients down (lazy)

    long long spec_launched = 0, spec_used = 0;   // speculative line-search evaluations issued / adopted (engine_logw.inl)
    bool nontemporal = true;         // stream yTilde with nt loads (matrix larger than MALL)
    bool nvec_nt = false;            // the batch's N-vectors are far larger than the caches: history loads and outputs of the
                                     // round's N-vector kernels nontemporal (kernels_logw.hip); set per run by the host-driven
    int nvec_nt_env = -1;            // engine from the batch's working set; BIOEN_HIP_NVEC_NT=0/1 forces it
    bioen::KernelTimer timer;

    // RCCL (lazy, dlopen)
    void* comm = nullptr;
    int comm_rank = 0, comm_nranks = 1;
    double* comm_buf = nullptr;
    size_t comm_buf_count = 0;
};

namespace bioen {

void set_last_error(const std::string& s);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define BIOEN_HIP_CHECK(expr)                                                      \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) return ::bioen::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace bioen
