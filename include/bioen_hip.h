/* bioen_hip.h -- C ABI of the MI355X (gfx950) BioEn optimizer hot path.
 *
 * This is the boundary a BioEn maintainer binds instead of the Cython module
 * bioen/optimize/ext/c_bioen.pyx (ctypes stub: bioen_amd/optimize/ext/c_bioen.py,
 * walk-through: INTEGRATION.md).  Plain C: pointers, sizes, PODs.  No torch
 * types, no C++ types.  All arithmetic is IEEE double.
 *
 * Ownership: every `const double*` / `double*` argument is a HOST buffer owned
 * by the caller and only read (inputs) or written (outputs) during the call;
 * nothing is retained after return (the batch optimizers register large result arrays with the
 * HIP runtime -- hipHostRegister -- while the call runs, so that finished problems leave by DMA,
 * and unregister them before they return).  Device memory belongs to the context and
 * is released by bioen_hip_ctx_destroy().  A context is not re-entrant (one
 * HIP stream, one scratch set); distinct contexts are independent.
 *
 * Return value: 0 on success, a negative BIOEN_HIP_E* code otherwise
 * (bioen_hip_strerror() gives the text; bioen_hip_last_error() the detail).
 * The L-BFGS drivers additionally report the liblbfgs status code through
 * `lbfgs_code` (0,1,2 = success; negative = liblbfgs error numbering,
 * third-party/liblbfgs-1.10/include/lbfgs.h:76-147), which the Python shim
 * turns into the reference's RuntimeError("... return code ...").
 */
#ifndef BIOEN_HIP_H
#define BIOEN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BIOEN_HIP_OK 0
#define BIOEN_HIP_EINVAL (-1)    /* bad argument (NULL, non-positive size, ...) */
#define BIOEN_HIP_ENODEV (-2)    /* no usable HIP device */
#define BIOEN_HIP_EHIP (-3)      /* a HIP runtime call failed */
#define BIOEN_HIP_ENOMEM (-4)    /* device or host allocation failed */
#define BIOEN_HIP_ERCCL (-5)     /* RCCL unavailable or an RCCL call failed */
#define BIOEN_HIP_ESTATE (-6)    /* call not valid in the context's state */

typedef struct bioen_hip_ctx bioen_hip_ctx;

/* Same fields, same order as the reference's lbfgs_config_params
 * (bioen/optimize/ext/c_bioen_common.h:69-79) so the shim mirrors its packing. */
typedef struct bioen_lbfgs_config {
    int linesearch;      /* 0 More-Thuente, 1 Armijo, 2 Wolfe (yaml default), 3 strong Wolfe */
    int max_iterations;
    double delta;
    double epsilon;
    double ftol;
    double gtol;
    double wolfe;
    int past;
    int max_linesearch;
} bioen_lbfgs_config;

/* Same fields, same order as the reference's gsl_config_params (c_bioen_common.h:62-67);
 * algorithm ids of c_bioen_common.h:28-34: 0 conjugate_fr, 1 conjugate_pr, 2 vector_bfgs2,
 * 3 vector_bfgs, 4 steepest_descent. */
typedef struct bioen_gsl_config {
    double step_size;
    double tol;
    int max_iterations;
    int algorithm;
} bioen_gsl_config;

/* reference visual_params, c_bioen_common.h:89-92 */
typedef struct bioen_visual_params {
    size_t debug;
    size_t verbose;
} bioen_visual_params;

/* What the reference only prints (c_bioen_kernels_logw.c:646-653) is returned. */
typedef struct bioen_opt_result {
    double fmin;          /* final negative log-posterior                        */
    double chi2;          /* 0.5 * |yTilde w - YTilde|^2 at the optimum          */
    double kl;            /* KL(w || w0) = -S at the optimum (S: utils.py:83-106) */
    double seconds;       /* wall time inside the minimiser (device-synchronised) */
    int lbfgs_code;       /* liblbfgs status (GSL status for the opt_gsl_* entry points) */
    int iterations;       /* progress-callback count (accepted line searches)     */
    int evaluations;      /* objective+gradient evaluations                       */
    int reserved;
} bioen_opt_result;

/* ---- library / device ------------------------------------------------------ */
const char* bioen_hip_version(void);
int bioen_hip_device_count(int* count);
const char* bioen_hip_strerror(int code);
const char* bioen_hip_last_error(void);
/* replaces lbfgs_strerror(), bioen/optimize/ext/c_bioen_error.c:23-115 */
const char* bioen_hip_lbfgs_strerror(int lbfgs_code);
/* replace _set_fast_openmp_flag/_get_fast_openmp_flag (c_bioen_common.c:46-55): accepted and stored.  What the
 * reference's flag = 0 buys -- the same bits whatever the thread count (serial sums, c_bioen_kernels_logw.c:58-93;
 * test/optimize/test_logw_reproducibility.py:14-46) -- holds here for BOTH values, and across GPU counts: every sum over
 * structures is formed per canonical column segment and the segments' totals are added in segment order (see the
 * sharded contexts below), so 1, 2, 4 and 8 GPUs return identical bits. */
void bioen_hip_set_fast_openmp_flag(int flag);
int bioen_hip_get_fast_openmp_flag(void);

/* ---- context: yTilde (m x n, row-major, host) is uploaded once and stays
 *      resident in HBM across evaluations, thetas and minimiser runs.
 *      Replaces the per-call pointer bag params_t (c_bioen_common.h:44-60) and
 *      the host-side transposed copy of c_bioen.pyx:471-473 (not needed). ---- */
int bioen_hip_ctx_create(int m, int n, const double* yTilde, const double* YTilde,
                         int device, bioen_hip_ctx** ctx);
/* Synthetic ensemble generated directly in HBM (bench / scale tests):
 *   yTilde[i][j] = (YTrue[i] + sig_sim[i] * z_ij) / sig_exp[i],  z_ij ~ N(0,1)
 * with z_ij a counter-based Box-Muller stream of (seed, i, j); recipe after
 * forces.gen_sythetic_ensemble (bioen/optimize/forces.py:44-68). */
int bioen_hip_ctx_create_synthetic(int m, int n, const double* YTrue, const double* sig_sim,
                                   const double* sig_exp, const double* YTilde,
                                   unsigned long long seed, int device, bioen_hip_ctx** ctx);
/* Structure-sharded contexts (multi-GPU, one process per GPU).  The n columns are cut into CANONICAL SEGMENTS: 8 of
 * them whenever `world` divides 8 (1, 2, 4, 8 GPUs), else `world`; S = ceil(n / segments) rounded up to 128 columns each.
 * Rank r keeps the (segments / world) consecutive segments from r * (segments / world) on, i.e. the column block
 * [r*P, min(n, (r+1)*P)), P = S * segments / world -- `yTilde` is the caller's FULL row-major matrix, only the block is
 * uploaded (resp. generated).  All N-vector arguments of the calls below stay GLOBAL (n long) on every
 * rank; the library slices inputs and gathers outputs.  Every reduction over structures is formed per segment and the
 * segments' totals are added in segment order -- the SAME shape on every GPU count, a single GPU included: contexts of
 * 1, 2, 4 and 8 ranks return bit-identical results.  Across ranks a reduction is completed by one in-place all-gather
 * per stage through the peer-to-peer mailboxes (bioen_hip_p2p_attach), RCCL (bioen_hip_comm_init) or, for
 * processes that can share neither, a host callback.  Every rank must
 * issue the same calls in the same order.  Supported on sharded contexts: logw_weights, logw_fdf, chi_squared,
 * opt_lbfgs_logw(_batch), opt_gsl_logw, and the forces method (forces_weights, forces_fdf(_batch),
 * opt_lbfgs_forces(_batch), opt_gsl_forces) for every M served by strip copies (M <= 1024: two passes; beyond: four
 * passes over row panels).
 * Minimum size: the LAST rank must still own a column, (world - 1) * P < n -- n > 512 for two ranks, > 768 for four,
 * > 896 for eight (S is at least 128); BIOEN_HIP_EINVAL ("too few structures to shard") on EVERY rank otherwise, so that
 * none waits in a collective for one that failed.  The leading dimension of the local data is segments-per-rank * S. */
int bioen_hip_ctx_create_sharded(int m, long long n, const double* yTilde, const double* YTilde,
                                 int device, int rank, int world, bioen_hip_ctx** ctx);
int bioen_hip_ctx_create_synthetic_sharded(int m, long long n, const double* YTrue,
                                           const double* sig_sim, const double* sig_exp,
                                           const double* YTilde, unsigned long long seed, int device,
                                           int rank, int world, bioen_hip_ctx** ctx);
/* host_buf = [world][count_per_rank] doubles; on entry this rank's segment is filled, on
 * return (0 = ok) all segments must be.  Called from the thread that called into the library. */
typedef int (*bioen_hip_exchange_fn)(void* user, double* host_buf, size_t count_per_rank);
int bioen_hip_ctx_set_exchange_callback(bioen_hip_ctx* ctx, bioen_hip_exchange_fn fn, void* user);
int bioen_hip_ctx_shard(const bioen_hip_ctx* ctx, int* rank, int* world, long long* n_global,
                        long long* col0, int* n_local);
/* Test hook for the stage exchanges: on an UNSHARDED context (world = 1) that has a communicator
 * (bioen_hip_comm_init(ctx, id, 0, 1)) or an exchange callback, every stage all-gather of the sharded
 * code path -- ybar + softmax totals, gradient dot products, Gram products; the forces method's two --
 * is executed although there is nobody to exchange with: a one-rank ncclAllGather on the context's stream,
 * between the same kernels as on 8 GPUs, leaving every bit of the result unchanged.  Also switched on by
 * BIOEN_HIP_FORCE_EXCHANGE=1 in the environment at context creation.  bioen_hip_exchange_counts tells how
 * many stage all-gathers went through RCCL / through the host callback so far. */
int bioen_hip_ctx_set_force_exchange(bioen_hip_ctx* ctx, int on);
/* Measurement aid: a MIRROR exchange.  A context created as rank r of `world` whose stage all-gathers copy ITS OWN part
 * into every other rank's part of the stage buffer -- one small kernel per exchange, no peer, no host.  The run is that of
 * a problem whose `world` column blocks are all equal to this rank's: numerically a valid problem, and kernel for kernel
 * the work ONE rank of a `world`-GPU run does per round, measured on a single GPU (bench.py, tools/engine_ab.py: the
 * per-rank share behind the multi-GPU projection).  Results are returned as for any sharded context (every block of
 * the gathered vectors equals this rank's). */
int bioen_hip_ctx_set_mirror_exchange(bioen_hip_ctx* ctx, int on);
int bioen_hip_exchange_counts(const bioen_hip_ctx* ctx, long long* rccl, long long* host_staged);
int bioen_hip_ctx_destroy(bioen_hip_ctx* ctx);
int bioen_hip_ctx_shape(const bioen_hip_ctx* ctx, int* m, int* n);
/* copy rows [row0,row0+rows) x cols [col0,col0+cols) of the resident matrix to host (row-major): the caller's
 * numbers bit for bit, whichever form of the matrix is resident */
int bioen_hip_ctx_read_ytilde(bioen_hip_ctx* ctx, int row0, int rows, int col0, int cols, double* out);
/* Which forms of the matrix are resident and what they occupy.  forms: bit 0 the row-major matrix (the form data
 * arrive in; the operand of the streaming kernels for M > 1024), bit 1 the strip-major copy in row-sum operand order,
 * bit 2 the one in column-sum operand order.  For M <= 1024 the strip copies hold the raw numbers and REPLACE the
 * row-major matrix once built: log-weights 2 x the matrix (bits 1 + 2; r06: 1 x, bit 1 alone, once a copy exceeds 1 GiB), forces method 1 x (bit 1; beyond 1024 rows,
 * where the copies are kept as row panels of <= 1024 rows, 2 x: both orders).  ONE copy (r05: bit 1 alone; log-weights at every M, and the forces method's row panels beyond 1024 rows): environment BIOEN_HIP_ONE_COPY=1 at context creation, or taken by itself when the second copy cannot
 * be allocated -- the adjoint then runs on the row-sum order copy (1-3 % slower per launch, same minima, last bits differ
 * from the two-copy default: the sums over rows are formed in another order). */
/* (bit 3: copies of the reduced-storage experiment, bioen_hip_ctx_set_storage, beside the FP64 row-major matrix) */
int bioen_hip_ctx_footprint(const bioen_hip_ctx* ctx, int* forms, long long* bytes);
/* r06: HOW the strip copies are held.  one_copy: 1 = the log-weights adjoint runs on the row-sum order copy (asked for, or
 * taken by THIS rank because the second copy could not be allocated -- its last bits then differ from a two-copy rank's:
 * sharded drivers compare this value across ranks, bioen_amd/sweep.py).  interleave: the local segments' strips are stored
 * interleaved by this many in the row-sum order copy (1: strip order) -- the log-weights passes want the context's local
 * segment count (one contiguous window of the copy is read at any moment), the forces passes 1; the copy is moved when the
 * other method starts on the context (relayouts counts the moves; results never depend on the layout).
 * BIOEN_HIP_STRIP_INTERLEAVE=0: strip order always (A/B). */
int bioen_hip_ctx_layout(const bioen_hip_ctx* ctx, int* one_copy, int* interleave, int* relayouts);
/* r05: ask for (1) / give up (0) the ONE-copy form described above; to be called before the context's first gradient
 * evaluation (BIOEN_HIP_ESTATE once the column-sum order copy exists).  Beyond 1024 rows it serves both methods: one set
 * of row panels instead of two.
 * r06: without this call (and without BIOEN_HIP_ONE_COPY=0 / 1 in the environment) the form follows the matrix's size:
 * a strip copy of the WHOLE matrix (all ranks' columns) above 1 GiB is kept once -- the one-copy adjoint then runs at the
 * two-copy kernel's time (the headline: 8.2 GB resident instead of 16.4, the sweep within its run-to-run spread) --
 * smaller matrices keep the dedicated second copy, whose kernel is the faster one where launches are short. */
int bioen_hip_ctx_set_one_copy(bioen_hip_ctx* ctx, int on);
int bioen_hip_ctx_set_ytilde_target(bioen_hip_ctx* ctx, const double* YTilde);
/* Affine observable model: the optimizer sees yTilde_eff[i][j] = row_offset[i] + row_scale[i] * yTilde[i][j]
 * without the resident matrix being touched.  This is how DEER (modulation depth m:
 * yTilde = 1/sigma + m (F-1)/sigma, bioen/analyze/observables/observables.py:133-134) and SAXS
 * (scaling c: yTilde = c I/sigma, :141) nuisance parameters enter, so the refit loop of
 * bioen/analyze/procedure.py:79-83 needs no rebuild / re-upload of yTilde (the reference rebuilds it
 * on the host, observables.py:110-143).  row_offset = NULL means 0, row_scale = NULL means 1 (both m
 * long; one value per DEER trace / data set, repeated over its rows).  Log-weights method;
 * chi_squared() returns the raw yTilde . w and the chi^2 of the affine model. */
int bioen_hip_ctx_set_affine(bioen_hip_ctx* ctx, const double* row_offset, const double* row_scale);
/* EXPERIMENT, opt-in, never the default and never part of a headline number ("keep FP64 as the graded path;
 * treat FP32/BF16-split as an experiment"; 8 f4): the two matrix passes of the LOG-WEIGHTS evaluation stream copies
 * of the centred operand yTilde_ij - YTilde_i held in fewer bytes and reassembled to FP64 in registers in front of the
 * FP64 matrix-core products (all sums stay FP64):
 *   format 1: fp32 high part + bf16 residual, 6 bytes per element, |error| <= 2^-33 of the centred element;
 *   format 2: fp32, 4 bytes, 2^-25;       format 0: back to FP64 (8 bytes, exact).
 * The FP64 row-major matrix stays resident beside them (read-back, chi_squared).  M <= 1024.  The forces method's fused
 * strip passes (M <= 1024: forces_fdf(_batch), the forces optimizers) stream the same reduced copy; everything else of
 * the forces method (forces_weights, M > 1024) returns BIOEN_HIP_ESTATE while a reduced format is selected.  No reference counterpart: the reference computes in double
 * throughout (bioen/optimize/ext/c_bioen_common.c:70-108). */
int bioen_hip_ctx_set_storage(bioen_hip_ctx* ctx, int format);
/* How the L-BFGS direction d = -H g is formed on the device.
 *   1  two-loop recursion on the vectors, liblbfgs' order of operations (lbfgs.c:571-598):
 *      13 fused vector sweeps and 13 dependent reductions per direction;
 *   2  the same recursion carried out on coefficients over {S_0..5, Y_0..5, g} from their
 *      inner products: one sweep commits (s, y) and yields all 39 new products, one sweep forms
 *      d -- half the vector traffic and ONE reduction stage instead of 14 (what matters when
 *      every stage is an all-gather between GPUs).  Mathematically identical; rounding differs;
 *   0  auto (default): 2.  Measured on one MI355X, same process: N = 1e6 x M = 1024 8-theta series
 *      1.54 s vs 1.62-1.68 s; N = 1e5 x M = 256 single theta 176 vs 200 us per iteration. */
int bioen_hip_ctx_set_direction_mode(bioen_hip_ctx* ctx, int mode);
int bioen_hip_synchronize(bioen_hip_ctx* ctx);

/* ---- log-weights method ---------------------------------------------------- */
/* _get_weights, c_bioen_kernels_logw.c:55-94: w = softmax(g); *log_s = log sum exp(g) */
int bioen_hip_logw_weights(bioen_hip_ctx* ctx, const double* g, double* w, double* log_s);
/* interface_lbfgs_logw, c_bioen_kernels_logw.c:525-561 (= _get_weights +
 * _bioen_log_posterior_logw :131-147 + _grad_bioen_log_posterior_logw :151-268).
 * `f` and/or `grad` may be NULL (f-only skips the adjoint pass). */
int bioen_hip_logw_fdf(bioen_hip_ctx* ctx, const double* g, const double* G, double theta,
                       double* f, double* grad);
/* _opt_lbfgs_logw, c_bioen_kernels_logw.c:581-669, with the liblbfgs loop
 * (lbfgs.c:245-641) device-resident.  result[n] = optimal log-weights;
 * w_opt[n] (optional, may be NULL) = softmax(result).
 * Inputs are not validated (the reference validates nothing, c_bioen.pyx:274-290).  Non-finite ones end the run the way the
 * reference's binary -- liblbfgs built with -ffast-math -- ends it: NaN / inf in the start point, prior, matrix, targets
 * or theta: lbfgs_code 2, the start point, a non-finite fmin, after one evaluation; a direction whose slope is not a
 * number (a degenerate pair beyond the rounding floor) or positive: lbfgs_code -994 and the last accepted point with ITS
 * objective.  The same holds for the forces method and for every member of a batch on its own. */
int bioen_hip_opt_lbfgs_logw(bioen_hip_ctx* ctx, const double* g0, const double* G, double theta,
                             const bioen_lbfgs_config* config, const bioen_visual_params* visual,
                             double* result, double* w_opt, bioen_opt_result* info);

/* A whole theta series in one call.  Up to `max_batch` (<= 8) thetas advance in lock step --
 * one evaluation each per round -- and SHARE every pass over yTilde, so the matrix bytes per
 * theta drop by the batch width; finished thetas hand their slot to the next one.  Each theta
 * does exactly the arithmetic of a single run (results are bitwise identical to
 * bioen_hip_opt_lbfgs_logw called per theta).  This replaces the serial loop of
 * bioen/analyze/procedure.py:62-83 for cold-started series.
 *   thetas[ntheta]; g0: start log-weights, shared (g0_stride = 0) or per theta (stride >= n);
 *   results[ntheta][n]; w_opt[ntheta][n] or NULL; infos[ntheta].
 *   max_batch bounds the THETAS in flight.  Columns of the matrix passes that no theta occupies may carry speculative
 *   line-search trials of the thetas that are in flight (the steps a backtracking search asks for after a rejected trial;
 *   results are identical with and without them): up to 8 - max_batch on an unsharded context at the headline size, up to
 *   two on a structure-sharded context -- their N-vectors (seven per column) are allocated only where they can be used. */
int bioen_hip_opt_lbfgs_logw_batch(bioen_hip_ctx* ctx, int ntheta, const double* thetas,
                                   const double* g0, size_t g0_stride, const double* G,
                                   const bioen_lbfgs_config* config, const bioen_visual_params* visual,
                                   int max_batch, double* results, double* w_opt,
                                   bioen_opt_result* infos);

/* ---- forces method --------------------------------------------------------- */
/* _get_weights_from_forces, c_bioen_kernels_forces.c:111-224 */
int bioen_hip_forces_weights(bioen_hip_ctx* ctx, const double* forces, const double* w0, double* w);
/* interface_lbfgs_forces, c_bioen_kernels_forces.c:43-76 (= F1 + :227-277 + :280-340) */
int bioen_hip_forces_fdf(bioen_hip_ctx* ctx, const double* forces, const double* w0, double theta,
                         double* f, double* grad);
/* The same evaluation for K <= 8 force vectors at once (forces[K][m], thetas[K]; f[K], grad[K][m] or NULL):
 * the K problems share every pass over yTilde, each result equal to the last bit to the single call. */
int bioen_hip_forces_fdf_batch(bioen_hip_ctx* ctx, int k, const double* forces, const double* w0,
                               const double* thetas, double* f, double* grad);
/* _opt_lbfgs_forces, c_bioen_kernels_forces.c:574-662.  result[m] = optimal forces. */
int bioen_hip_opt_lbfgs_forces(bioen_hip_ctx* ctx, const double* forces0, const double* w0,
                               double theta, const bioen_lbfgs_config* config,
                               const bioen_visual_params* visual, double* result, double* w_opt,
                               bioen_opt_result* info);

/* theta series of the forces method in one call (cf. bioen_hip_opt_lbfgs_logw_batch): the M
 * variables of every problem stay on the host, the up to `max_batch` problems of a round share
 * all matrix passes of the evaluation (two for M <= 1024, else four).  forces0: shared (f0_stride = 0) or per theta
 * (stride >= m); results[ntheta][m]; w_opt[ntheta][n] or NULL. */
int bioen_hip_opt_lbfgs_forces_batch(bioen_hip_ctx* ctx, int ntheta, const double* thetas,
                                     const double* forces0, size_t f0_stride, const double* w0,
                                     const bioen_lbfgs_config* config, const bioen_visual_params* visual,
                                     int max_batch, double* results, double* w_opt,
                                     bioen_opt_result* infos);

/* ---- shared pieces --------------------------------------------------------- */
/* _bioen_chi_squared (c_bioen_common.c:70-108) / _getAve (c_bioen_kernels_forces.c:93-109):
 * yave[m] = yTilde . w ; *chi2 = 0.5 |yave - YTilde|^2.  Either output may be NULL.
 * With an affine row model set (bioen_hip_ctx_set_affine) chi2 is that of the EFFECTIVE observables
 * off_i + sc_i (yTilde . w)_i (sum w = 1 assumed, as everywhere in BioEn), while yave stays the RAW product
 * of the resident matrix -- the quantity a nuisance refit needs; bioen_hip_last_average hands out both. */
int bioen_hip_chi_squared(bioen_hip_ctx* ctx, const double* w, double* yave, double* chi2);
/* Averages left on the device by the most recent SINGLE-problem call (bioen_hip_chi_squared, bioen_hip_logw_fdf,
 * bioen_hip_forces_fdf, bioen_hip_opt_lbfgs_logw, bioen_hip_opt_lbfgs_forces, bioen_hip_opt_gsl_logw):
 * yraw[m] = yTilde . w of the resident matrix at the point that call ended on, yeff[m] = off + sc * yraw.  Either
 * may be NULL.  2 m doubles cross PCIe -- a refit between two optimizations (analyze/procedure.py:82-83) never moves
 * the N weights.  After a multi-problem call (*_batch with more than one problem) there is no single point to report:
 * BIOEN_HIP_ESTATE until the next single-problem call. */
int bioen_hip_last_average(bioen_hip_ctx* ctx, double* yraw, double* yeff);

/* ---- measurement hooks (bench.py) -------------------------------------------- */
/* Average device time (HIP events on the context's stream) and launch count of the
 * two matrix-streaming kernels since the last reset. which: 0 = forward, 1 = adjoint. */
int bioen_hip_kernel_stats(bioen_hip_ctx* ctx, int which, double* total_ms, long long* launches);
/* as above, plus the sum over launches of the batch width (problems served per matrix pass) */
int bioen_hip_kernel_stats_ex(bioen_hip_ctx* ctx, int which, double* total_ms, long long* launches,
                              long long* problem_passes);
int bioen_hip_kernel_stats_reset(bioen_hip_ctx* ctx);
int bioen_hip_kernel_stats_enable(bioen_hip_ctx* ctx, int enable);

/* Line-search evaluations the log-weights batch engine issued speculatively in idle batch slots (the steps a
 * backtracking search asks for after a rejected trial, evaluated in the same matrix passes as the trial) and how
 * many of them it adopted, summed over the context's life.  BIOEN_HIP_SPECULATE=0 switches the mechanism off;
 * results are the same to the last bit either way. */
int bioen_hip_speculation_stats(bioen_hip_ctx* ctx, long long* issued, long long* adopted);

/* Diagnostic builds only (-DSTRIP_DIAG=4, tools/strip_probe.py): per-phase cycle sums of the last forces strip
 * launch, out[nblocks][16 waves][8 phases]; a normal build leaves the buffer untouched. */
int bioen_hip_debug_strip_stamps(bioen_hip_ctx* ctx, int enable, long long* out, int nblocks);

/* Measurement aid (tools/pass_probe.py): the two matrix passes of a log-weights evaluation alone --
 * `reps` launches each at batch width k (1..8) over whatever the problem slots hold -- mean milliseconds per launch,
 * HIP events on the context's stream.  Produces and changes no result. */
int bioen_hip_debug_pass_probe(bioen_hip_ctx* ctx, int k, int reps, double* fwd_ms, double* adj_ms);

/* ---- host self-test of the L-BFGS driver (no GPU needed) ------------------------------
 * Runs the SAME driver + line-search code as the optimizers above on a built-in analytic
 * objective evaluated on the host: kind 0 = extended Rosenbrock, kind 1 = ill-conditioned
 * convex quadratic + quartic.  Lets the CPU test-suite pin the control flow
 * (lbfgs.c:245-641, 645-734, 812-1296) without a device. */
int bioen_hip_selftest_lbfgs(int kind, int n, const double* x0, const bioen_lbfgs_config* config,
                             double* x_out, bioen_opt_result* info);

/* Forces method on structure-sharded contexts: bioen_hip_forces_weights (one all-gather), bioen_hip_forces_fdf and the
 * forces optimizers (two all-gathers per evaluation) -- the two strip passes for M <= 1024, the four
 * passes over row panels beyond (r05: both in canonical segments, the bits of the single-GPU run).  Only the r01
 * streaming kernels (no strip copies: BIOEN_HIP_PANELS=0, BIOEN_HIP_STRIP_TALL=0, or a device without the memory for
 * them) remain unsharded-only: BIOEN_HIP_ESTATE there. */

/* ---- yTilde assembled on the device from raw observables ---------------------------------
 * Replaces the host loops of bioen/analyze/observables/observables.py:110-143 (sim / sigma built
 * element by element, structure by structure) and the host division: yTilde_ij = sim_ij / err_i,
 * YTilde_i = exp_i / err_i, divided on the device (true division: bitwise what numpy gives).
 *   structure_major = 0: sim is [m][n] (observables-major, like yTilde);
 *   structure_major = 1: sim is [n][m] -- each structure's (model's) M observables contiguous, the
 *     order in which simulated data arrive; uploaded in chunks and transposed on the device. */
int bioen_hip_ctx_create_raw(int m, long long n, int structure_major, const double* sim,
                             const double* exp_values, const double* exp_err, int device,
                             bioen_hip_ctx** ctx);

/* ---- GSL-style minimizers on the device objective -----------------------------------------
 * Replace _opt_bfgs_logw (c_bioen_kernels_logw.c:366-509) and _opt_bfgs_forces
 * (c_bioen_kernels_forces.c), i.e. gsl_multimin_fdfminimizer_{conjugate_fr, conjugate_pr,
 * vector_bfgs2, vector_bfgs, steepest_descent} (GSL 2.5) driven by the reference's loop with its
 * max-norm gradient test (c_bioen_common.c:112-138).  GSL is not linked: the five algorithms are
 * restated in bioen_amd/csrc/multimin.hpp.  Log-weights: variables, gradient and all work vectors
 * stay in HBM; forces: the M variables stay on the host.
 * info->lbfgs_code carries the GSL status: 0 success, -2 GSL_CONTINUE (iteration budget used),
 * 27 GSL_ENOPROG, 13 GSL_EBADTOL -- the reference treats {0, -2, 27} as success
 * (c_bioen.pyx:109-116).  info->iterations = driver iterations, info->evaluations = f + gradient
 * evaluations.  Sharded contexts (r05): served -- the minimizers' inner products and norms are sums over structures
 * like every other one (per canonical segment, one stage all-gather each), so every rank takes the same steps. */
const char* bioen_hip_gsl_strerror(int gsl_code);   /* replaces bioen_gsl_error(), c_bioen_error.c:14-20 */
int bioen_hip_opt_gsl_logw(bioen_hip_ctx* ctx, const double* g0, const double* G, double theta,
                           const bioen_gsl_config* config, const bioen_visual_params* visual,
                           double* result, double* w_opt, bioen_opt_result* info);
int bioen_hip_opt_gsl_forces(bioen_hip_ctx* ctx, const double* forces0, const double* w0, double theta,
                             const bioen_gsl_config* config, const bioen_visual_params* visual,
                             double* result, double* w_opt, bioen_opt_result* info);
/* GSL's own multimin test programme (multimin/test.c:106-160, test_funcs.c) on the SAME minimizer
 * code with host vectors, no GPU needed: kind 0 Roth, 1 Wood, 2 Rosenbrock, 3 SimpleAbs.
 * info->lbfgs_code = last status, info->reserved = gradient evaluations. */
int bioen_hip_selftest_multimin(int algorithm, int kind, const double* x0, double* x_out,
                                bioen_opt_result* info);

/* The same minimizer code on a host objective the CALLER supplies (host vectors, no GPU): run under the
 * reference's driver loop (c_bioen_kernels_logw.c:434-464) exactly as bioen_hip_opt_gsl_* are.
 * objective(user, x, &f, grad): grad == NULL asks for f alone (GSL's `f` callback), otherwise f and
 * the gradient (`df` / `fdf`); a non-zero return aborts the run.  The tests drive this with the
 * reference's own C objective to pin the minimizers against real GSL runs bit for bit. */
typedef int (*bioen_host_objective)(void* user, const double* x, double* f, double* grad);
int bioen_hip_multimin_host(int n, bioen_host_objective objective, void* user, const double* x0,
                            const bioen_gsl_config* config, double* x_out, bioen_opt_result* info);

/* ---- theta-sweep gather over RCCL (multi-GPU; one process per GPU) ----------------- */
/* rank 0 obtains the 128-byte ncclUniqueId; the host side ships it to the other ranks */
int bioen_hip_comm_unique_id(unsigned char id[128]);
int bioen_hip_comm_init(bioen_hip_ctx* ctx, const unsigned char id[128], int rank, int nranks);
/* all-gather of `count` doubles per rank, host in / host out (staged through HBM, RCCL over xGMI) */
int bioen_hip_comm_allgather(bioen_hip_ctx* ctx, const double* send, size_t count, double* recv);
/* != 0: a bioen_hip_comm_init was given up at its time bound (a rank never called it) and its helper thread is still
 * blocked inside ncclCommInitRank.  The process goes on without RCCL; it should leave through _exit (after flushing its
 * output) rather than through normal teardown, where the collective library's static destructors would run under that
 * thread.  Reset if the late peer shows up (the orphaned communicator is then aborted). */
int bioen_hip_comm_init_abandoned(void);
/* average wall time (microseconds) of one stage exchange of `count` doubles per rank on a sharded
 * context, back to back on the context's stream: lets the host decide whether splitting the
 * structures beats dealing thetas for a given problem size */
int bioen_hip_exchange_probe(bioen_hip_ctx* ctx, size_t count, int reps, double* usec_per_exchange);

/* Measurement aid (bench.py: roofline.read_ceiling): streams the resident form of the matrix `reps` times with a plain
 * read-only kernel (wide nontemporal loads, no LDS, no matrix cores) and reports the rate -- what this box's memory
 * system delivers to a read stream of exactly the bytes the matrix passes read.  form: 0 = whichever form is resident
 * (strip copy first), 1 / 2 / 4 = the form of that bioen_hip_ctx_footprint bit.  No reference counterpart. */
int bioen_hip_read_probe(bioen_hip_ctx* ctx, int form, int reps, double* gbytes_per_s, long long* bytes);
int bioen_hip_comm_destroy(bioen_hip_ctx* ctx);

/* ---- peer-to-peer stage exchange (r04): a third transport for the stage all-gathers of structure-sharded contexts,
 *      beside RCCL (bioen_hip_comm_init) and the host callback.  Every rank owns a mailbox in its HBM; the peers map
 *      it through hipIpc (xGMI between GPUs; between processes sharing ONE GPU too, which is how the single-GPU test
 *      box runs it) and an exchange is one small kernel on the context's stream: stores of this rank's segment into
 *      every peer's mailbox, a system-scope flag per peer, a bounded wait for the peers' flags -- no collective launch
 *      and no host.  No reference counterpart (bioen/analyze/procedure.py:62-63 is a serial loop in one process).
 *        1. every rank: bioen_hip_p2p_export(ctx, handle)           -- allocates the mailbox, returns its 64-byte hipIpc handle
 *        2. the host side all-gathers the handles (bioen_amd.sweep: SocketComm)
 *        3. every rank: bioen_hip_p2p_attach(ctx, handles[world][64]) -- maps the peers; from now on the exchanges of this
 *           context use this transport (it takes precedence over a communicator / callback that is also set)
 *      bioen_hip_p2p_detach unmaps and frees (also done by bioen_hip_ctx_destroy).  world = 1 with
 *      bioen_hip_ctx_set_force_exchange: attach(ctx, NULL) runs the (empty) exchange kernels all the same. */
int bioen_hip_p2p_export(bioen_hip_ctx* ctx, unsigned char handle[64]);
int bioen_hip_p2p_attach(bioen_hip_ctx* ctx, const unsigned char* handles);
int bioen_hip_p2p_detach(bioen_hip_ctx* ctx);
/* which transport the next stage exchange would use: 0 none (unsharded), 1 RCCL, 2 host callback, 3 peer-to-peer */
int bioen_hip_exchange_transport(const bioen_hip_ctx* ctx);
/* Self-test of whichever transport is active (sharded context, or world = 1 with forced exchanges): `reps` stage
 * exchanges of varying size queued back to back, each rank's segment a pattern of (rank, exchange, index), every segment
 * checked on the device after each exchange.  *mismatches = wrong doubles seen (0 = the transport delivers). */
int bioen_hip_exchange_selftest(bioen_hip_ctx* ctx, int reps, long long* mismatches);
int bioen_hip_exchange_counts3(const bioen_hip_ctx* ctx, long long* rccl, long long* host_staged, long long* p2p);
/* Every wait of the library on a round's results or on a peer is bounded: after `seconds` (default 60, environment
 * BIOEN_HIP_WAIT_TIMEOUT) without the awaited word the call returns BIOEN_HIP_ERCCL (a context that exchanges: a peer
 * is gone; an RCCL communicator is aborted with ncclCommAbort so that its kernel returns) or BIOEN_HIP_EHIP, with
 * bioen_hip_last_error() naming what was awaited.  The context is unusable afterwards (BIOEN_HIP_ESTATE): destroy it. */
int bioen_hip_ctx_set_wait_timeout(bioen_hip_ctx* ctx, double seconds);

#ifdef __cplusplus
}
#endif
#endif /* BIOEN_HIP_H */
