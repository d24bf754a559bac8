// extern "C" entry points of libbioen_hip.so (declared in include/bioen_hip.h) and the
// two L-BFGS backends that sit on the kernels of kernels_*.hip.
#include <dlfcn.h>
#include <sys/mman.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <new>
#include <atomic>
#include <memory>
#include <string>
#include <thread>

#include "ctx.hpp"
#include "kernels.hpp"
#include "lbfgs.hpp"
#include "multimin.hpp"

static_assert(bioen::kHistory == bioen::kLbfgsM, "history length");

namespace bioen {

static thread_local std::string g_last_error;
static int g_fast_openmp_flag = 0;

void set_last_error(const std::string& s) { g_last_error = s; }

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_last_error = buf;
    return BIOEN_HIP_EHIP;
}

static int fail(int code, const char* msg) {
    g_last_error = msg;
    return code;
}

// ---------------------------------------------------------------------------------
// allocation helpers
// ---------------------------------------------------------------------------------
static int dalloc_zero(double** p, size_t count, hipStream_t s) {
    *p = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(double));
    if (e != hipSuccess) {
        hip_fail(e, "hipMalloc", __FILE__, __LINE__);
        return BIOEN_HIP_ENOMEM;
    }
    BIOEN_HIP_CHECK(hipMemsetAsync(*p, 0, count * sizeof(double), s));
    return 0;
}

// (per SEGMENT: the tiles are those of a GPU holding one segment; a context with vr of them has vr x as many)
static void choose_fwd_tiling(bioen_hip_ctx* c) {
    const int total_steps = c->segcols / 128;
    const int row_blocks = c->mp / kRowAlign;
    // Tiles of >= 42 steps (a block ends with a 63-shuffle reduction and K x 32 scattered stores),
    // between 512 and ~6144 blocks.  N = 1e6 x M = 1024: 5952 blocks (3072 and 2048 measured within
    // +-1.5 % of it); N = 1.25e5 (one rank of an 8-way sharded run): 736 blocks, 165 us per launch
    // against 179 us with 6-step tiles.
    const int min_tiles = (512 + row_blocks - 1) / row_blocks, max_tiles = (6144 + row_blocks - 1) / row_blocks;
    int want_tiles = std::max(min_tiles, std::min(max_tiles, total_steps / 42));
    want_tiles = std::max(1, std::min(want_tiles, total_steps));
    int spt = (total_steps + want_tiles - 1) / want_tiles;
    if (spt & 1) ++spt;   // two 1-KiB steps in flight per row
    c->fwd_steps = spt;
    c->fwd_ctiles = (total_steps + spt - 1) / spt * c->vr;
}

static int alloc_slot(bioen_hip_ctx* c, int s, bool with_history, bool with_spare = false) {
    ProblemSlot& sl = c->slot[s];
    int rc = 0;
    if (!sl.allocated) {
        double** vecs[] = {&sl.xa, &sl.xb, &sl.ga, &sl.gb, &sl.d, &sl.w, &sl.a};
        for (double** p : vecs)
            if ((rc = dalloc_zero(p, c->ld, c->stream)) != 0) return rc;
        sl.x = sl.xa;
        sl.xp = sl.xb;
        sl.g = sl.ga;
        sl.gp = sl.gb;
        sl.scal = c->scal + (size_t)s * kScalStride;
        sl.gram = c->gram + (size_t)s * kGramStride;
        sl.part = c->part + (size_t)s * P_COUNT * kPartStride;
        sl.allocated = true;
    }
    if (with_history && !sl.history) {
        for (int i = 0; i < kHistory; ++i) {
            if ((rc = dalloc_zero(&sl.S[i], c->ld, c->stream)) != 0) return rc;
            if ((rc = dalloc_zero(&sl.Yh[i], c->ld, c->stream)) != 0) return rc;
        }
        sl.history = true;
    }
    if (with_spare && !sl.Ssp) {        // the pending (s, y) pair of the device-resident engine
        if ((rc = dalloc_zero(&sl.Ssp, c->ld, c->stream)) != 0) return rc;
        if ((rc = dalloc_zero(&sl.Ysp, c->ld, c->stream)) != 0) return rc;
    }
    return 0;
}

// Canonical segments (ctx.hpp): nseg = 8 whenever `world` divides 8, else world; segcols = ceil(n / nseg) rounded up
// to 128; rank r holds the vr = nseg / world segments [r vr, (r + 1) vr), i.e. the columns [r vr segcols, ...).
// BIOEN_HIP_SEGMENTS=1 (r06, unsharded contexts only): ONE segment -- the opt-out from the canonical shape for a caller
// who will never compare with a run on another number of GPUs: every sum over structures is then one tree over the whole
// matrix, the forces passes write one partial set per block instead of eight (3-7 % of a pass, profiles/
// r05_forces_ab_vs_r04.txt), the N-vector kernels one total instead of eight.  The results are those of ONE GPU only.
static void segment_geometry(long long n_global, int world, int* nseg, int* vr, long long* segcols) {
    *nseg = (world <= kMaxSeg && kMaxSeg % world == 0) ? kMaxSeg : world;
    if (world == 1) {
        const char* e = std::getenv("BIOEN_HIP_SEGMENTS");
        if (e && e[0] == '1' && e[1] == 0) *nseg = 1;
    }
    *vr = *nseg / world;
    *segcols = (long long)round_up((size_t)((n_global + *nseg - 1) / *nseg), kColAlign);
}

static int ctx_alloc(int m, long long n_global, int device, int rank, int world, bioen_hip_ctx** out) {
    if (!out) return fail(BIOEN_HIP_EINVAL, "ctx pointer is NULL");
    *out = nullptr;
    if (m <= 0 || n_global <= 0) return fail(BIOEN_HIP_EINVAL, "m and n must be positive");
    if (world < 1 || world > kMaxPartials / 8 || rank < 0 || rank >= world)
        return fail(BIOEN_HIP_EINVAL, "bad rank / world");
    int nseg = 0, vr = 0;
    long long segcols = 0;
    segment_geometry(n_global, world, &nseg, &vr, &segcols);
    const long long per = segcols * vr;
    const long long col0 = per * rank;
    long long n_local = std::max(0ll, std::min(per, n_global - col0));
    // rank-independent test (every rank must fail together, else the others hang in the first collective):
    // the LAST rank still has to own at least one column
    if (world > 1 && (long long)(world - 1) * per >= n_global)
        return fail(BIOEN_HIP_EINVAL, "too few structures to shard: the columns go to the ranks in runs of (8 / world) segments of "
                                      "round_up(ceil(n / 8), 128) columns (world dividing 8; else one segment of round_up(ceil(n / "
                                      "world), 128) per rank), and the last rank would be empty");
    if (per > 0x7fffffff) return fail(BIOEN_HIP_EINVAL, "n per GPU exceeds 2^31-1");
    const int n = (int)n_local;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(BIOEN_HIP_ENODEV, "no HIP device visible (libbioen_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(BIOEN_HIP_EINVAL, "device index out of range");
    BIOEN_HIP_CHECK(hipSetDevice(device));

    bioen_hip_ctx* c = new (std::nothrow) bioen_hip_ctx();
    if (!c) return fail(BIOEN_HIP_ENOMEM, "host allocation failed");
    c->device = device;
    c->m = m;
    c->n = n;
    c->rank = rank;
    c->world = world;
    c->n_global = n_global;
    c->col0 = col0;
    c->mp = (int)round_up((size_t)m, kRowAlign);
    c->nseg = nseg;
    c->vr = vr;
    c->seg0 = rank * vr;
    c->segcols = (int)segcols;
    c->ld = (size_t)per;                 // vr segments of segcols columns: identical on every rank
    choose_fwd_tiling(c);
    // stream yTilde with non-temporal loads once it no longer fits the 256 MiB Infinity Cache
    c->nontemporal = (size_t)c->mp * c->ld * sizeof(double) > (size_t)192 * 1024 * 1024;
    if (const char* e = std::getenv("BIOEN_HIP_NVEC_NT")) c->nvec_nt_env = e[0] == '1' ? 1 : 0;
    c->nvec_nt = c->nvec_nt_env == 1;
    if (const char* e = std::getenv("BIOEN_HIP_FORCE_EXCHANGE")) c->force_exchange = (e[0] == '1') ? 1 : 0;
    if (const char* e = std::getenv("BIOEN_HIP_WAIT_TIMEOUT")) {      // seconds; bound of every wait on a round or an exchange
        const double v = std::atof(e);
        if (v > 0.0) c->wait_timeout_s = v;
    }
    {
        const char* e = std::getenv("BIOEN_HIP_KEEP_ROWMAJOR");    // A/B: keep the row-major matrix beside the strip copies
        c->keep_rowmajor = (e && e[0] == '1') ? 1 : 0;
        e = std::getenv("BIOEN_HIP_ONE_COPY");         // log-weights on ONE strip copy (ctx.hpp: one_copy)
        c->one_copy_wanted = (e && e[0] == '1') ? 1 : (e && e[0] == '0') ? 0 : -1;     // (-1: by size, kernels_strip.hip: one_copy_by_default)
        e = std::getenv("BIOEN_HIP_FWD_STREAM");       // A/B: the streaming forward kernel on the row-major matrix
        c->fwd_stream = (e && e[0] == '1') ? 1 : 0;
        e = std::getenv("BIOEN_HIP_PANELS");           // A/B: M > 1024 on the r01 streaming kernels instead of row panels
        c->panel_off = (e && e[0] == '0') ? 1 : 0;
    }

    int rc = 0;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__);
    }
#define TRY(x) if ((rc = (x)) != 0) { bioen_hip_ctx_destroy(c); return rc; }
    TRY(dalloc_zero(&c->Y, (size_t)c->mp * c->ld, c->stream));
    TRY(dalloc_zero(&c->YT, c->mp, c->stream));
    TRY(dalloc_zero(&c->row_offset, c->mp, c->stream));
    TRY(dalloc_zero(&c->row_scale, c->mp, c->stream));
    {
        std::vector<double> ones(c->mp, 1.0);
        hipError_t e1 = hipMemcpyAsync(c->row_scale, ones.data(), (size_t)c->mp * sizeof(double), hipMemcpyHostToDevice,
                                       c->stream);
        if (e1 == hipSuccess) e1 = hipStreamSynchronize(c->stream);
        if (e1 != hipSuccess) {
            bioen_hip_ctx_destroy(c);
            return hip_fail(e1, "row_scale init", __FILE__, __LINE__);
        }
    }
    TRY(dalloc_zero(&c->ybar_c, (size_t)c->mp * kMaxBatch, c->stream));
    TRY(dalloc_zero(&c->r_c, (size_t)c->mp * kMaxBatch, c->stream));
    TRY(dalloc_zero(&c->um, (size_t)c->mp * kMaxBatch, c->stream));
    TRY(dalloc_zero(&c->gm, (size_t)c->mp * kMaxBatch, c->stream));
    TRY(dalloc_zero(&c->fixed, c->ld, c->stream));
    TRY(dalloc_zero(&c->t, c->ld, c->stream));
    // per (row, problem) one partial per column tile of the forward pass, or per block of the fused
    // forces pass (forces_fused_blocks)
    {
        // (the sets the strip passes leave on this context: kernels.hpp: StripSets; an unfolded forward pass -- sharded
        // contexts, BIOEN_HIP_STRIP_FOLD=0 -- leaves nch per group)
        const StripSets fs = strip_sets(c), gs = forces_sets(c);
        const long long sets = std::max<long long>((long long)fs.gs * fs.nch * c->vr, (long long)gs.sets * c->vr);
        if (sets > kPartStride) { bioen_hip_ctx_destroy(c); return fail(BIOEN_HIP_EINVAL, "strip sets exceed kPartStride"); }
        TRY(dalloc_zero(&c->fwd_partial, (size_t)kMaxBatch * c->mp * std::max<long long>(std::max(c->fwd_ctiles, kFusedBlocks), sets), c->stream));
    }
    TRY(dalloc_zero(&c->part, (size_t)kMaxBatch * P_COUNT * kPartStride, c->stream));
    TRY(dalloc_zero(&c->scal, (size_t)kMaxBatch * kScalStride, c->stream));
    TRY(dalloc_zero(&c->gram, (size_t)kMaxBatch * kGramStride, c->stream));
    {   // exchange stages: [nseg][capacity per segment] = [world][capacity per rank]
        const size_t npl = (size_t)vec_grid(c);
        const size_t arrays[X_COUNT] = {1, 3, 0, 3, 2, 1, 1, 1, 0, kGramDots, 0};
        for (int st = 0; st < X_COUNT; ++st) {
            size_t cap = arrays[st] * kMaxBatch * npl;
            if (st == X_YBAR) cap = (size_t)(c->mp + 3) * kMaxBatch;
            if (st == X_GRAMR) cap = (size_t)kDevRankSums * kMaxBatch;
            if (st == X_VEC) cap = world > 1 ? (size_t)c->segcols : 0;
            c->xcap[st] = cap * vr;
            if (cap) TRY(dalloc_zero(&c->xbuf[st], cap * nseg, c->stream));
        }
    }
    TRY(alloc_slot(c, 0, false));
#undef TRY
    e = hipHostMalloc(reinterpret_cast<void**>(&c->host_scal), (size_t)kMaxBatch * kScalStride * sizeof(double),
                      hipHostMallocDefault);
    if (e != hipSuccess) {
        bioen_hip_ctx_destroy(c);
        return hip_fail(e, "hipHostMalloc", __FILE__, __LINE__);
    }
    e = hipHostMalloc(reinterpret_cast<void**>(&c->live), (size_t)kMaxBatch * (kScalStride + 1) * sizeof(double),
                      hipHostMallocCoherent | hipHostMallocMapped);
    if (e == hipSuccess) {
        std::memset(c->live, 0, (size_t)kMaxBatch * (kScalStride + 1) * sizeof(double));
    } else {                      // no coherent host memory on this system: copy + synchronise per round instead
        (void)hipGetLastError();
        c->live = nullptr;
        c->live_off = 1;
    }
    if (const char* v = std::getenv("BIOEN_HIP_LIVE")) c->live_off = c->live_off || (v[0] == '0');
    *out = c;
    return 0;
}

// Host-to-device copy of a CALLER's (pageable) buffer.  r05, ROCm 7.2: the runtime pins a pageable source of this size for
// the transfer, and refuses -- "invalid argument", from the asynchronous and the plain copy alike -- a buffer that took
// (part of) the address range of one it has pinned before: a fresh numpy vector on the heap range of a freed one, in a
// process that has already closed a context (tools/attic/onecopy_probe.py ran into it four times out of four).  The
// transfer then goes through a pinned buffer of our own, 8 MB at a time: nothing of the caller's is pinned.
static int h2d_staged(bioen_hip_ctx* c, char* dst, size_t dpitch, const char* src, size_t spitch, size_t width, size_t height) {
    constexpr size_t kChunk = (size_t)8 << 20;
    if (!c->stage_host) {
        void* p = nullptr;
        BIOEN_HIP_CHECK(hipHostMalloc(&p, kChunk, hipHostMallocDefault));
        c->stage_host = static_cast<char*>(p);
    }
    for (size_t row = 0; row < height; ++row)
        for (size_t off = 0; off < width; off += kChunk) {
            const size_t nb = std::min(kChunk, width - off);
            std::memcpy(c->stage_host, src + row * spitch + off, nb);
            BIOEN_HIP_CHECK(hipMemcpyAsync(dst + row * dpitch + off, c->stage_host, nb, hipMemcpyHostToDevice, c->stream));
            BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));      // the one buffer is written again
        }
    return 0;
}

// tests: BIOEN_HIP_TEST_STAGED_UPLOAD=1 sends every such copy -- uploads and downloads of caller buffers -- through the
// staging buffers (the refusal itself cannot be provoked at will)
static bool staged_upload_forced() {
    const char* e = std::getenv("BIOEN_HIP_TEST_STAGED_UPLOAD");
    return e && e[0] == '1';
}

static int h2d_user(bioen_hip_ctx* c, void* dst, const void* src, size_t bytes) {
    const hipError_t e = staged_upload_forced() ? hipErrorInvalidValue
                                                : hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipErrorInvalidValue) {
        (void)hipGetLastError();
        return h2d_staged(c, static_cast<char*>(dst), bytes, static_cast<const char*>(src), bytes, bytes, 1);
    }
    BIOEN_HIP_CHECK(e);
    return 0;
}

// The same towards a caller's buffer: device -> host.  The staged form is synchronous on `stream` and brings its own pinned
// chunk (the result deliveries run on threads of their own: nothing shared).
static hipError_t d2h_user(hipStream_t stream, void* dst, const void* src, size_t bytes) {
    hipError_t e = staged_upload_forced() ? hipErrorInvalidValue : hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream);
    if (e != hipErrorInvalidValue) return e;
    (void)hipGetLastError();
    constexpr size_t kChunk = (size_t)8 << 20;
    void* stage = nullptr;
    e = hipHostMalloc(&stage, std::min(kChunk, std::max<size_t>(bytes, 1)), hipHostMallocDefault);
    for (size_t off = 0; e == hipSuccess && off < bytes; off += kChunk) {
        const size_t nb = std::min(kChunk, bytes - off);
        e = hipMemcpyAsync(stage, static_cast<const char*>(src) + off, nb, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e == hipSuccess) std::memcpy(static_cast<char*>(dst) + off, stage, nb);
    }
    if (stage) (void)hipHostFree(stage);
    return e;
}

static hipError_t d2h_user_2d(hipStream_t stream, char* dst, size_t dpitch, const char* src, size_t spitch, size_t width, size_t height) {
    hipError_t e = staged_upload_forced() ? hipErrorInvalidValue
                                          : hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, stream);
    if (e != hipErrorInvalidValue) return e;
    (void)hipGetLastError();
    e = hipSuccess;
    for (size_t row = 0; row < height && e == hipSuccess; ++row) {
        e = d2h_user(stream, dst + row * dpitch, src + row * spitch, width);      // row by row: each tries the direct copy first
    }
    return e;
}

// host N-vectors are always GLOBAL (n_global long); a sharded context takes its slice
static int upload_n(bioen_hip_ctx* c, double* dst, const double* src) {
    return h2d_user(c, dst, src + c->col0, (size_t)c->n * sizeof(double));
}

static int rccl_allgather_inplace(bioen_hip_ctx* c, double* base, size_t count);   // defined with the RCCL glue
static int rccl_async_error(bioen_hip_ctx* c);      // != 0: the communicator reports a failure (last_error set)
static void rccl_abort(bioen_hip_ctx* c);           // ncclCommAbort: kernels of a collective a dead peer left hanging return

// What went wrong asynchronously on this context's transports since the last look: the error word of the peer-to-peer
// exchange kernels (kernels_p2p.hip) and the RCCL communicator's own.  0 = nothing.
static int transport_error(bioen_hip_ctx* c) {
    if (c->p2p_err) {
        const unsigned long long w = __atomic_load_n(c->p2p_err, __ATOMIC_ACQUIRE);
        if (w) {
            static const char* const why[] = {"?", "timed out waiting for", "received an ABORT flag from", "an earlier exchange failed; peer",
                                              "is OUT OF STEP (another stage or payload under this exchange number) with"};
            char buf[256];
            std::snprintf(buf, sizeof buf, "peer-to-peer exchange %llu (stage %d): %s rank %d (BIOEN_HIP_WAIT_TIMEOUT = %g s)",
                          w & 0xffffffffffull, (int)((w >> 52) & 0xff), why[std::min<unsigned long long>(4, w >> 60)],
                          (int)((w >> 40) & 0xfff), c->wait_timeout_s);
            if (!c->failed) c->failed_p2p = 1;
            c->failed = 1;
            if (c->fail_msg.empty()) c->fail_msg = buf;
            return fail(BIOEN_HIP_ERCCL, c->fail_msg.c_str());
        }
    }
    if (c->comm) {
        const int rc = rccl_async_error(c);
        if (rc) {
            c->failed = 1;
            if (c->fail_msg.empty()) c->fail_msg = g_last_error;
            return rc;
        }
    }
    return 0;
}

// A bounded host wait on a word a kernel publishes (the live pages of the engines).  tick() is called once per look at
// the word: it spins politely (pause, later yield), and every 4096 looks asks the stream (a failed launch ends the wait
// with its error), the transports (above) and the clock: past c->wait_timeout_s the wait fails with BIOEN_HIP_ERCCL
// on a context that exchanges (a peer is gone: the communicator is aborted so that its kernel returns) and BIOEN_HIP_EHIP
// otherwise -- never a hang.
struct BoundedWait {
    bioen_hip_ctx* c;
    const char* what;
    unsigned spins = 0;
    bool armed = false;
    std::chrono::steady_clock::time_point deadline;
    BoundedWait(bioen_hip_ctx* ctx, const char* w) : c(ctx), what(w) {}
    // published(): re-reads the word; -> 0 keep waiting, 1 it is there, < 0 failure
    template <class Pred>
    int tick(Pred published) {
        ++spins;
        if ((spins & 0xfffu) == 0) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) {
                if (published()) return 1;
                const int te = transport_error(c);
                if (te) return te;
                return fail(BIOEN_HIP_ESTATE, "round finished without publishing its results");
            }
            if (q != hipErrorNotReady) return hip_fail(q, "hipStreamQuery", __FILE__, __LINE__);
            const int te = transport_error(c);
            if (te) return te;
            const auto now = std::chrono::steady_clock::now();
            if (!armed) {
                armed = true;
                deadline = now + std::chrono::duration_cast<std::chrono::steady_clock::duration>(
                                     std::chrono::duration<double>(c->wait_timeout_s + (c->p2p_on ? 2.0 : 0.0)));
            } else if (now > deadline) {
                char buf[200];
                std::snprintf(buf, sizeof buf, "timed out after %g s waiting for %s (BIOEN_HIP_WAIT_TIMEOUT)",
                              c->wait_timeout_s, what);
                c->failed = 1;
                if (c->fail_msg.empty()) c->fail_msg = buf;
                const bool exchanging = c->world > 1 || c->comm || c->exchange_cb || c->p2p_on;
                if (c->comm) rccl_abort(c);
                return fail(exchanging ? BIOEN_HIP_ERCCL : BIOEN_HIP_EHIP, buf);
            }
        }
        if (spins < 20000u) {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
        } else {
            std::this_thread::yield();
        }
        return 0;
    }
};

// One in-place all-gather of an exchange stage ([world][payload] doubles).  world == 1: nothing.  Three transports, in
// this order of precedence: the peer-to-peer mailboxes (kernels_p2p.hip: one kernel, no collective launch, no host), RCCL
// (ncclAllGather on the context's stream), the host callback (D2H, the caller's all-gather, H2D: processes that share
// neither).  All three leave the same bytes in the stage buffer.
// world == 1 with force_exchange (bioen_hip_ctx_set_force_exchange / BIOEN_HIP_FORCE_EXCHANGE=1) and a communicator
// or callback in place: the one-rank all-gather is executed all the same -- a copy of the rank's segment onto
// itself, same bits -- so the stage path can run under test on a single GPU.
static bool exchanges_forced(const bioen_hip_ctx* c) {
    return c->world == 1 && c->force_exchange && (c->comm || c->exchange_cb || c->p2p_on);
}

// `seg_payload` = doubles per SEGMENT (what the kernels' stage views are built with); a rank ships its vr segments
static int exchange_raw(bioen_hip_ctx* c, int stage, size_t payload);
static int exchange(bioen_hip_ctx* c, int stage, size_t seg_payload) {
    return exchange_raw(c, stage, seg_payload * (size_t)c->vr);
}
static int exchange_raw(bioen_hip_ctx* c, int stage, size_t payload) {      // payload = doubles per RANK
    if (c->world == 1 && !exchanges_forced(c)) return 0;
    if (c->failed) {
        const std::string m = "this context failed earlier and must be destroyed: " + c->fail_msg;
        return fail(BIOEN_HIP_ESTATE, m.c_str());
    }
    double* base = c->xbuf[stage];
    if (c->mirror_exchange) {      // measurement aid: this rank's part over every other rank's (one kernel)
        launch_xch_mirror(c, stage, payload);
        return 0;
    }
    if (c->p2p_on) {          // stores into the peers' mailboxes + flags, one kernel (kernels_p2p.hip)
        if (payload > c->p2p_cap) return fail(BIOEN_HIP_EINVAL, "exchange payload exceeds the mailbox slot");
        ++c->n_p2p_exchanges;
        launch_p2p_exchange(c, stage, payload);
        return 0;
    }
    if (c->comm) {
        ++c->n_rccl_exchanges;
        return rccl_allgather_inplace(c, base, payload);
    }
    if (!c->exchange_cb) return fail(BIOEN_HIP_ESTATE, "sharded context without a communicator");
    ++c->n_host_exchanges;
    // host-staged path (processes that cannot share an RCCL communicator, e.g. tests on one GPU)
    const size_t total = payload * c->world;
    if (c->exchange_host_count < total) {
        if (c->exchange_host) hipHostFree(c->exchange_host);
        c->exchange_host = nullptr;
        BIOEN_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&c->exchange_host), total * sizeof(double),
                                      hipHostMallocDefault));
        c->exchange_host_count = total;
    }
    BIOEN_HIP_CHECK(hipMemcpyAsync(c->exchange_host + (size_t)c->rank * payload, base + (size_t)c->rank * payload,
                                   payload * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->exchange_cb(c->exchange_user, c->exchange_host, payload) != 0) {
        c->exchange_error = 1;
        c->failed = 1;
        if (c->fail_msg.empty()) c->fail_msg = "the host-staged exchange callback failed (a peer gone, or its transport timed out)";
        return fail(BIOEN_HIP_ERCCL, c->fail_msg.c_str());
    }
    BIOEN_HIP_CHECK(hipMemcpyAsync(base, c->exchange_host, total * sizeof(double), hipMemcpyHostToDevice, c->stream));
    return 0;
}

// gather a sharded device N-vector into a GLOBAL host vector (world == 1: plain download)
static int download_n(bioen_hip_ctx* c, double* dst_global, const double* src_local) {
    if (c->world == 1) {
        BIOEN_HIP_CHECK(d2h_user(c->stream, dst_global, src_local, (size_t)c->n * sizeof(double)));
        return 0;
    }
    double* base = c->xbuf[X_VEC];
    BIOEN_HIP_CHECK(hipMemcpyAsync(base + (size_t)c->rank * c->ld, src_local, c->ld * sizeof(double),
                                   hipMemcpyDeviceToDevice, c->stream));
    int rc = exchange_raw(c, X_VEC, c->ld);
    if (rc) return rc;
    for (int r = 0; r < c->world; ++r) {
        long long col0, nl;
        rank_columns(c, r, &col0, &nl);
        if (nl > 0)
            BIOEN_HIP_CHECK(d2h_user(c->stream, dst_global + col0, base + (size_t)r * c->ld, (size_t)nl * sizeof(double)));
    }
    return 0;
}

static int read_scalars(bioen_hip_ctx* c, int nslots = 1) {
    BIOEN_HIP_CHECK(hipMemcpyAsync(c->host_scal, c->scal, (size_t)nslots * kScalStride * sizeof(double),
                                   hipMemcpyDeviceToHost, c->stream));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    return transport_error(c);      // scalars behind a failed exchange are not results
}

// The scalars of a round whose last kernel published them live (k_finish_eval): wait for the n flags to reach the
// round number, then mirror the slots where read_scalars would have put them.  The stream is polled now and then,
// so that a failed launch ends the wait with its error instead of hanging the caller.
static int await_live(bioen_hip_ctx* c, unsigned long long round, const int* slots, int n) {
    const volatile unsigned long long* flag =
        reinterpret_cast<const volatile unsigned long long*>(c->live + (size_t)kMaxBatch * kScalStride);
    BoundedWait w(c, "a round's scalars (log-weights engine)");
    for (int a = 0; a < n; ++a) {
        while (flag[a] != round) {
            const int t = w.tick([&] { return flag[a] == round; });
            if (t < 0) return t;
            if (t > 0) break;
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    // the flags are there -- but a round whose exchange kernel gave up (ABORT from a peer, OUT OF STEP, a deadline) still
    // publishes them, from stale stage buffers: scalars behind a failed exchange are not results (one host-mapped word)
    if (const int te = transport_error(c)) return te;
    for (int a = 0; a < n; ++a)
        std::memcpy(c->host_scal + (size_t)slots[a] * kScalStride, c->live + (size_t)a * kScalStride,
                    kScalStride * sizeof(double));
    return 0;
}

// log sum exp(G) of the prior into scal[S_LOGS0] of the round's problems (A2, c_bioen_kernels_logw.c:29-53, hoisted:
// once per run).  Sharded contexts (r04): every rank leaves the {max, sum} pairs of its blocks in its segment of the
// X_GRAD stage, ONE exchange, and every rank merges world x blocks pairs in the same order -- the same value everywhere,
// without the 5-10 ms a host loop over 1e6 exponentials cost every run.
static int enqueue_logs0(bioen_hip_ctx* c, const Round& r) {
    launch_logw_logs0_part(c);
    const int rc = exchange(c, X_GRAD, 2 * (size_t)vec_grid(c));      // (per segment)
    if (rc) return rc;
    launch_logw_logs0_merge(c, r);
    return 0;
}

static int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch", __FILE__, __LINE__);
    return 0;
}

static void print_config(const bioen_lbfgs_config& p);
static void print_summary(const bioen_hip_ctx* c, const bioen_opt_result& r);

// ---------------------------------------------------------------------------------
// evaluation pipelines (all asynchronous on c->stream)
// ---------------------------------------------------------------------------------
static Round make_round(bioen_hip_ctx* c, const int* slots, int k, const double* stp, const double* theta) {
    Round r{};
    r.n = k;
    for (int a = 0; a < k; ++a) {
        ProblemSlot& sl = c->slot[slots[a]];
        r.x[a] = sl.x;
        r.xp[a] = sl.xp;
        r.g[a] = sl.g;
        r.gp[a] = sl.gp;
        r.d[a] = sl.d;
        r.w[a] = sl.w;
        r.a[a] = sl.a;
        r.scal[a] = sl.scal;
        r.part[a] = sl.part;
        r.stp[a] = stp ? stp[a] : 0.0;
        r.theta[a] = theta ? theta[a] : 0.0;
    }
    return r;
}

// log-weights: r.x must hold the points and P_MAX their block maxima (launch_trial does both).
// (the caller has produced this rank's block maxima in its X_MAX segment; they are not exchanged)
static int enqueue_logw_adjoint(bioen_hip_ctx* c, const Round& r);

static int enqueue_logw_eval(bioen_hip_ctx* c, const Round& r, bool with_grad) {
    int rc;
    c->last_width = r.n;
    c->last_pos = 0;
    c->last_centered = false;
    launch_logw_exp(c, r);                 // A1: e = exp(x - m_r) + prior partials (shift: this rank's own maximum)
    Vec8 w{};
    for (int a = 0; a < r.n; ++a) w.p[a] = r.w[a];
    int nblk = fwd_strip_blocks(c);
    if (nblk > 0) {
        // both strip copies before the first pass, so that one evaluation never mixes the two kernel families; a
        // copy that cannot be allocated switches the context to the streaming kernels for good (strips_unavailable)
        rc = ensure_strip_copy(c, 0);
        if (!rc && with_grad) rc = ensure_strip_copy_colsum(c);
        if (rc && !c->strips_unavailable) return rc;
        if (rc) nblk = 0;
    }
    if (nblk > 0) {
        // matrix pass 1 streams the strip-major copy (M > 1024: row panel by row panel) straight into the matrix cores
        // (kernels_strip.hip: k_strip_fwd) -- the same time for every batch width; the centre returns in
        // k_rows_combine, which leaves the RAW ybar in ybar_c for the centred adjoint below
        launch_fwd_strip(c, r.n, w, nblk);
        launch_fwd_rows_local(c, r.n, true, nblk, true);
        if ((rc = exchange(c, X_YBAR, (size_t)ybar_payload(c, r.n, true)))) return rc;
        launch_rows_combine(c, r, true, c->strip_center, true);
        return with_grad ? enqueue_logw_adjoint(c, r) : 0;
    }
    if ((rc = ensure_rowmajor(c))) return rc;
    launch_fwd_partial(c, r.n, w);         // A4: this rank's share of yTilde . e_a            [matrix pass 1]
    launch_fwd_rows_local(c, r.n, true);   //     + this rank's {sum e, sum e (x - G), m_r}
    if ((rc = exchange(c, X_YBAR, (size_t)ybar_payload(c, r.n, true)))) return rc;
    launch_rows_combine(c, r, true);       //     normalisation, ybar, r, chi^2, A5: f (identical on every rank)
    return with_grad ? enqueue_logw_adjoint(c, r) : 0;
}

// second half of an evaluation; needs w, scal[S_P] and the compact ybar_c / r_c of the SAME round
// still in place (nothing else evaluated in between)
static int enqueue_logw_adjoint(bioen_hip_ctx* c, const Round& r) {
    int rc;
    MVec8 out{};
    for (int a = 0; a < r.n; ++a) out.p[a] = r.a[a];
    const int nblk = fwd_strip_blocks(c);
    if (nblk > 0) {                              // matrix pass 2 on the column-sum strip copy (kernels_strip.hip)
        if ((rc = ensure_strip_copy_colsum(c))) return rc;
        MVec8 sc{};
        for (int a = 0; a < r.n; ++a) sc.p[a] = r.scal[a];
        launch_adj_strip(c, r.n, c->r_c, out, sc, nblk);
    } else {
        if ((rc = ensure_rowmajor(c))) return rc;
        launch_adj(c, r.n, c->r_c, out, true);   // A6: a_k = sum_i r_i (yTilde_ik - ybar_i)  [matrix pass 2]
    }
    launch_logw_grad(c, r);                //     gradient epilogue + g.d, g.g, x.x
    if ((rc = exchange(c, X_GRAD, 3 * r.n * (size_t)vec_grid(c)))) return rc;
    launch_finish_eval(c, r);
    return 0;
}

// forces: um holds the forces of the round's K problems, compact [row*K + a]
static ForcesRound make_forces_round(bioen_hip_ctx* c, const int* slots, int k, const double* theta) {
    ForcesRound r{};
    r.n = k;
    for (int a = 0; a < k; ++a) {
        ProblemSlot& sl = c->slot[slots[a]];
        r.a[a] = sl.a;
        r.w[a] = sl.w;
        r.t[a] = sl.d;
        r.scal[a] = sl.scal;
        r.part[a] = sl.part;
        r.theta[a] = theta ? theta[a] : 0.0;
    }
    return r;
}

static int enqueue_forces_weights(bioen_hip_ctx* c, const ForcesRound& fr) {     // streaming kernels (no row panels)
    MVec8 out{};
    for (int a = 0; a < fr.n; ++a) out.p[a] = fr.a[a];
    const int rc = ensure_rowmajor(c);        // the row-major matrix (back from the strip copy if it was freed)
    if (rc) return rc;
    launch_adj(c, fr.n, c->um, out, false);   // F1: x_j = sum_i f_i yTilde_ij     [matrix pass 1]
    launch_forces_max(c, fr);
    launch_forces_exp(c, fr);
    launch_forces_norm(c, fr);                // w ; KL partials
    return 0;
}

static int enqueue_forces_eval(bioen_hip_ctx* c, const ForcesRound& fr, const Round& r, bool with_grad) {
    int nblk = forces_fused_blocks(c);
    int rc;
    if (nblk > 0 && (rc = ensure_strip_copy(c, 1))) {
        if (!c->strips_unavailable) return rc;
        nblk = forces_fused_blocks(c);        // = 0 now: the kernels on the row-major matrix take over
    }
    c->last_width = r.n;
    c->last_pos = 0;
    c->last_centered = false;
    if (nblk > 0) {
        // M <= 1024: two passes over LDS-resident column strips instead of four streaming ones.
        // Measured at N = 1e6 x M = 512 (r01): pass 2 alone 0.85 ms at K = 1 against 1.22 ms for the
        // two passes it replaces (K = 4: 1.03 / 1.34, K = 8: 1.74 / 1.52).  Used for every K: the
        // paths add in different orders and a batched series must equal the single runs bit for bit.
        // Sharded contexts: every rank does this on its columns; the shares of ybar travel with the
        // rank's softmax totals in ONE all-gather (the layout of the log-weights rounds, finished by
        // the same k_rows_combine), the shares of the gradient in a second one.
        // Both passes read the strip-major copy centred on the targets (kernels_strip.hip); ybar_c then holds
        // ybar - center, the row offset of k_rows_combine puts the centre back for r, chi^2 and f.
        launch_forces_xy(c, fr, nblk);        // F1 + F2: x, online softmax, this rank's ybar   [matrix pass 1]
        if ((rc = exchange(c, X_YBAR, (size_t)ybar_payload(c, fr.n, true)))) return rc;
        launch_rows_combine(c, r, true, c->strip_center, false);   //     normalisation, ybar (centred), r, chi^2, KL, f
        c->last_centered = true;               // ybar_c = ybar - strip_center (bioen_hip_last_average adds it back)
        if (with_grad) {
            launch_forces_bt(c, fr, nblk);    // F3: b, t, product with t            [matrix pass 2]
            launch_fwd_rows_forces_grad_share(c, fr.n, nblk, &fr, true);   // every segment's share (one GPU: all eight)
            if ((rc = exchange(c, X_YBAR, (size_t)c->mp * fr.n))) return rc;
            launch_forces_grad_sum_ranks(c, fr.n);                         // ... added in segment order
        } else {
            launch_forces_w_from_x(c, fr);    // f-only evaluations hand out the weights
        }
        return 0;
    }
    // M > 1024: four passes.  r03: on the strip kernels over row panels of <= 1024 rows (kernels_strip.hip) -- the two
    // column-sum passes uncentred, the two row-sum passes centred on the targets as in the two-pass path, T = sum_j t_j
    // taken off in the gradient's reduction -- with the r01 streaming kernels as the fallback (BIOEN_HIP_PANELS=0, or
    // no memory for the panel copies; unsharded contexts only).
    int psets = strip_panels(c) ? fwd_strip_blocks(c) : 0;
    if (psets > 0) {
        rc = ensure_strip_copy(c);
        if (!rc) rc = ensure_strip_copy_colsum(c);
        if (rc && !c->strips_unavailable) return rc;
        if (rc) psets = 0;
    }
    if (psets > 0) {
        // r05: the panel path in canonical segments, on any number of ranks (two stage exchanges per evaluation, as in
        // the two-pass path): the softmax is merged segment by segment (kernels_forces.hip: k_forces_seg_exp), the row
        // sums of the two forward passes are the log-weights forward kernel's sets, shared out per segment
        MVec8 out{};
        Vec8 v{};
        Round rx = r;
        for (int a = 0; a < fr.n; ++a) {
            out.p[a] = fr.a[a];
            rx.x[a] = fr.a[a];
            v.p[a] = fr.w[a];
        }
        const StripSets ss = strip_sets(c);
        launch_adj_strip(c, fr.n, c->um, out, MVec8{}, psets, true);   // F1: x_j = sum_i f_i yTilde_ij   [matrix pass 1]
        launch_max(c, rx);                                             //     block maxima of x per segment
        launch_forces_seg_exp(c, fr);                                  //     e = w0 exp(x - m_v) -> w ; shares of sum e, sum e x
        launch_fwd_strip(c, fr.n, v, psets);                           // F2: (yTilde - centre) . e      [matrix pass 2]
        launch_fwd_rows_local(c, fr.n, true, psets, true);             //     the segments' shares + softmax totals
        if ((rc = exchange(c, X_YBAR, (size_t)ybar_payload(c, fr.n, true)))) return rc;
        launch_rows_combine(c, r, true, c->strip_center, false);       //     normalisation, ybar (centred), r, chi^2, KL, f
        c->last_centered = true;
        launch_scale_w(c, r);                                          //     w = e S_INV[v]: the weights
        if (with_grad) {
            launch_adj_strip(c, fr.n, c->r_c, out, MVec8{}, psets, true);   // F3: b = yTilde^T r  [matrix pass 3]
            launch_forces_seg_t(c, fr, ss.gs * (ss.fold ? 1 : ss.nch));     //     t_j ; T_v = sum over segment v
            for (int a = 0; a < fr.n; ++a) v.p[a] = fr.t[a];
            launch_fwd_strip(c, fr.n, v, psets);                            //     sum_j (yTilde_ij - c_i) t_j   [matrix pass 4]
            launch_fwd_rows_forces_grad_share(c, fr.n, 0, &fr, false);      //     ... - (ybar_i - c_i) T_v, per segment
            if ((rc = exchange(c, X_YBAR, (size_t)c->mp * fr.n))) return rc;
            launch_forces_grad_sum_ranks(c, fr.n);                          //     added in segment order
        }
        return 0;
    }
    if (c->world != 1)      // (no strip copies: switched off by the environment, or no memory for them)
        return fail(BIOEN_HIP_ESTATE, "the forces method on a sharded context needs the strip copies of the matrix (M > 1024: the row panels)");
    if ((rc = enqueue_forces_weights(c, fr))) return rc;
    Vec8 v{};
    for (int a = 0; a < fr.n; ++a) v.p[a] = fr.w[a];
    launch_fwd_partial(c, fr.n, v);           // F2: ybar                         [matrix pass 2]
    launch_fwd_rows_local(c, fr.n, false);
    launch_rows_combine(c, r, false);
    launch_forces_scalars(c, fr);             //     f = theta KL + 0.5 chi^2
    if (with_grad) {
        MVec8 out{};
        for (int a = 0; a < fr.n; ++a) out.p[a] = fr.a[a];
        launch_adj(c, fr.n, c->r_c, out, false);   // F3: b = yTilde^T r          [matrix pass 3]
        launch_forces_t(c, fr);               //     t_j
        for (int a = 0; a < fr.n; ++a) v.p[a] = fr.t[a];
        launch_fwd_partial(c, fr.n, v, true); //     gm_i = sum_j (yTilde_ij - ybar_i) t_j  [matrix pass 4]
        launch_fwd_rows_forces_grad(c, fr.n, c->fwd_ctiles);
    }
    return 0;
}

#include "engine_logw.inl"
#include "engine_devls.inl"
#include "engine_forces.inl"
#include "selftest_lbfgs.inl"

static void print_config(const bioen_lbfgs_config& p) {
    // same table the reference prints when verbose (c_bioen_kernels_logw.c:620-634)
    std::printf("\t=========================\n");
    std::printf("\tdevice-resident L-BFGS   : gfx950 / HIP\n");
    std::printf("\tlinesearch               : %d\n", p.linesearch);
    std::printf("\tmax_iterations           : %d\n", p.max_iterations);
    std::printf("\tdelta                    : %lf\n", p.delta);
    std::printf("\tepsilon                  : %lf\n", p.epsilon);
    std::printf("\tftol                     : %lf\n", p.ftol);
    std::printf("\tgtol                     : %lf\n", p.gtol);
    std::printf("\twolfe                    : %lf\n", p.wolfe);
    std::printf("\tpast                     : %d\n", p.past);
    std::printf("\tmax_linesearch           : %d\n", p.max_linesearch);
    std::printf("\t=========================\n");
}

static void print_summary(const bioen_hip_ctx* c, const bioen_opt_result& r) {
    std::printf("\t%s\n", lbfgs_code_string(r.lbfgs_code));
    std::printf("\tConfig: m=%d and n=%d\n", c->m, c->n);
    std::printf("\tCurrent function value  = %.6lf\n", r.fmin);
    std::printf("\tIterations              : %d\n", r.iterations);
    std::printf("\tEvaluations             : %d\n", r.evaluations);
    std::printf("\tTime(s) of L-BFGS       : %.12lf\n", r.seconds);
    std::printf("\tTime(s) per iter        : %.12lf\n", r.iterations ? r.seconds / r.iterations : 0.0);
    std::fflush(stdout);
}

static void resolve_timers(bioen_hip_ctx* c) {
    KernelTimer& t = c->timer;
    for (auto& p : t.pending) {
        hipEventSynchronize(p.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            t.total_ms[p.which] += ms;
            t.launches[p.which] += 1;
            t.problem_passes[p.which] += p.k;
        }
        t.pool.push_back(p);
    }
    t.pending.clear();
}

}  // namespace bioen

using namespace bioen;

// =====================================================================================
// C ABI
// =====================================================================================
extern "C" {

const char* bioen_hip_version(void) { return "bioen_hip 0.1 (gfx950)"; }

int bioen_hip_device_count(int* count) {
    if (!count) return fail(BIOEN_HIP_EINVAL, "count is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return 0;
}

const char* bioen_hip_strerror(int code) {
    switch (code) {
        case BIOEN_HIP_OK: return "success";
        case BIOEN_HIP_EINVAL: return "invalid argument";
        case BIOEN_HIP_ENODEV: return "no HIP device available";
        case BIOEN_HIP_EHIP: return "HIP runtime error";
        case BIOEN_HIP_ENOMEM: return "out of memory";
        case BIOEN_HIP_ERCCL: return "RCCL error";
        case BIOEN_HIP_ESTATE: return "invalid state";
        default: return "unknown bioen_hip error";
    }
}

const char* bioen_hip_last_error(void) { return g_last_error.c_str(); }
const char* bioen_hip_lbfgs_strerror(int code) { return lbfgs_code_string(code); }
void bioen_hip_set_fast_openmp_flag(int flag) { g_fast_openmp_flag = flag; }
int bioen_hip_get_fast_openmp_flag(void) { return g_fast_openmp_flag; }

int bioen_hip_ctx_create_sharded(int m, long long n, const double* yTilde, const double* YTilde, int device,
                                 int rank, int world, bioen_hip_ctx** ctx) {
    if (!yTilde || !YTilde) return fail(BIOEN_HIP_EINVAL, "yTilde / YTilde is NULL");
    bioen_hip_ctx* c = nullptr;
    int rc = ctx_alloc(m, n, device, rank, world, &c);
    if (rc) return rc;
    // this rank's column block [col0, col0 + n_local) of the host matrix
    hipError_t e = staged_upload_forced() ? hipErrorInvalidValue
                                          : hipMemcpy2DAsync(c->Y, c->ld * sizeof(double), yTilde + c->col0, (size_t)n * sizeof(double),
                                                             (size_t)c->n * sizeof(double), (size_t)m, hipMemcpyHostToDevice, c->stream);
    if (e == hipErrorInvalidValue) {            // (h2d_staged: a caller's buffer the runtime will not pin)
        (void)hipGetLastError();
        if (h2d_staged(c, reinterpret_cast<char*>(c->Y), c->ld * sizeof(double), reinterpret_cast<const char*>(yTilde + c->col0),
                       (size_t)n * sizeof(double), (size_t)c->n * sizeof(double), (size_t)m)) {
            bioen_hip_ctx_destroy(c);
            return BIOEN_HIP_EHIP;
        }
        e = hipSuccess;
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(c->YT, YTilde, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        bioen_hip_ctx_destroy(c);
        return hip_fail(e, "upload of yTilde", __FILE__, __LINE__);
    }
    *ctx = c;
    return 0;
}

int bioen_hip_ctx_create(int m, int n, const double* yTilde, const double* YTilde, int device,
                         bioen_hip_ctx** ctx) {
    return bioen_hip_ctx_create_sharded(m, n, yTilde, YTilde, device, 0, 1, ctx);
}

int bioen_hip_ctx_create_raw(int m, long long n, int structure_major, const double* sim, const double* exp_values,
                             const double* exp_err, int device, bioen_hip_ctx** ctx) {
    if (!sim || !exp_values || !exp_err) return fail(BIOEN_HIP_EINVAL, "sim / exp / exp_err is NULL");
    for (int i = 0; i < m; ++i)
        if (!(exp_err[i] > 0.0)) return fail(BIOEN_HIP_EINVAL, "experimental errors must be positive");
    bioen_hip_ctx* c = nullptr;
    int rc = ctx_alloc(m, n, device, 0, 1, &c);
    if (rc) return rc;
    std::vector<double> yt(m);
    for (int i = 0; i < m; ++i) yt[i] = exp_values[i] / exp_err[i];
    double* sigma = nullptr;     // device copy of the errors
    double* stage = nullptr;     // structure-major chunks on their way through the transposer
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&sigma), (size_t)m * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(sigma, exp_err, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->YT, yt.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && !structure_major) {
        e = staged_upload_forced() ? hipErrorInvalidValue
                                   : hipMemcpy2DAsync(c->Y, c->ld * sizeof(double), sim, (size_t)n * sizeof(double), (size_t)n * sizeof(double),
                                                      (size_t)m, hipMemcpyHostToDevice, c->stream);
        if (e == hipErrorInvalidValue) {        // (h2d_staged: a caller's buffer the runtime will not pin)
            (void)hipGetLastError();
            e = h2d_staged(c, reinterpret_cast<char*>(c->Y), c->ld * sizeof(double), reinterpret_cast<const char*>(sim),
                           (size_t)n * sizeof(double), (size_t)n * sizeof(double), (size_t)m) ? hipErrorUnknown : hipSuccess;
        }
        if (e == hipSuccess) launch_rows_div(c, sigma);
    } else if (e == hipSuccess) {
        const long long chunk = std::max<long long>(1, std::min<long long>(n, (256ll << 20) / ((long long)m * 8)));
        e = hipMalloc(reinterpret_cast<void**>(&stage), (size_t)chunk * m * sizeof(double));
        for (long long j0 = 0; e == hipSuccess && j0 < n; j0 += chunk) {
            const long long nc = std::min(chunk, n - j0);
            e = h2d_user(c, stage, sim + (size_t)j0 * m, (size_t)nc * m * sizeof(double)) ? hipErrorUnknown : hipSuccess;
            if (e == hipSuccess) launch_transpose_div(c, stage, (int)nc, (size_t)j0, sigma);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (sigma) hipFree(sigma);
    if (stage) hipFree(stage);
    if (e != hipSuccess) {
        bioen_hip_ctx_destroy(c);
        return hip_fail(e, "assembly of yTilde", __FILE__, __LINE__);
    }
    *ctx = c;
    return 0;
}

int bioen_hip_ctx_create_synthetic_sharded(int m, long long n, const double* YTrue, const double* sig_sim,
                                           const double* sig_exp, const double* YTilde, unsigned long long seed,
                                           int device, int rank, int world, bioen_hip_ctx** ctx) {
    if (!YTrue || !sig_sim || !sig_exp || !YTilde) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    bioen_hip_ctx* c = nullptr;
    int rc = ctx_alloc(m, n, device, rank, world, &c);
    if (rc) return rc;
    // stage the three M-vectors in ybar_c / r_c / um (all >= mp long), then generate in place
    hipError_t e = hipMemcpyAsync(c->ybar_c, YTrue, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(c->r_c, sig_sim, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(c->um, sig_exp, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(c->YT, YTilde, (size_t)m * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_generate(c, c->ybar_c, c->r_c, c->um, seed);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemsetAsync(c->ybar_c, 0, c->mp * sizeof(double), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->r_c, 0, c->mp * sizeof(double), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->um, 0, c->mp * sizeof(double), c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        bioen_hip_ctx_destroy(c);
        return hip_fail(e, "synthetic generation", __FILE__, __LINE__);
    }
    *ctx = c;
    return 0;
}

int bioen_hip_ctx_create_synthetic(int m, int n, const double* YTrue, const double* sig_sim,
                                   const double* sig_exp, const double* YTilde, unsigned long long seed,
                                   int device, bioen_hip_ctx** ctx) {
    return bioen_hip_ctx_create_synthetic_sharded(m, n, YTrue, sig_sim, sig_exp, YTilde, seed, device, 0, 1, ctx);
}

int bioen_hip_ctx_set_exchange_callback(bioen_hip_ctx* c, bioen_hip_exchange_fn fn, void* user) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    c->exchange_cb = fn;
    c->exchange_user = user;
    return 0;
}

int bioen_hip_ctx_set_force_exchange(bioen_hip_ctx* c, int on) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    c->force_exchange = on ? 1 : 0;
    return 0;
}

int bioen_hip_ctx_set_mirror_exchange(bioen_hip_ctx* c, int on) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    c->mirror_exchange = on ? 1 : 0;
    return 0;
}

int bioen_hip_exchange_counts(const bioen_hip_ctx* c, long long* rccl, long long* host_staged) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    if (rccl) *rccl = c->n_rccl_exchanges;
    if (host_staged) *host_staged = c->n_host_exchanges;
    return 0;
}

int bioen_hip_ctx_shard(const bioen_hip_ctx* c, int* rank, int* world, long long* n_global, long long* col0,
                        int* n_local) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (n_global) *n_global = c->n_global;
    if (col0) *col0 = c->col0;
    if (n_local) *n_local = c->n;
    return 0;
}

int bioen_hip_ctx_destroy(bioen_hip_ctx* c) {
    if (!c) return 0;
    hipSetDevice(c->device);
    if (c->stream && c->comm) {      // a collective a dead peer left hanging must not hang the destructor: bounded, then abort
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(c->failed ? 0.0 : c->wait_timeout_s);
        while (hipStreamQuery(c->stream) == hipErrorNotReady) {
            if (std::chrono::steady_clock::now() > deadline) {
                c->failed = 1;
                rccl_abort(c);
                break;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    if (c->stream) hipStreamSynchronize(c->stream);
    bioen_hip_p2p_detach(c);
    bioen_hip_comm_destroy(c);
    resolve_timers(c);
    for (auto& p : c->timer.pool) {
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    double* bufs[] = {c->Y, c->Ys, c->Ys1, c->strip_center, c->zero_center, c->strip_stamps, c->YT, c->row_offset, c->row_scale, c->gram, c->ybar_c, c->r_c, c->um, c->gm, c->fixed,
                      c->t, c->g0, c->fwd_partial, c->part, c->scal};
    for (double* p : bufs)
        if (p) hipFree(p);
    for (int s = 0; s < kMaxBatch; ++s) {
        ProblemSlot& sl = c->slot[s];
        double* v[] = {sl.xa, sl.xb, sl.ga, sl.gb, sl.d, sl.w, sl.a, sl.Ssp, sl.Ysp};
        for (double* p : v)
            if (p) hipFree(p);
        for (int i = 0; i < kHistory; ++i) {
            if (sl.S[i]) hipFree(sl.S[i]);
            if (sl.Yh[i]) hipFree(sl.Yh[i]);
        }
    }
    for (int st = 0; st < X_COUNT; ++st)
        if (c->xbuf[st]) hipFree(c->xbuf[st]);
    if (c->exchange_host) hipHostFree(c->exchange_host);
    if (c->stage_host) hipHostFree(c->stage_host);
    if (c->host_scal) hipHostFree(c->host_scal);
    if (c->live) hipHostFree(c->live);
    if (c->live2) hipHostFree(c->live2);
    for (int p = 0; p < bioen_hip_ctx::kMaxPanels; ++p) {
        if (c->Yp[p]) hipFree(c->Yp[p]);
        if (c->Y1p[p]) hipFree(c->Y1p[p]);
    }
    if (c->Yr) hipFree(c->Yr);
    if (c->Yr1) hipFree(c->Yr1);
    if (c->live_f) hipHostFree(c->live_f);
    if (c->dev_tab) hipFree(c->dev_tab);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->host_m) hipHostFree(c->host_m);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int bioen_hip_ctx_shape(const bioen_hip_ctx* c, int* m, int* n) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    if (m) *m = c->m;
    if (n) *n = c->n;
    return 0;
}

int bioen_hip_ctx_read_ytilde(bioen_hip_ctx* c, int row0, int rows, int col0, int cols, double* out) {
    if (!c || !out) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    // (col0 is relative to this context's own column block when the matrix is sharded)
    if (row0 < 0 || col0 < 0 || rows <= 0 || cols <= 0 || row0 + rows > c->m || col0 + cols > c->n)
        return fail(BIOEN_HIP_EINVAL, "block out of range");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    if (c->Y) {
        BIOEN_HIP_CHECK(d2h_user_2d(c->stream, reinterpret_cast<char*>(out), (size_t)cols * sizeof(double),
                                    reinterpret_cast<const char*>(c->Y + (size_t)row0 * c->ld + col0), c->ld * sizeof(double),
                                    (size_t)cols * sizeof(double), (size_t)rows));
        BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
        return 0;
    }
    // the row-major matrix has made way for the strip copy, which holds the same numbers: gathered back in column
    // chunks through a bounded staging buffer
    const int chunk = std::max(1, std::min(cols, (int)((64ll << 20) / ((long long)rows * 8))));
    double* stage = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&stage), (size_t)rows * chunk * sizeof(double));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc (read-back staging)", __FILE__, __LINE__);
    int rc = 0;
    for (int c0 = 0; c0 < cols && !rc; c0 += chunk) {
        const int nc = std::min(chunk, cols - c0);
        rc = gather_block(c, row0, rows, (size_t)col0 + c0, nc, stage);
        if (!rc) {
            e = d2h_user_2d(c->stream, reinterpret_cast<char*>(out + c0), (size_t)cols * sizeof(double),
                            reinterpret_cast<const char*>(stage), (size_t)nc * sizeof(double), (size_t)nc * sizeof(double), (size_t)rows);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) rc = hip_fail(e, "read-back", __FILE__, __LINE__);
        }
    }
    (void)hipFree(stage);
    return rc;
}

int bioen_hip_ctx_layout(const bioen_hip_ctx* c, int* one_copy, int* interleave, int* relayouts) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    if (one_copy) *one_copy = c->one_copy;
    if (interleave) *interleave = std::max(1, c->strip_ilv);
    if (relayouts) *relayouts = c->strip_relayouts;
    return 0;
}

int bioen_hip_ctx_footprint(const bioen_hip_ctx* c, int* forms, long long* bytes) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    const long long rowmajor = (long long)c->mp * (long long)c->ld * 8;
    const long long strips = (long long)(c->ld / 16) * ((c->m + 15) / 16 * 16) * 16 * 8;
    int f = 0;
    long long b = 0;
    if (c->Y) { f |= 1; b += rowmajor; }
    if (c->Ys) { f |= 2; b += strips; }
    if (c->Ys1) { f |= 4; b += strips; }
    for (int p = 0; p < bioen_hip_ctx::kMaxPanels; ++p) {        // M > 1024: row panels of <= 1024 rows, both orders
        const long long rows = std::min(1024, c->m - p * 1024);
        if (rows <= 0) break;
        const long long panel = (long long)(c->ld / 16) * ((rows + 15) / 16 * 16) * 16 * 8;
        if (c->Yp[p]) { f |= 2; b += panel; }
        if (c->Y1p[p]) { f |= 4; b += panel; }
    }
    if (c->storage) {            // reduced-storage experiment: centred copies of 6 or 4 bytes per element, rows padded to 64 (128)
        const long long rows = (long long)round_up((size_t)c->m, c->mp > 512 ? 128 : 64);
        const long long each = (long long)(c->ld / 16) * (rows / 64) * (c->storage == 1 ? 6144 : 4096);
        if (c->Yr) { f |= 8; b += each; }
        if (c->Yr1) { f |= 8; b += each; }
    }
    if (forms) *forms = f;
    if (bytes) *bytes = b;
    return 0;
}

int bioen_hip_ctx_set_ytilde_target(bioen_hip_ctx* c, const double* YTilde) {
    if (!c || !YTilde) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    BIOEN_HIP_CHECK(hipMemcpyAsync(c->YT, YTilde, (size_t)c->m * sizeof(double), hipMemcpyHostToDevice, c->stream));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int bioen_hip_ctx_set_affine(bioen_hip_ctx* c, const double* row_offset, const double* row_scale) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    std::vector<double> off(c->mp, 0.0), sc(c->mp, 1.0);
    bool affine = false;
    for (int i = 0; i < c->m; ++i) {
        if (row_offset) off[i] = row_offset[i];
        if (row_scale) sc[i] = row_scale[i];
        affine = affine || off[i] != 0.0 || sc[i] != 1.0;
    }
    BIOEN_HIP_CHECK(hipMemcpyAsync(c->row_offset, off.data(), (size_t)c->mp * sizeof(double), hipMemcpyHostToDevice,
                                   c->stream));
    BIOEN_HIP_CHECK(hipMemcpyAsync(c->row_scale, sc.data(), (size_t)c->mp * sizeof(double), hipMemcpyHostToDevice,
                                   c->stream));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->affine = affine;
    return 0;
}

int bioen_hip_ctx_set_storage(bioen_hip_ctx* c, int format) {
    if (!c || format < 0 || format > 2) return fail(BIOEN_HIP_EINVAL, "format must be 0 (FP64), 1 (fp32 + bf16 split) or 2 (fp32)");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    if (format != 0 && c->mp > 1024) return fail(BIOEN_HIP_ESTATE, "the reduced-storage experiment serves M <= 1024");
    const int rc = set_storage_format(c, format);
    if (rc == BIOEN_HIP_ESTATE) return fail(rc, "no form of the matrix to build the copies from");
    return rc;
}

int bioen_hip_ctx_set_one_copy(bioen_hip_ctx* c, int on) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    if (on && (c->Ys1 || c->Y1p[0])) return fail(BIOEN_HIP_ESTATE, "the column-sum order copy exists already: ask before the first gradient evaluation");
    if (!on && c->one_copy) {        // back to two copies: the second one is built at the next gradient evaluation
        c->one_copy = 0;
    }
    c->one_copy_wanted = on ? 1 : 0;
    return 0;
}

int bioen_hip_ctx_set_direction_mode(bioen_hip_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 2) return fail(BIOEN_HIP_EINVAL, "mode must be 0 (auto), 1 (two-loop) or 2 (Gram form)");
    c->direction_mode = mode;
    return 0;
}

int bioen_hip_synchronize(bioen_hip_ctx* c) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

// ---- log-weights ----------------------------------------------------------------------
int bioen_hip_logw_weights(bioen_hip_ctx* c, const double* g, double* w, double* log_s) {
    if (!c || !g) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    ProblemSlot& s0 = c->slot[0];
    int rc = upload_n(c, s0.x, g);
    if (rc) return rc;
    BIOEN_HIP_CHECK(hipMemsetAsync(c->fixed, 0, c->ld * sizeof(double), c->stream));
    const int one[1] = {0};
    const Round r = make_round(c, one, 1, nullptr, nullptr);
    const size_t gsz = (size_t)vec_grid(c);
    launch_max(c, r);
    launch_logw_exp(c, r);
    if ((rc = exchange(c, X_EXP, 3 * gsz))) return rc;      // (per segment)
    launch_logw_norm(c, r);
    if ((rc = check_launch())) return rc;
    if (w && (rc = download_n(c, w, s0.w))) return rc;
    if ((rc = read_scalars(c))) return rc;
    if (log_s) *log_s = c->host_scal[S_LOGS];
    return 0;
}

int bioen_hip_logw_fdf(bioen_hip_ctx* c, const double* g, const double* G, double theta, double* f,
                       double* grad) {
    if (!c || !g || !G) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    ProblemSlot& s0 = c->slot[0];
    int rc;
    if ((rc = upload_n(c, s0.x, g))) return rc;
    if ((rc = upload_n(c, c->fixed, G))) return rc;
    if (grad) BIOEN_HIP_CHECK(hipMemsetAsync(s0.d, 0, c->ld * sizeof(double), c->stream));
    const int one[1] = {0};
    const Round r = make_round(c, one, 1, nullptr, &theta);
    if ((rc = enqueue_logs0(c, r))) return rc;
    launch_max(c, r);
    if ((rc = enqueue_logw_eval(c, r, grad != nullptr))) return rc;
    if ((rc = check_launch())) return rc;
    if (grad && (rc = download_n(c, grad, s0.g))) return rc;
    if ((rc = read_scalars(c))) return rc;
    if (f) *f = c->host_scal[S_F];
    return 0;
}

namespace {
// The caller's result arrays (numpy.empty: fresh, pageable) are where every finished problem's optimum and weights go --
// 16 MB per theta at the headline.  A device-to-host copy into PAGEABLE memory is staged by the runtime piece by piece and
// competes with the rounds still running (measured at the headline: the slowest theta itself takes 1.094 s with pageable
// destinations, 1.068 s with pinned ones), and the first write to every page is a page fault.  So a helper thread, while
// the GPU works, (1) populates the pages (MADV_POPULATE_WRITE: page tables made writable, contents untouched -- safe
// beside deliveries that already write there) and (2) registers the ranges with the runtime (hipHostRegister) for the
// duration of the call: deliveries issued after that are plain DMA transfers.  Unregistered before the call returns; a
// range the caller has registered itself, a locked-memory limit, or a kernel without the advice simply leave that step
// out.  BIOEN_HIP_PIN_RESULTS=0 switches the registration off (A/B).  Whole sweep at the headline, same box: 1.100-1.110 s
// -> 1.072-1.077 s.
struct ResultPinner {
    std::thread th;
    void* reg[2] = {nullptr, nullptr};
    ResultPinner(int device, void* a, size_t na, void* b, size_t nb) {
        if (na + nb < ((size_t)8 << 20)) return;
        const char* e = std::getenv("BIOEN_HIP_PIN_RESULTS");
        const bool pin = !(e && e[0] == '0');
        void** regp = reg;
        try {
            th = std::thread([device, a, na, b, nb, pin, regp]() {
                const size_t page = (size_t)sysconf(_SC_PAGESIZE);
                void* ptr[2] = {a, b};
                const size_t len[2] = {na, nb};
                if (pin && hipSetDevice(device) != hipSuccess) (void)hipGetLastError();
                for (int i = 0; i < 2; ++i) {
                    if (!ptr[i] || !len[i]) continue;
                    const size_t lo = (size_t)ptr[i] / page * page, hi = ((size_t)ptr[i] + len[i] + page - 1) / page * page;
#ifdef MADV_POPULATE_WRITE
                    bool populated = true;
                    for (size_t at = lo; at < hi && populated; at += (size_t)16 << 20)   // in pieces: the first problems to finish come first
                        populated = madvise(reinterpret_cast<void*>(at), std::min(hi - at, (size_t)16 << 20), MADV_POPULATE_WRITE) == 0;
#endif
                    if (pin) {
                        if (hipHostRegister(reinterpret_cast<void*>(lo), hi - lo, hipHostRegisterDefault) == hipSuccess)
                            regp[i] = reinterpret_cast<void*>(lo);
                        else
                            (void)hipGetLastError();
                    }
                }
            });
        } catch (...) {
        }
    }
    ~ResultPinner() {
        if (th.joinable()) th.join();
        for (void* r : reg)
            if (r && hipHostUnregister(r) != hipSuccess) (void)hipGetLastError();
    }
};
}  // namespace

int bioen_hip_opt_lbfgs_logw_batch(bioen_hip_ctx* c, int ntheta, const double* thetas, const double* g0,
                                   size_t g0_stride, const double* G, const bioen_lbfgs_config* config,
                                   const bioen_visual_params* visual, int max_batch, double* results,
                                   double* w_opt, bioen_opt_result* infos) {
    if (!c || !thetas || !g0 || !G || !config || !results || !infos || ntheta <= 0)
        return fail(BIOEN_HIP_EINVAL, "NULL argument or ntheta <= 0");
    if (g0_stride != 0 && g0_stride < (size_t)c->n_global)
        return fail(BIOEN_HIP_EINVAL, "g0_stride must be 0 or >= n");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    const bool verbose = visual && visual->verbose;
    if (verbose) {
        std::printf("L-BFGS minimizer (%d theta value%s, up to %d per matrix pass)\n", ntheta, ntheta > 1 ? "s" : "",
                    std::max(1, std::min(max_batch, (int)kMaxBatch)));
        print_config(*config);
    }
    ResultPinner pin(c->device, results, (size_t)ntheta * c->n_global * sizeof(double), w_opt,
                     w_opt ? (size_t)ntheta * c->n_global * sizeof(double) : 0);
    LogwBatchEngine eng(c, *config, verbose);
    const int rc = eng.run(ntheta, thetas, g0, g0_stride, G, max_batch, results, w_opt, infos);
    if (ntheta > 1) c->last_width = 0;       // bioen_hip_last_average: single-problem calls only
    return rc;
}

int bioen_hip_opt_lbfgs_logw(bioen_hip_ctx* c, const double* g0, const double* G, double theta,
                             const bioen_lbfgs_config* config, const bioen_visual_params* visual,
                             double* result, double* w_opt, bioen_opt_result* info) {
    if (!info) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    return bioen_hip_opt_lbfgs_logw_batch(c, 1, &theta, g0, 0, G, config, visual, 1, result, w_opt, info);
}

// ---- forces ---------------------------------------------------------------------------
static bool is_affine(const bioen_hip_ctx* c) { return c->affine; }

// the forces evaluation in canonical segments: the two-pass strip kernels (M <= 1024) or the row panels (M > 1024)
static bool forces_canonical(const bioen_hip_ctx* c) {
    return forces_fused_blocks(c) > 0 || (strip_panels(c) && fwd_strip_blocks(c) > 0);
}

static int forces_guard(const bioen_hip_ctx* c, bool strip_path_ok = true) {
    // sharded contexts run the forces method in canonical segments only (not on the r01 streaming kernels)
    if (c->world != 1 && !(strip_path_ok && forces_canonical(c)))
        return fail(BIOEN_HIP_ESTATE, "not available on this structure-sharded context");
    if (is_affine(c)) return fail(BIOEN_HIP_ESTATE, "the affine observable model is implemented for the log-weights method");
    if (c->storage && !(strip_path_ok && forces_fused_blocks(c) > 0))
        return fail(BIOEN_HIP_ESTATE, "the reduced-storage experiment serves the strip passes (M <= 1024) only");
    return 0;
}

int bioen_hip_forces_weights(bioen_hip_ctx* c, const double* forces, const double* w0, double* w) {
    if (!c || !forces || !w0 || !w) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    int rc;
    if (forces_canonical(c) && !c->storage) {
        // r05: the first half of an evaluation IS _get_weights_from_forces -- x = yTilde^T f, the softmax over all
        // structures merged segment by segment -- so sharded contexts serve the call too (one stage exchange), and every rank
        // count returns the single-GPU bits
        if ((rc = forces_guard(c))) return rc;
        BIOEN_HIP_CHECK(hipSetDevice(c->device));
        if ((rc = upload_n(c, c->fixed, w0))) return rc;
        bioen_lbfgs_config dummy{};
        ForcesBatchEngine eng(c, dummy, false);
        const int one[1] = {0};
        const double* pt[1] = {forces};
        const double theta0 = 0.0;
        eng.evaluate(one, 1, pt, &theta0, false);          // an f-only evaluation hands out the weights (slot 0)
        if (eng.rc) return eng.rc;
        if ((rc = download_n(c, w, c->slot[0].w))) return rc;
        BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
        return transport_error(c);
    }
    rc = forces_guard(c, false);              // streaming kernels (no strip copies): unsharded contexts only
    if (rc) return rc;
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    if ((rc = upload_n(c, c->fixed, w0))) return rc;
    BIOEN_HIP_CHECK(hipMemcpyAsync(c->um, forces, (size_t)c->m * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const int one[1] = {0};
    if ((rc = enqueue_forces_weights(c, make_forces_round(c, one, 1, nullptr)))) return rc;
    if ((rc = check_launch())) return rc;
    BIOEN_HIP_CHECK(d2h_user(c->stream, w, c->slot[0].w, (size_t)c->n * sizeof(double)));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int bioen_hip_forces_fdf(bioen_hip_ctx* c, const double* forces, const double* w0, double theta, double* f,
                         double* grad) {
    if (!c || !forces || !w0) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    int rc = forces_guard(c);
    if (rc) return rc;
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    if ((rc = upload_n(c, c->fixed, w0))) return rc;
    bioen_lbfgs_config dummy{};
    ForcesBatchEngine eng(c, dummy, false);
    const int one[1] = {0};
    const double* pt[1] = {forces};
    eng.evaluate(one, 1, pt, &theta, grad != nullptr);
    if (eng.rc) return eng.rc;
    if (grad) std::memcpy(grad, eng.gm_h, (size_t)c->m * sizeof(double));
    if (f) *f = c->host_scal[S_F];
    return 0;
}

int bioen_hip_speculation_stats(bioen_hip_ctx* c, long long* issued, long long* adopted) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    if (issued) *issued = c->spec_launched;
    if (adopted) *adopted = c->spec_used;
    return 0;
}

// diagnostic builds (-DSTRIP_DIAG=4): phase-cycle sums of the last forces strip launch, [blocks][16][8]
int bioen_hip_debug_strip_stamps(bioen_hip_ctx* c, int enable, long long* out, int nblocks) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    const size_t cnt = (size_t)kPartStride * 16 * 8;
    if (enable && !c->strip_stamps) {
        int rc = dalloc_zero(&c->strip_stamps, cnt, c->stream);
        if (rc) return rc;
    }
    if (out && c->strip_stamps) {
        BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
        BIOEN_HIP_CHECK(hipMemcpy(out, c->strip_stamps, (size_t)nblocks * 16 * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    }
    return 0;
}

// Measurement aid (tools/pass_probe.py): the two log-weights matrix passes alone, `reps` launches each at batch width k on
// whatever the slots' vectors hold, timed with events on the context's stream.  No result is produced or changed.
int bioen_hip_debug_pass_probe(bioen_hip_ctx* c, int k, int reps, double* fwd_ms, double* adj_ms) {
    if (!c || !fwd_ms || !adj_ms || reps <= 0) return fail(BIOEN_HIP_EINVAL, "bad argument");
    if (k < 1 || k > kMaxBatch) return fail(BIOEN_HIP_EINVAL, "k must be in [1, 8]");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    int rc;
    for (int s = 0; s < k; ++s)
        if ((rc = alloc_slot(c, s, false))) return rc;
    const int nblk = fwd_strip_blocks(c);
    if (nblk <= 0) return fail(BIOEN_HIP_ESTATE, "the strip passes do not serve this context");
    if ((rc = ensure_strip_copy(c, 0)) || (rc = ensure_strip_copy_colsum(c))) return rc;
    Vec8 w{};
    MVec8 out{}, sc{};
    for (int a = 0; a < k; ++a) {
        w.p[a] = c->slot[a].w;
        out.p[a] = c->slot[a].a;
        sc.p[a] = c->slot[a].scal;
    }
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipEventCreate(&ev[i]);
    float f_ms = 0.f, a_ms = 0.f;
    if (e == hipSuccess) {
        for (int i = 0; i < 2; ++i) {
            launch_fwd_strip(c, k, w, nblk);
            launch_adj_strip(c, k, c->r_c, out, sc, nblk);
        }
        e = hipEventRecord(ev[0], c->stream);
        for (int i = 0; i < reps; ++i) launch_fwd_strip(c, k, w, nblk);
        if (e == hipSuccess) e = hipEventRecord(ev[1], c->stream);
        for (int i = 0; i < reps; ++i) launch_adj_strip(c, k, c->r_c, out, sc, nblk);
        if (e == hipSuccess) e = hipEventRecord(ev[2], c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(ev[2]);
        if (e == hipSuccess) e = hipEventElapsedTime(&f_ms, ev[0], ev[1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&a_ms, ev[1], ev[2]);
        if (e == hipSuccess) e = hipGetLastError();
    }
    for (int i = 0; i < 3; ++i)
        if (ev[i]) (void)hipEventDestroy(ev[i]);
    if (e != hipSuccess) return hip_fail(e, "pass probe", __FILE__, __LINE__);
    *fwd_ms = f_ms / reps;
    *adj_ms = a_ms / reps;
    return 0;
}

int bioen_hip_forces_fdf_batch(bioen_hip_ctx* c, int k, const double* forces, const double* w0, const double* thetas,
                               double* f, double* grad) {
    if (!c || !forces || !w0 || !thetas) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    if (k < 1 || k > kMaxBatch) return fail(BIOEN_HIP_EINVAL, "k must be in [1, 8]");
    int rc = forces_guard(c);
    if (rc) return rc;
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    for (int s = 0; s < k; ++s)
        if ((rc = alloc_slot(c, s, false))) return rc;
    if ((rc = upload_n(c, c->fixed, w0))) return rc;
    bioen_lbfgs_config dummy{};
    ForcesBatchEngine eng(c, dummy, false);
    int slots[kMaxBatch];
    const double* pt[kMaxBatch];
    for (int a = 0; a < k; ++a) {
        slots[a] = a;
        pt[a] = forces + (size_t)a * c->m;
    }
    eng.evaluate(slots, k, pt, thetas, grad != nullptr);
    if (eng.rc) return eng.rc;
    for (int a = 0; a < k; ++a) {
        if (grad)
            for (int i = 0; i < c->m; ++i) grad[(size_t)a * c->m + i] = eng.gm_h[(size_t)i * k + a];
        if (f) f[a] = c->host_scal[(size_t)a * kScalStride + S_F];
    }
    if (k > 1) c->last_width = 0;
    return 0;
}

int bioen_hip_opt_lbfgs_forces_batch(bioen_hip_ctx* c, int ntheta, const double* thetas, const double* forces0,
                                     size_t f0_stride, const double* w0, const bioen_lbfgs_config* config,
                                     const bioen_visual_params* visual, int max_batch, double* results,
                                     double* w_opt, bioen_opt_result* infos) {
    if (!c || !thetas || !forces0 || !w0 || !config || !results || !infos || ntheta <= 0)
        return fail(BIOEN_HIP_EINVAL, "NULL argument or ntheta <= 0");
    if (f0_stride != 0 && f0_stride < (size_t)c->m) return fail(BIOEN_HIP_EINVAL, "f0_stride must be 0 or >= m");
    int rc = forces_guard(c);
    if (rc) return rc;
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    const bool verbose = visual && visual->verbose;
    if (verbose) {
        std::printf("L-BFGS minimizer (forces, %d theta value%s, up to %d per matrix pass)\n", ntheta,
                    ntheta > 1 ? "s" : "", std::max(1, std::min(max_batch, (int)kMaxBatch)));
        print_config(*config);
    }
    ResultPinner pin(c->device, w_opt, w_opt ? (size_t)ntheta * c->n_global * sizeof(double) : 0, nullptr, 0);
    ForcesBatchEngine eng(c, *config, verbose);
    rc = eng.run(ntheta, thetas, forces0, f0_stride, w0, max_batch, results, w_opt, infos);
    if (ntheta > 1) c->last_width = 0;       // bioen_hip_last_average: single-problem calls only
    return rc;
}

int bioen_hip_opt_lbfgs_forces(bioen_hip_ctx* c, const double* forces0, const double* w0, double theta,
                               const bioen_lbfgs_config* config, const bioen_visual_params* visual,
                               double* result, double* w_opt, bioen_opt_result* info) {
    if (!info) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    // one problem; the batch width left over serves its line search's speculative trials (engine_forces.inl)
    return bioen_hip_opt_lbfgs_forces_batch(c, 1, &theta, forces0, 0, w0, config, visual, kMaxBatch, result, w_opt, info);
}

// ---- shared ---------------------------------------------------------------------------
int bioen_hip_chi_squared(bioen_hip_ctx* c, const double* w, double* yave, double* chi2) {
    if (!c || !w) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    ProblemSlot& s0 = c->slot[0];
    int rc;
    if ((rc = upload_n(c, s0.w, w))) return rc;       // sharded: this rank's block of the (global) w
    const int one[1] = {0};
    const Round r = make_round(c, one, 1, nullptr, nullptr);
    c->last_width = 1;
    c->last_pos = 0;
    c->last_centered = false;
    Vec8 v{};
    v.p[0] = s0.w;
    int nblk = fwd_strip_blocks(c);
    if (c->mp > 1024 && !c->Yp[0]) nblk = 0;      // row panels are built for the optimizer's passes, not for one product
    if (c->storage) nblk = 0;                     // reduced-storage experiment: its copies are pre-centred; the FP64 matrix serves
    if (nblk > 0 && (rc = ensure_strip_copy(c))) {
        if (!c->strips_unavailable) return rc;
        nblk = 0;
    }
    if (nblk > 0) {            // the strip copy, uncentred (any w, not only normalised ones)
        launch_fwd_strip(c, 1, v, nblk, true);
        launch_fwd_rows_local(c, 1, false, nblk, true);
    } else {
        if ((rc = ensure_rowmajor(c))) return rc;
        launch_fwd_partial(c, 1, v);
        launch_fwd_rows_local(c, 1, false);
    }
    if ((rc = exchange(c, X_YBAR, (size_t)ybar_payload(c, 1, false)))) return rc;   // the ranks' shares of yTilde . w
    launch_rows_combine(c, r, false);
    launch_forces_scalars(c, make_forces_round(c, one, 1, nullptr));   // S_CHI (the KL part is irrelevant here)
    if ((rc = check_launch())) return rc;
    if (yave)   // K = 1: the compact layout is the plain M-vector
        BIOEN_HIP_CHECK(hipMemcpyAsync(yave, c->ybar_c, (size_t)c->m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if ((rc = read_scalars(c))) return rc;
    if (chi2) *chi2 = 0.5 * c->host_scal[S_CHI];
    return 0;
}

int bioen_hip_last_average(bioen_hip_ctx* c, double* yraw, double* yeff) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    if (c->last_width <= 0)
        return fail(BIOEN_HIP_ESTATE, "no single-problem call to take the averages from (a multi-problem call ran last)");
    std::vector<double> raw((size_t)c->m), off((size_t)c->m), sc((size_t)c->m), cen;
    if (c->last_centered) {        // the forces strip passes keep ybar - centre; the centre goes back here
        cen.resize((size_t)c->m);
        BIOEN_HIP_CHECK(hipMemcpyAsync(cen.data(), c->strip_center, (size_t)c->m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    // compact layout of the round that wrote it: ybar_c[row * width + column]
    BIOEN_HIP_CHECK(hipMemcpy2DAsync(raw.data(), sizeof(double), c->ybar_c + c->last_pos,
                                     (size_t)c->last_width * sizeof(double), sizeof(double), (size_t)c->m,
                                     hipMemcpyDeviceToHost, c->stream));
    BIOEN_HIP_CHECK(hipMemcpyAsync(off.data(), c->row_offset, (size_t)c->m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    BIOEN_HIP_CHECK(hipMemcpyAsync(sc.data(), c->row_scale, (size_t)c->m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < c->m; ++i) {
        if (c->last_centered) raw[i] += cen[i];
        if (yraw) yraw[i] = raw[i];
        if (yeff) yeff[i] = off[i] + sc[i] * raw[i];
    }
    return 0;
}

// ---- host self-test -----------------------------------------------------------------------
int bioen_hip_selftest_lbfgs(int kind, int n, const double* x0, const bioen_lbfgs_config* config, double* x_out,
                             bioen_opt_result* info) {
    if (!x0 || !config || !x_out || !info || n <= 0 || kind < 0 || kind > 1)
        return fail(BIOEN_HIP_EINVAL, "bad argument");
    std::memset(info, 0, sizeof *info);
    HostSelftestBackend B(kind, n, x0);
    double fx = 0.0;
    info->lbfgs_code = lbfgs_run(B, n, *config, &fx, &info->iterations, &info->evaluations);
    info->fmin = fx;
    const std::vector<double>& res = B.result_is_trial ? B.x : B.xp;
    std::memcpy(x_out, res.data(), (size_t)n * sizeof(double));
    return 0;
}

// ---- measurement ------------------------------------------------------------------------
int bioen_hip_kernel_stats(bioen_hip_ctx* c, int which, double* total_ms, long long* launches) {
    if (!c || which < 0 || which > 1) return fail(BIOEN_HIP_EINVAL, "bad argument");
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    resolve_timers(c);
    if (total_ms) *total_ms = c->timer.total_ms[which];
    if (launches) *launches = c->timer.launches[which];
    return 0;
}

int bioen_hip_kernel_stats_ex(bioen_hip_ctx* c, int which, double* total_ms, long long* launches,
                              long long* problem_passes) {
    int rc = bioen_hip_kernel_stats(c, which, total_ms, launches);
    if (rc) return rc;
    if (problem_passes) *problem_passes = c->timer.problem_passes[which];
    return 0;
}

int bioen_hip_kernel_stats_reset(bioen_hip_ctx* c) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    resolve_timers(c);
    c->timer.total_ms[0] = c->timer.total_ms[1] = 0.0;
    c->timer.launches[0] = c->timer.launches[1] = 0;
    c->timer.problem_passes[0] = c->timer.problem_passes[1] = 0;
    return 0;
}

int bioen_hip_kernel_stats_enable(bioen_hip_ctx* c, int enable) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    c->timer.enabled = enable != 0;
    return 0;
}

// ---- RCCL (resolved lazily so single-GPU use never needs librccl) ---------------------------
namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                         // optional
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;  // optional
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.h) return 0;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
        if (h) break;
    }
    for (const char* nm : names) {
        if (h) break;
        h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) return fail(BIOEN_HIP_ERCCL, "librccl.so not found");
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(h, "ncclAllGather"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(h, "ncclCommAbort"));
    g_rccl.CommGetAsyncError = reinterpret_cast<decltype(g_rccl.CommGetAsyncError)>(dlsym(h, "ncclCommGetAsyncError"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy)
        return fail(BIOEN_HIP_ERCCL, "librccl.so lacks a required symbol");
    g_rccl.h = h;
    return 0;
}

int rccl_fail(ncclResult_t r, const char* what) {
    std::string s = std::string(what) + " failed: " +
                    (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "(no error string)");
    set_last_error(s);
    return BIOEN_HIP_ERCCL;
}
}  // namespace

}  // extern "C" (reopened below)

namespace bioen {
static int rccl_allgather_inplace(bioen_hip_ctx* c, double* base, size_t count) {
    ncclResult_t r = g_rccl.AllGather(base + (size_t)c->rank * count, base, count, ncclDouble,
                                      static_cast<ncclComm_t>(c->comm), c->stream);
    if (r != ncclSuccess) return rccl_fail(r, "ncclAllGather");
    return 0;
}

// the communicator's asynchronous state (a peer that died, a transport error): polled from the bounded waits
static int rccl_async_error(bioen_hip_ctx* c) {
    if (!c->comm || !g_rccl.CommGetAsyncError) return 0;
    ncclResult_t st = ncclSuccess;
    const ncclResult_t r = g_rccl.CommGetAsyncError(static_cast<ncclComm_t>(c->comm), &st);
    if (r != ncclSuccess) return rccl_fail(r, "ncclCommGetAsyncError");
    if (st != ncclSuccess && st != ncclInProgress) {
        const int rc = rccl_fail(st, "RCCL communicator (asynchronous error)");
        rccl_abort(c);
        return rc;
    }
    return 0;
}

static std::atomic<int> g_comm_init_abandoned{0};      // helper threads of given-up (bounded) ncclCommInitRank calls that are still inside RCCL

static void rccl_abort(bioen_hip_ctx* c) {
    if (!c->comm) return;
    if (g_rccl.CommAbort) g_rccl.CommAbort(static_cast<ncclComm_t>(c->comm));    // frees the communicator as well
    else if (g_rccl.CommDestroy) g_rccl.CommDestroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
}
}  // namespace bioen

extern "C" {

int bioen_hip_comm_unique_id(unsigned char id[128]) {
    if (!id) return fail(BIOEN_HIP_EINVAL, "id is NULL");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId u;
    ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) return rccl_fail(r, "ncclGetUniqueId");
    std::memcpy(id, u.internal, 128);
    return 0;
}

int bioen_hip_comm_init(bioen_hip_ctx* c, const unsigned char id[128], int rank, int nranks) {
    if (!c || !id || nranks <= 0 || rank < 0 || rank >= nranks) return fail(BIOEN_HIP_EINVAL, "bad argument");
    if (c->comm) return fail(BIOEN_HIP_ESTATE, "communicator already initialised");
    if (c->world > 1 && (rank != c->rank || nranks != c->world))
        return fail(BIOEN_HIP_EINVAL, "communicator rank/size must match the context's shard");
    int rc = load_rccl();
    if (rc) return rc;
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, 128);
    // ncclCommInitRank is a collective of all ranks and has no time limit of its own: a rank that never calls it (it failed
    // before, or died) would keep the others here for good.  It runs on a helper thread; this thread waits for it for three
    // times the context's wait bound (topology detection on a full node takes seconds) and then gives the transport up --
    // the helper stays behind, blocked, in a process that goes on without RCCL (the callers fall back).
    // One state word per job decides who owns the outcome (two release / acquire flags read crosswise are a
    // Dekker pattern -- both sides could miss each other, and a late-successful communicator was then neither adopted nor
    // aborted): RUNNING -> DONE by the helper, RUNNING -> ABANDONED by the caller, one sequentially consistent
    // compare-exchange each -- whoever LOSES the exchange knows the other side's move and acts on it.  The process-wide
    // count of helpers still blocked inside RCCL goes up when a job is abandoned and down whenever such a helper returns,
    // successful or not.
    struct InitJob {
        enum { RUNNING = 0, DONE = 1, ABANDONED = 2 };
        ncclComm_t comm = nullptr;
        ncclResult_t r = ncclSuccess;
        std::atomic<int> state{RUNNING};
    };
    auto job = std::make_shared<InitJob>();
    const int dev = c->device;
    try {
        std::thread([job, nranks, u, rank, dev]() {
            if (hipSetDevice(dev) != hipSuccess) (void)hipGetLastError();
            job->r = g_rccl.CommInitRank(&job->comm, nranks, u, rank);
            int expected = InitJob::RUNNING;
            if (job->state.compare_exchange_strong(expected, InitJob::DONE)) return;      // the caller is still waiting: it adopts the result
            // ABANDONED: the caller walked away (and counted this helper as blocked).  A peer that arrived late completes
            // the init now: the orphan is aborted here, so that the peers' collectives on it fail fast instead of waiting
            // for a rank that will never join them; then this helper is no longer inside RCCL, whatever the init returned
            if (job->r == ncclSuccess && job->comm) {
                if (g_rccl.CommAbort) g_rccl.CommAbort(job->comm);
                else if (g_rccl.CommDestroy) g_rccl.CommDestroy(job->comm);
            }
            g_comm_init_abandoned.fetch_sub(1);
        }).detach();
    } catch (...) {
        return fail(BIOEN_HIP_ERCCL, "could not start the thread that initialises the RCCL communicator");
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(3.0 * std::max(c->wait_timeout_s, 1.0));
    while (job->state.load() == InitJob::RUNNING) {
        if (std::chrono::steady_clock::now() > deadline) {
            // A helper thread stays behind, blocked inside RCCL: static destructors of librccl / HIP running under it at
            // interpreter teardown can hang or crash.  The process-wide count (bioen_hip_comm_init_abandoned) tells the
            // host layer to leave through os._exit after flushing its output (bioen_amd/_lib.py: leave_process).
            g_comm_init_abandoned.fetch_add(1);                      // before the exchange: the helper's decrement comes after it
            int expected = InitJob::RUNNING;
            if (job->state.compare_exchange_strong(expected, InitJob::ABANDONED)) {
                char buf[200];
                std::snprintf(buf, sizeof buf, "ncclCommInitRank (rank %d of %d) did not return within %g s: a rank is missing "
                              "(3 x BIOEN_HIP_WAIT_TIMEOUT)", rank, nranks, 3.0 * std::max(c->wait_timeout_s, 1.0));
                return fail(BIOEN_HIP_ERCCL, buf);
            }
            g_comm_init_abandoned.fetch_sub(1);                      // the helper finished in that very moment: its result is ours
            break;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    ncclComm_t comm = job->comm;
    const ncclResult_t r = job->r;
    if (r != ncclSuccess) return rccl_fail(r, "ncclCommInitRank");
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_nranks = nranks;
    return 0;
}

int bioen_hip_comm_init_abandoned(void) { return g_comm_init_abandoned.load() > 0 ? 1 : 0; }

int bioen_hip_comm_allgather(bioen_hip_ctx* c, const double* send, size_t count, double* recv) {
    if (!c || !send || !recv || count == 0) return fail(BIOEN_HIP_EINVAL, "bad argument");
    if (!c->comm) return fail(BIOEN_HIP_ESTATE, "communicator not initialised");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    const size_t need = count * (size_t)(c->comm_nranks + 1);
    if (c->comm_buf_count < need) {
        if (c->comm_buf) hipFree(c->comm_buf);
        c->comm_buf = nullptr;
        c->comm_buf_count = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->comm_buf), need * sizeof(double));
        if (e != hipSuccess) {
            hip_fail(e, "hipMalloc", __FILE__, __LINE__);
            return BIOEN_HIP_ENOMEM;
        }
        c->comm_buf_count = need;
    }
    double* dsend = c->comm_buf;
    double* drecv = c->comm_buf + count;
    BIOEN_HIP_CHECK(hipMemcpyAsync(dsend, send, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    ncclResult_t r = g_rccl.AllGather(dsend, drecv, count, ncclDouble, static_cast<ncclComm_t>(c->comm), c->stream);
    if (r != ncclSuccess) return rccl_fail(r, "ncclAllGather");
    BIOEN_HIP_CHECK(hipMemcpyAsync(recv, drecv, count * c->comm_nranks * sizeof(double), hipMemcpyDeviceToHost,
                                   c->stream));
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    return 0;
}

int bioen_hip_exchange_probe(bioen_hip_ctx* c, size_t count, int reps, double* usec_per_exchange) {
    if (!c || !usec_per_exchange || reps <= 0 || count == 0) return fail(BIOEN_HIP_EINVAL, "bad argument");
    if (count > c->xcap[X_YBAR]) count = c->xcap[X_YBAR];
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    int rc = 0;
    for (int i = 0; i < 5 && !rc; ++i) rc = exchange_raw(c, X_YBAR, count);   // warm-up (connection set-up)
    if (rc) return rc;
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps && !rc; ++i) rc = exchange_raw(c, X_YBAR, count);
    if (rc) return rc;
    BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    if ((rc = transport_error(c))) return rc;
    *usec_per_exchange = 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
    return 0;
}

int bioen_hip_exchange_selftest(bioen_hip_ctx* c, int reps, long long* mismatches) {
    if (!c || !mismatches || reps <= 0) return fail(BIOEN_HIP_EINVAL, "bad argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    unsigned long long* bad = nullptr;
    BIOEN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&bad), sizeof *bad));
    hipError_t e = hipMemsetAsync(bad, 0, sizeof *bad, c->stream);
    int rc = e == hipSuccess ? 0 : hip_fail(e, "memset", __FILE__, __LINE__);
    // payloads of every shape the stages use: odd and even counts, a few doubles to the whole slot, back to back
    const size_t cap = c->xcap[X_YBAR];
    const size_t sizes[] = {1, 2, 3, 42, 169, 336, 1027, cap / 2, cap > 1 ? cap - 1 : 1, cap};
    for (int rep = 0; rep < reps && !rc; ++rep) {
        const size_t payload = std::max<size_t>(1, std::min(cap, sizes[rep % (sizeof sizes / sizeof *sizes)]));
        launch_xch_fill(c, X_YBAR, (int)payload, rep);
        rc = exchange_raw(c, X_YBAR, payload);
        if (!rc) launch_xch_check(c, X_YBAR, (int)payload, rep, bad);
        if (!rc && c->xcap[X_VEC] && rep % 8 == 7) {       // the result gathers' shape: a whole N-vector share per rank
            const size_t pv = rep % 16 == 7 ? c->xcap[X_VEC] : std::max<size_t>(1, c->xcap[X_VEC] - 1);
            launch_xch_fill(c, X_VEC, (int)pv, rep);
            rc = exchange_raw(c, X_VEC, pv);
            if (!rc) launch_xch_check(c, X_VEC, (int)pv, rep, bad);
        }
    }
    unsigned long long h = 0;
    if (!rc) {
        e = hipMemcpyAsync(&h, bad, sizeof h, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = hip_fail(e, "exchange self-test", __FILE__, __LINE__);
    }
    if (!rc) rc = transport_error(c);
    (void)hipFree(bad);
    *mismatches = (long long)h;
    return rc;
}

int bioen_hip_read_probe(bioen_hip_ctx* c, int form, int reps, double* gbytes_per_s, long long* bytes) {
    if (!c || !gbytes_per_s || reps <= 0) return fail(BIOEN_HIP_EINVAL, "bad argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    // form: 0 = whichever is resident (strip copy first), 1 = row-major, 2 = row-sum strips, 4 = column-sum strips
    // (matrices taller than 1024 rows keep their strip copies as row panels: the first panel stands for the form)
    const double* ys = c->Ys ? c->Ys : c->Yp[0];
    const double* ys1 = c->Ys1 ? c->Ys1 : c->Y1p[0];
    const double* src = form == 1 ? c->Y : form == 2 ? ys : form == 4 ? ys1 : (ys ? ys : c->Y);
    if (!src) return fail(BIOEN_HIP_ESTATE, "that form of the matrix is not resident");
    const size_t strip_rows_ = (size_t)(std::min(c->m, 1024) + 15) / 16 * 16;
    const size_t doubles = src == c->Y ? (size_t)c->mp * c->ld : (size_t)(c->ld / 16) * strip_rows_ * 16;
    double* sink = nullptr;
    BIOEN_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&sink), 256 * sizeof(double)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms = 0.f;
    if (e == hipSuccess) {
        launch_read_probe(c, src, doubles, sink);                 // warm-up (TLB, clocks)
        launch_read_probe(c, src, doubles, sink);
        e = hipEventRecord(e0, c->stream);
        for (int i = 0; i < reps; ++i) launch_read_probe(c, src, doubles, sink);
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(sink);
    if (e != hipSuccess) return hip_fail(e, "read probe", __FILE__, __LINE__);
    *gbytes_per_s = (double)doubles * 8.0 * reps / (ms * 1e-3) / 1e9;
    if (bytes) *bytes = (long long)(doubles * 8);
    return 0;
}


// ---- peer-to-peer stage exchange (kernels_p2p.hip) ------------------------------------------------------------
int bioen_hip_p2p_export(bioen_hip_ctx* c, unsigned char handle[64]) {
    if (!c || !handle) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    if (c->p2p_on) return fail(BIOEN_HIP_ESTATE, "peer-to-peer exchange already attached");
    if (!c->p2p_box) {
        size_t cap = 0;
        for (int st = 0; st < X_COUNT; ++st) cap = std::max(cap, c->xcap[st]);
        cap = round_up(cap, 2);
        const size_t doubles = p2p_mailbox_doubles(c->world, cap);
        void* p = nullptr;
        // uncached: a polling wave must see what a peer's store put into this GPU's memory, not a line its L2 kept
        hipError_t e = hipExtMallocWithFlags(&p, doubles * sizeof(double), hipDeviceMallocUncached);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipExtMallocWithFlags(&p, doubles * sizeof(double), hipDeviceMallocFinegrained);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            hip_fail(e, "hipExtMallocWithFlags (mailbox)", __FILE__, __LINE__);
            return BIOEN_HIP_ENOMEM;
        }
        c->p2p_box = static_cast<double*>(p);
        c->p2p_cap = cap;
        c->p2p_bytes = doubles * sizeof(double);
        BIOEN_HIP_CHECK(hipMemsetAsync(c->p2p_box, 0, c->p2p_bytes, c->stream));
        BIOEN_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    hipIpcMemHandle_t h;
    BIOEN_HIP_CHECK(hipIpcGetMemHandle(&h, c->p2p_box));
    std::memcpy(handle, &h, 64);
    return 0;
}

int bioen_hip_p2p_detach(bioen_hip_ctx* c) {
    if (!c) return 0;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    c->p2p_on = 0;
    if (c->failed_p2p) {     // the failure was this transport's own; the stream has drained and the transport goes: the
        c->failed = 0;       // context (and an RCCL communicator it may hold) is usable again
        c->failed_p2p = 0;
        c->fail_msg.clear();
    }
    for (int r = 0; r < 128; ++r)
        if (c->p2p_mapped[r]) {
            hipIpcCloseMemHandle(c->p2p_mapped[r]);
            c->p2p_mapped[r] = nullptr;
        }
    if (c->p2p_peers) hipFree(c->p2p_peers);
    c->p2p_peers = nullptr;
    if (c->p2p_box) hipFree(c->p2p_box);
    c->p2p_box = nullptr;
    if (c->p2p_err) hipHostFree(c->p2p_err);
    c->p2p_err = nullptr;
    if (c->p2p_dev_err) hipFree(c->p2p_dev_err);
    c->p2p_dev_err = nullptr;
    if (c->p2p_cnt) hipFree(c->p2p_cnt);
    c->p2p_cnt = nullptr;
    c->p2p_seq = 0;
    (void)hipGetLastError();
    return 0;
}

int bioen_hip_p2p_attach(bioen_hip_ctx* c, const unsigned char* handles) {
    if (!c || (!handles && c->world > 1)) return fail(BIOEN_HIP_EINVAL, "NULL argument");
    BIOEN_HIP_CHECK(hipSetDevice(c->device));
    if (c->p2p_on) return fail(BIOEN_HIP_ESTATE, "peer-to-peer exchange already attached");
    if (!c->p2p_box) return fail(BIOEN_HIP_ESTATE, "bioen_hip_p2p_export first");
    std::vector<double*> peers((size_t)c->world, nullptr);
    int rc = 0;
    for (int r = 0; r < c->world && !rc; ++r) {
        if (r == c->rank) {
            peers[r] = c->p2p_box;
            continue;
        }
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + (size_t)r * 64, 64);
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            char buf[160];
            std::snprintf(buf, sizeof buf, "hipIpcOpenMemHandle of rank %d's mailbox failed: %s", r, hipGetErrorString(e));
            rc = fail(BIOEN_HIP_ERCCL, buf);
            break;
        }
        c->p2p_mapped[r] = p;
        peers[r] = static_cast<double*>(p);
        // a mailbox on ANOTHER device is only usable where this device reaches that one's memory: ask before the first kernel
        // does (a store the fabric cannot route is a GPU fault, not an error code); where the runtime cannot tell, the
        // self-test decides
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, p) == hipSuccess && at.device != c->device) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, c->device, at.device) == hipSuccess && !can) {
                char buf[160];
                std::snprintf(buf, sizeof buf, "device %d has no peer access to device %d (rank %d's mailbox)", c->device, at.device, r);
                rc = fail(BIOEN_HIP_ERCCL, buf);
                break;
            }
        }
        (void)hipGetLastError();
    }
    if (!rc && !c->p2p_err) {
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&c->p2p_err), 64, hipHostMallocCoherent | hipHostMallocMapped);
        if (e == hipSuccess) {
            std::memset(c->p2p_err, 0, 64);
            e = hipMalloc(reinterpret_cast<void**>(&c->p2p_dev_err), 64);
        }
        if (e == hipSuccess) e = hipMemset(c->p2p_dev_err, 0, 64);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->p2p_cnt), 128 * sizeof(unsigned int));
        if (e == hipSuccess) e = hipMemset(c->p2p_cnt, 0, 128 * sizeof(unsigned int));
        if (e != hipSuccess) rc = hip_fail(e, "error words of the peer-to-peer exchange", __FILE__, __LINE__);
    }
    if (!rc) {
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->p2p_peers), (size_t)c->world * sizeof(double*));
        if (e == hipSuccess)
            e = hipMemcpy(c->p2p_peers, peers.data(), (size_t)c->world * sizeof(double*), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = hip_fail(e, "peer table", __FILE__, __LINE__);
    }
    if (rc) {
        const std::string keep = g_last_error;
        bioen_hip_p2p_detach(c);
        g_last_error = keep;
        return rc;
    }
    c->p2p_seq = 0;
    c->p2p_on = 1;
    return 0;
}

int bioen_hip_exchange_transport(const bioen_hip_ctx* c) {
    if (!c) return -1;
    if (c->p2p_on) return 3;
    if (c->comm) return 1;
    if (c->exchange_cb) return 2;
    return 0;
}

int bioen_hip_exchange_counts3(const bioen_hip_ctx* c, long long* rccl, long long* host_staged, long long* p2p) {
    if (!c) return fail(BIOEN_HIP_EINVAL, "ctx is NULL");
    if (rccl) *rccl = c->n_rccl_exchanges;
    if (host_staged) *host_staged = c->n_host_exchanges;
    if (p2p) *p2p = c->n_p2p_exchanges;
    return 0;
}

int bioen_hip_ctx_set_wait_timeout(bioen_hip_ctx* c, double seconds) {
    if (!c || !(seconds > 0.0)) return fail(BIOEN_HIP_EINVAL, "bad argument");
    c->wait_timeout_s = seconds;
    return 0;
}

int bioen_hip_comm_destroy(bioen_hip_ctx* c) {
    if (!c) return 0;
    if (c->comm && c->failed) rccl_abort(c);      // a destroy would wait for peers that may be gone
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
    if (c->comm_buf) hipFree(c->comm_buf);
    c->comm_buf = nullptr;
    c->comm_buf_count = 0;
    return 0;
}

}  // extern "C"

#include "api_multimin.inl"
