// Peer-to-peer stage exchange over xGMI (gfx950): the all-gather of an exchange stage (ctx.hpp: XStage) as ONE small
// kernel on the context's stream -- stores into the peers' mailboxes, a flag per (peer, exchange), a bounded wait -- in
// place of an ncclAllGather launch.  No reference counterpart: the reference is one process
// (bioen/analyze/procedure.py:62-63 is a serial loop); this is the transport SURVEY 5 sketches ("fused P2P write + flag").
//
// Mailbox of a rank (device memory, allocated uncached so that a polling wave reads what a PEER's store put into HBM):
//     flags : [2 halves][world] u64, one per 64-byte line      flag[h][src] = number of the last exchange of parity h whose
//                                                               segment rank `src` has delivered here (| kAbort: src gave up)
//     data  : [2 halves][world][cap] doubles                   data[h][src] = that segment
// Exchange number s (1, 2, ...; every rank issues the same exchanges in the same order) uses half s & 1.  Block p of rank r:
//     1. copies r's segment (stage buffer, written by the previous kernel on this stream) into data[s&1][r] of PEER p,
//     2. every thread fences at system scope, the block meets, one lane releases s into flag[s&1][r] of peer p,
//     3. one lane polls flag[s&1][p] of its OWN mailbox until it reaches s (relaxed system-scope loads, s_sleep between;
//        bounded by `timeout` ticks of the 100 MHz wall clock), the block meets, fences (acquire),
//     4. copies data[s&1][p] of its own mailbox into the stage buffer's segment p.
// Why two halves are enough: r stores exchange s + 2 into the half p read for s only after r has seen p's flag s + 1, which p
// released after its kernel of exchange s -- copy-out included -- had finished (stream order on p).
// Beside its flag (second word of the flag's line) the sender leaves {stage, payload} of the exchange; a receiver that finds
// another stage or size than its own there is OUT OF STEP with that peer (the ranks composed different rounds): an error
// like the two below, instead of a stage buffer full of somebody else's numbers.
// A wait that expires, or an ABORT flag from the peer, makes the kernel record {stage, peer, exchange} in the host-mapped
// error word (the host's waits poll it) and in a device word; from then on every exchange of this rank delivers ABORT flags
// and waits for nothing: the failure travels at flag speed and the stream drains.
#include "device_utils.hpp"

#include <algorithm>
#include <cstdlib>

namespace bioen {

constexpr unsigned long long kP2PAbort = 1ull << 63;
constexpr int kP2PFlagStride = 8;      // u64 per flag: one 64-byte line each

struct P2PArgs {
    double* const* peers;              // [world] mailbox bases as mapped in this process
    double* local;                     // stage buffer [world][payload]
    unsigned long long* err_host;      // host-mapped
    unsigned long long* err_dev;
    unsigned long long seq;
    unsigned long long timeout;        // 100 MHz ticks
    size_t cap;
    int payload, world, rank, stage;
};

__device__ __forceinline__ unsigned long long p2p_tag(const P2PArgs& q) {   // what an exchange is: its stage and its size
    return ((unsigned long long)(q.stage & 0xff) << 32) | (unsigned int)q.payload;
}

__device__ __forceinline__ size_t p2p_data_offset(int world) {       // doubles in front of the data halves
    return (size_t)2 * world * kP2PFlagStride;
}

__global__ __launch_bounds__(1024) void k_p2p_exchange(P2PArgs q) {
    const int p = blockIdx.x;
    if (p == q.rank) return;                                        // own segment: already in place
    __shared__ unsigned long long verdict;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int half = (int)(q.seq & 1ull);
    unsigned long long failed_before = 0;
    if (tid == 0) failed_before = *q.err_dev;                        // in flight while the segment is copied
    {   // 1. this rank's segment -> peer p
        const double* src = q.local + (size_t)q.rank * q.payload;
        double* dst = q.peers[p] + p2p_data_offset(q.world) + ((size_t)half * q.world + q.rank) * q.cap;
        if (((q.payload | (int)(q.cap & 1)) & 1) == 0 && (((size_t)q.rank * q.payload) & 1) == 0) {
            for (int i = tid; i < q.payload / 2; i += nt)
                *reinterpret_cast<d2*>(dst + 2 * i) = *reinterpret_cast<const d2*>(src + 2 * i);
        } else {
            for (int i = tid; i < q.payload; i += nt) dst[i] = src[i];
        }
    }
    __threadfence_system();                                         // 2. every storing thread: its stores have left
    __syncthreads();
    if (tid == 0) {
        unsigned long long* flag = reinterpret_cast<unsigned long long*>(q.peers[p]) +
                                   ((size_t)half * q.world + q.rank) * kP2PFlagStride;
        __hip_atomic_store(flag + 1, p2p_tag(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // ordered by the release below
        __hip_atomic_store(flag, failed_before ? (kP2PAbort | q.seq) : q.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        unsigned long long v = 0, why = failed_before ? 3 : 0;     // 3: an earlier exchange failed (nothing to wait for)
        if (!failed_before) {                                       // 3. peer p's segment
            const unsigned long long* mine = reinterpret_cast<const unsigned long long*>(q.peers[q.rank]) +
                                             ((size_t)half * q.world + p) * kP2PFlagStride;
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (v & kP2PAbort) { why = 2; break; }
                if (v >= q.seq) break;
                if (wall_clock64() - t0 > q.timeout) { why = 1; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            if (!why && v == q.seq) {             // 4: the peer's exchange of this number is another one than ours
                __threadfence_system();           // (the tag was stored before the flag's release: read it behind an acquire)
                if (__hip_atomic_load(mine + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != p2p_tag(q)) why = 4;
            }
            if (why) {      // first failure wins; {why:4 | stage:8 | peer:12 | exchange:40}
                const unsigned long long word = (why << 60) | ((unsigned long long)(q.stage & 0xff) << 52) |
                                                ((unsigned long long)(p & 0xfff) << 40) | (q.seq & 0xffffffffffull);
                if (atomicCAS(q.err_dev, 0ull, word) == 0ull)
                    __hip_atomic_store(q.err_host, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        verdict = why;
    }
    __syncthreads();
    if (verdict) return;
    __threadfence_system();                                         //    acquire: nothing of the segment is cached here
    {   // 4. peer p's segment -> stage buffer
        const double* src = q.peers[q.rank] + p2p_data_offset(q.world) + ((size_t)half * q.world + p) * q.cap;
        double* dst = q.local + (size_t)p * q.payload;
        if (((q.payload | (int)(q.cap & 1)) & 1) == 0 && (((size_t)p * q.payload) & 1) == 0) {
            for (int i = tid; i < q.payload / 2; i += nt)
                *reinterpret_cast<d2*>(dst + 2 * i) = *reinterpret_cast<const d2*>(src + 2 * i);
        } else {
            for (int i = tid; i < q.payload; i += nt) dst[i] = src[i];
        }
    }
}

// The same exchange for LARGE segments (the result gathers: a rank's share of an N-vector, 1 MB at the headline on 8 GPUs,
// 4 MB on 2): `nsub` blocks per peer, each moving its slice.  The flag for peer p may only go up when ALL of them have
// stored and fenced: every sub-block counts itself in on a device counter after its fence, the one that completes the count
// resets it and releases the flag; every sub-block then polls the flag of ITS peer and copies its slice in.  Exchanges are
// serialised on the stream, so one counter per peer serves them all.
__global__ __launch_bounds__(1024) void k_p2p_exchange_big(P2PArgs q, unsigned int* cnt) {
    const int p = blockIdx.x, sub = blockIdx.y, nsub = gridDim.y;
    if (p == q.rank) return;
    __shared__ unsigned long long verdict;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int half = (int)(q.seq & 1ull);
    // this sub-block's slice, in 16-byte pairs where the payload allows it (it does for every vector gather: ld is even)
    const bool wide = ((q.payload | (int)(q.cap & 1)) & 1) == 0;
    const int units = wide ? q.payload / 2 : q.payload;
    const int per = (units + nsub - 1) / nsub;
    const int u0 = min(units, sub * per), u1 = min(units, u0 + per);
    unsigned long long failed_before = 0;
    if (tid == 0) failed_before = *q.err_dev;
    {
        const double* src = q.local + (size_t)q.rank * q.payload;
        double* dst = q.peers[p] + p2p_data_offset(q.world) + ((size_t)half * q.world + q.rank) * q.cap;
        if (wide) {
            for (int i = u0 + tid; i < u1; i += nt) *reinterpret_cast<d2*>(dst + 2 * i) = *reinterpret_cast<const d2*>(src + 2 * i);
        } else {
            for (int i = u0 + tid; i < u1; i += nt) dst[i] = src[i];
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        const unsigned int before = __hip_atomic_fetch_add(cnt + p, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before == (unsigned int)nsub - 1u) {                        // the peer's whole segment has left
            __hip_atomic_store(cnt + p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long* flag = reinterpret_cast<unsigned long long*>(q.peers[p]) +
                                       ((size_t)half * q.world + q.rank) * kP2PFlagStride;
            __hip_atomic_store(flag + 1, p2p_tag(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(flag, failed_before ? (kP2PAbort | q.seq) : q.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        unsigned long long v = 0, why = failed_before ? 3 : 0;
        if (!failed_before) {
            const unsigned long long* mine = reinterpret_cast<const unsigned long long*>(q.peers[q.rank]) +
                                             ((size_t)half * q.world + p) * kP2PFlagStride;
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (v & kP2PAbort) { why = 2; break; }
                if (v >= q.seq) break;
                if (wall_clock64() - t0 > q.timeout) { why = 1; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            if (!why && v == q.seq) {             // 4: the peer's exchange of this number is another one than ours
                __threadfence_system();           // (the tag was stored before the flag's release: read it behind an acquire)
                if (__hip_atomic_load(mine + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != p2p_tag(q)) why = 4;
            }
            if (why) {
                const unsigned long long word = (why << 60) | ((unsigned long long)(q.stage & 0xff) << 52) |
                                                ((unsigned long long)(p & 0xfff) << 40) | (q.seq & 0xffffffffffull);
                if (atomicCAS(q.err_dev, 0ull, word) == 0ull)
                    __hip_atomic_store(q.err_host, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        verdict = why;
    }
    __syncthreads();
    if (verdict) return;
    __threadfence_system();
    {
        const double* src = q.peers[q.rank] + p2p_data_offset(q.world) + ((size_t)half * q.world + p) * q.cap;
        double* dst = q.local + (size_t)p * q.payload;
        if (wide) {
            for (int i = u0 + tid; i < u1; i += nt) *reinterpret_cast<d2*>(dst + 2 * i) = *reinterpret_cast<const d2*>(src + 2 * i);
        } else {
            for (int i = u0 + tid; i < u1; i += nt) dst[i] = src[i];
        }
    }
}

// self-test (bioen_hip_exchange_selftest): a rank's segment of exchange `rep` is a pattern of (rank, rep, index); after the
// exchange every segment of the stage buffer must hold its owner's pattern.  Queued back to back, no host in between.
__device__ __forceinline__ double p2p_pattern(int rank, int rep, int i) {
    return (double)rank * 1048576.0 + (double)rep * 4096.0 + (double)(i % 4093) + 0.25;
}
__global__ void k_xch_fill(double* seg, int payload, int rank, int rep) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < payload; i += gridDim.x * blockDim.x)
        seg[i] = p2p_pattern(rank, rep, i);
}
__global__ void k_xch_check(const double* base, int payload, int world, int rep, unsigned long long* bad) {
    unsigned long long mine = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)payload * world; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / payload), k = (int)(i - (size_t)r * payload);
        if (base[i] != p2p_pattern(r, rep, k)) ++mine;
    }
    if (mine) atomicAdd(bad, mine);
}
// mirror exchange (bioen_hip_ctx_set_mirror_exchange): block p copies this rank's part of the stage onto rank p's
__global__ void k_xch_mirror(double* base, int payload, int rank) {
    const int p = blockIdx.x;
    if (p == rank) return;
    const double* src = base + (size_t)rank * payload;
    double* dst = base + (size_t)p * payload;
    for (int i = threadIdx.x; i < payload; i += blockDim.x) dst[i] = src[i];
}
void launch_xch_mirror(bioen_hip_ctx* c, int stage, size_t payload) {
    const int threads = (int)std::min<size_t>(1024, std::max<size_t>(64, (payload + 63) / 64 * 64));
    hipLaunchKernelGGL(k_xch_mirror, dim3(c->world), dim3(threads), 0, c->stream, c->xbuf[stage], (int)payload, c->rank);
}

void launch_xch_fill(bioen_hip_ctx* c, int stage, int payload, int rep) {
    hipLaunchKernelGGL(k_xch_fill, dim3(8), dim3(256), 0, c->stream, c->xbuf[stage] + (size_t)c->rank * payload, payload, c->rank, rep);
}
void launch_xch_check(bioen_hip_ctx* c, int stage, int payload, int rep, unsigned long long* bad) {
    hipLaunchKernelGGL(k_xch_check, dim3(16), dim3(256), 0, c->stream, c->xbuf[stage], payload, c->world, rep, bad);
}

size_t p2p_mailbox_doubles(int world, size_t cap) {
    return (size_t)2 * world * kP2PFlagStride + (size_t)2 * world * cap;
}

void launch_p2p_exchange(bioen_hip_ctx* c, int stage, size_t payload) {
    P2PArgs q{};
    q.peers = c->p2p_peers;
    q.local = c->xbuf[stage];
    q.err_host = c->p2p_err;
    q.err_dev = c->p2p_dev_err;
    q.seq = ++c->p2p_seq;
    q.timeout = (unsigned long long)(c->wait_timeout_s * 1e8);
    q.cap = c->p2p_cap;
    q.payload = (int)payload;
    q.world = c->world;
    q.rank = c->rank;
    q.stage = stage;
    static long big = -1;                  // doubles from which a segment is moved by several blocks per peer
    if (big < 0) {
        const char* e = std::getenv("BIOEN_HIP_P2P_BIG");
        big = e ? std::max(1L, std::atol(e)) : 16384;
    }
    if ((long)payload >= big && c->p2p_cnt) {
        const int nsub = (int)std::min<size_t>(16, std::max<size_t>(2, payload / 8192));
        hipLaunchKernelGGL(k_p2p_exchange_big, dim3(c->world, nsub), dim3(1024), 0, c->stream, q, c->p2p_cnt);
        return;
    }
    // a block per peer; as many threads as the segment has 16-byte pieces, within [64, 1024]
    int threads = (int)std::min<size_t>(1024, std::max<size_t>(64, (payload / 2 + 63) / 64 * 64));
    hipLaunchKernelGGL(k_p2p_exchange, dim3(c->world), dim3(threads), 0, c->stream, q);
}

}  // namespace bioen
