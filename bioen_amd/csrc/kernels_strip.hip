// Forces method, M <= 1024 (k_strip: M <= 512; k_strip2: 512 < M <= 1024): the whole evaluation in TWO passes over a strip-major copy of yTilde, with both
// products of each pass on the FP64 matrix cores (v_mfma_f64_4x4x4_4b_f64).
//
// Reference: _get_weights_from_forces (c_bioen_kernels_forces.c:111-224), _bioen_log_posterior_forces
// (:227-277), _grad_bioen_log_posterior_forces (:280-340) -- five passes over the matrix there.
//
// Layout.  The row-major matrix serves the log-weights kernels (whole rows / 128-column strips of rows
// stream at 6.8 TB/s).  The forces evaluation needs whole COLUMNS (x_j = sum_i Y_ij f_i) and whole ROWS
// (ybar_i = sum_j Y_ij e_j) of the same data in one pass, i.e. a block must hold all rows of a few
// columns: 128-byte row segments 8 MB apart in the row-major matrix, which reach 4.9 TB/s at best (r01).
// So the matrix passes for M <= 1024 (log-weights: any M, over row panels of <= 1024 rows) read strip-major copies,
// built on first use:
//     Ys[strip s][row][c ^ swz(row)] = Y[row][16 s + c]          (raw numbers; rows padded to 16)
// * strip-major: the 16 columns x all rows a block works on are ONE contiguous chunk (64 KB at
//   M = 512) -- every wave-load is a contiguous KiB, as in the streaming kernels;
// * the kernels subtract center = YTilde (the targets, identical on every rank of a sharded context) from every
//   operand on its way from the load registers into the products: Y' = Y - center.  The softmax is invariant under
//   x_j -> x_j + const, ybar_i = center_i + sum_j Y'_ij w_j, the adjoint picks up the constant
//   B0 = sum_i center_i r_i, and the reference's centred gradient sum becomes
//       sum_j (Y_ij - ybar_i) t_j  =  sum_j Y'_ij t_j  -  (ybar_i - center_i) sum_j t_j
//   with BOTH terms at the scale of the data's spread instead of its offset: the plain matrix product
//   the matrix cores compute loses nothing to cancellation, and no per-problem centring is needed
//   inside the product.  (r02 stored Y' in the copies; the raw copies of r03 give the same operands -- the same
//   subtraction, a register later -- and let the copies REPLACE the row-major matrix: read_ytilde is exact from them.)
// * the XOR swizzle (columns permuted by bits 1..4 of the row) makes the LDS image of a strip -- a plain
//   copy, 16 doubles per row, no padding -- conflict-free for both operand fetch patterns below.
//
// Kernel (K = batch width as a template parameter, both passes from one template):
//   a wave owns 64 rows of the strip: it prefetches them two strips ahead (2 x 8 KiB in registers: 128 KB
//   in flight per CU), copies them to its slice of the LDS tile and is the only reader of that slice (no
//   block barrier around the tile);
//   P1  column sums  D1[c][k] = sum_i Y'[i][c] u[i][k]:  16 x ceil(K/4) matrix instructions per wave; A = 4 rows
//       x 16 columns from the tile, B = 4 rows x 4 problems of u = forces | residuals from an LDS table;
//       the waves' partial D1 meet in LDS                                                        -> barrier
//   P2  16 K threads (a problem's 16 columns in one 16-lane group): xy: x_j out, online softmax (running
//       maximum per block), e_j;   bt: t_j = (theta (1 + log w_j/w0_j) + b_j) w_j                 -> barrier
//   P3  row sums  D3[i][k] += sum_c Y'[i][c] v[c][k]:  16 x ceil(K/4) matrix instructions per wave into
//       persistent accumulators; A = 16 rows x 4 columns from the tile (fetched BEFORE the barriers: it does
//       not depend on P2), B = v (e | t) from LDS.
//   The 4x4x4 four-block form computes exactly the K <= 4 (or 8) problems -- the 16x16x4 form pads them to 16
//   at the same 32 FLOP/clk/SIMD, which is also the vector ALU's FP64 rate -- and delivers the cross-lane sums
//   of P1 without a single shuffle; all operand fetches of a phase are issued before its first instruction.
//   Measured (r02, N = 1e6 x M = 512, 4.1 GB per pass): 0.60 / 0.62 ms per pass at K = 1 (6.7 TB/s), 0.68 /
//   0.78 ms at K = 8, against 0.81 / 0.85 ms and 2.28 / 1.61 ms for the r01 kernels on the row-major matrix.
#include <atomic>

#include "device_utils.hpp"

#ifndef STRIP_WAVES_PER_SIMD
#define STRIP_WAVES_PER_SIMD 2
#endif
#ifndef STRIP_DEPTH
#define STRIP_DEPTH 2      // strips in flight per wave (register sets)
#endif
#ifndef ADJ_DIAG
#define ADJ_DIAG 0      // diagnostic builds: 1 = no output stores in k_strip_adj (timing only: results are garbage)
#endif
#ifndef STRIP_DIAG
#define STRIP_DIAG 0      // diagnostic builds: 1 = no MFMAs, 2 = no matrix loads
#endif
#ifndef FWD_DIAG
#define FWD_DIAG 0        // diagnostic builds of k_strip_fwd (timing only, results are garbage): 1 = the blocks sweep the copy as ONE
#endif                    // front (slot s takes strips s, s + slots, ...: the r04 pattern), 2 = no chunk fold
#ifndef STRIP_PRECENTERED
#define STRIP_PRECENTERED 0   // diagnostic builds: 1 = the r02 layout (copies hold Y - centre, no subtraction in the kernels;
#endif                        // read_ytilde is then off by the centre): A/B of what the in-kernel centring costs

namespace bioen {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kStripCols = 16;
constexpr int kWaveRows = 64;

// physical column of logical column c in row r:  c ^ strip_swz(r)
__device__ __forceinline__ int strip_swz(int row) { return (((row >> 1) & 7) << 1) ^ ((row >> 4) & 1); }

// Where strip s of the row-sum order copy lives (r06).  A context that holds ilv > 1 canonical segments (one GPU: all
// eight) stores the strips of its segments INTERLEAVED: strip r of local segment v at position r ilv + v.  The row-sum
// passes give every slot the strips g, g + gs, ... of ONE segment (kernels.hpp: StripSets), so in segment order the 256
// slots of a launch read eight windows of the copy a segment (1 GB at the headline) apart; interleaved, the same slots at
// the same moment read ONE contiguous window, as the whole-matrix sweep of r04 did -- which strips a set is summed over,
// and in which order, does not change (same bits), only where they lie.  Measured (tools/pass_probe.py, profiles/
// r06_fwd_ab.txt): 1.5-3 % of the forward pass on boxes whose memory system does not mind the eight windows, 7 % on those
// that do (r04's kernel 1.155-1.169 ms in the headline sweep against 1.247-1.253 ms for r05's on the same box).
__device__ __forceinline__ int strip_phys(int s, int sps, int ilv) {
    if (ilv <= 1) return s;
    const int v = s / sps;
    return (s - v * sps) * ilv + v;
}

// ---- one-time construction of the strip-major copy -------------------------------------------------
// Within a strip the 64-row slice of wave w is stored in the order the ROW-SUM product wants its matrix
// operand, so a wave-load (1 KiB, 16 B per lane) lands in the operand registers with no further movement:
//   chunk i = 2 h + qp (h = 16-row block 0..3, qp = column octet 0..1), lane l = 16 lq + lr:
//     .x = Y'[64 w + 16 h + lr][8 qp + lq]        (operand of column quad qq = 2 qp)
//     .y = Y'[64 w + 16 h + lr][8 qp + 4 + lq]    (                      qq = 2 qp + 1)
// The column-sum product reads the same data through an LDS image (row-major, 16 doubles per row, columns
// XOR-swizzled by strip_swz(row)), which the waves fill from those registers.
__device__ __forceinline__ size_t strip_pos(int row, int col) {          // index inside a strip (doubles)
    const int w = row >> 6, h = (row >> 4) & 3, lr = row & 15;
    const int qp = col >> 3, hi = (col >> 2) & 1, lq = col & 3;
    return ((size_t)((w * 4 + h) * 2 + qp) * 64 + (lq * 16 + lr)) * 2 + hi;
}

// The same strips in the operand order of the COLUMN-SUM product (the log-weights adjoint streams this one):
//   chunk i (row groups 2 i, 2 i + 1 of the wave's 64 rows), lane l = 16 lq + lr:
//     .x = Y'[64 w + 8 i + lq][lr]      .y = Y'[64 w + 8 i + 4 + lq][lr]
__device__ __forceinline__ size_t strip_pos_colsum(int row, int col) {
    const int w = row >> 6, g = (row >> 2) & 15, lq = row & 3;
    return ((size_t)(w * 8 + (g >> 1)) * 64 + (lq * 16 + col)) * 2 + (g & 1);
}

// The copies hold the RAW matrix (r03; r02 stored Y - centre): the centring is applied to the operand registers
// inside the kernels -- the same subtraction, hence the same bits in every product -- so that the strip copies can
// REPLACE the row-major matrix instead of standing beside it: bioen_hip_ctx_read_ytilde gathers the caller's numbers
// back out of them bit for bit, and the row-major copy is freed once the row-sum copy exists (ctx.hpp: Y).
template <bool COLSUM>
__global__ __launch_bounds__(256) void k_build_strips(const double* __restrict__ Y, size_t ld, int mp, int mps, int n,
                                                      double* __restrict__ Ys, int nstrips,
                                                      const double* __restrict__ center_diag, int sps, int ilv) {
    for (int s = blockIdx.x; s < nstrips; s += gridDim.x) {
        double* dst = Ys + (size_t)strip_phys(s, sps, ilv) * mps * kStripCols;
        for (int p = threadIdx.x; p < mps * 8; p += 256) {
            const int row = p >> 3, part = p & 7;
            d2 v{0.0, 0.0};
            if (row < mp) {
                const size_t col = (size_t)s * kStripCols + part * 2;
                v = *reinterpret_cast<const d2*>(Y + (size_t)row * ld + col);
                v.x = col < (size_t)n ? v.x : 0.0;
                v.y = col + 1 < (size_t)n ? v.y : 0.0;
#if STRIP_PRECENTERED
                if (col < (size_t)n) v.x -= center_diag[row];
                if (col + 1 < (size_t)n) v.y -= center_diag[row];
#endif
            }
            dst[COLSUM ? strip_pos_colsum(row, part * 2) : strip_pos(row, part * 2)] = v.x;
            dst[COLSUM ? strip_pos_colsum(row, part * 2 + 1) : strip_pos(row, part * 2 + 1)] = v.y;
        }
    }
}

// row-sum order copy -> column-sum order copy (the log-weights adjoint's), strip by strip through LDS-free index maps
__global__ __launch_bounds__(256) void k_restripe(const double* __restrict__ Ys, int mps, double* __restrict__ Ys1,
                                                  int nstrips, int sps, int ilv) {
    for (int s = blockIdx.x; s < nstrips; s += gridDim.x) {
        const double* src = Ys + (size_t)strip_phys(s, sps, ilv) * mps * kStripCols;    // (the column-sum copy: strip order)
        double* dst = Ys1 + (size_t)s * mps * kStripCols;
        for (int p = threadIdx.x; p < mps * kStripCols; p += 256) {
            const int row = p >> 4, col = p & 15;
            dst[strip_pos_colsum(row, col)] = src[strip_pos(row, col)];
        }
    }
}

// row-sum order copy, segments interleaved by ilv_from -> the same strips interleaved by ilv_to (whole strips move)
__global__ __launch_bounds__(256) void k_relayout(const double* __restrict__ from, double* __restrict__ to, int mps, int nstrips,
                                                  int sps, int ilv_from, int ilv_to) {
    for (int s = blockIdx.x; s < nstrips; s += gridDim.x) {
        const d2* src = reinterpret_cast<const d2*>(from + (size_t)strip_phys(s, sps, ilv_from) * mps * kStripCols);
        d2* dst = reinterpret_cast<d2*>(to + (size_t)strip_phys(s, sps, ilv_to) * mps * kStripCols);
        for (int p = threadIdx.x; p < mps * (kStripCols / 2); p += 256) dst[p] = src[p];
    }
}

// row-sum order copy -> row-major block out[rows][cols] (device), rows [row0, row0 + rows), columns [col0, col0 + cols)
__global__ __launch_bounds__(256) void k_gather_strips(const double* __restrict__ Ys, int mps, int row0, int rows,
                                                       size_t col0, int cols, double* __restrict__ out, size_t ldo,
                                                       int sps, int ilv) {
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < (size_t)rows * cols; p += (size_t)gridDim.x * 256) {
        const int r = (int)(p / cols);
        const size_t cc = col0 + (p - (size_t)r * cols);
        out[(size_t)r * ldo + (p - (size_t)r * cols)] =
            Ys[(size_t)strip_phys((int)(cc / kStripCols), sps, ilv) * mps * kStripCols + strip_pos(row0 + r, (int)(cc % kStripCols))];
    }
}

// A wave loads its 64-row slice of a strip as 8 chunks of 1 KiB; chunks 2 h and 2 h + 1 hold row block h (16 rows)
// in both operand orders.  Row blocks beyond the strip's last one (mps is a multiple of 16, not of 64) do not exist in
// the copy: their chunks are redirected to the slice's first row block.  Offsets in doubles, wave-uniform.
__device__ __forceinline__ void strip_chunk_offsets(int mps, int rsrc, int (&off)[kWaveRows / 8]) {
    const int nh = min(kWaveRows / 16, (mps - rsrc) / 16);       // row blocks of this wave's slice
#pragma unroll
    for (int i = 0; i < kWaveRows / 8; ++i) off[i] = __builtin_amdgcn_readfirstlane(((i >> 1) < nh ? i : (i & 1)) * 128);
}

// ---- reduced-byte storage EXPERIMENT (r04; SURVEY 7 "treat FP32/BF16-split as an experiment", 8 f4; never the default,
// never the headline): the log-weights matrix passes can stream copies that hold the CENTRED operand Y' = Y - centre as
//   STORE 1: fp32 high part + bf16 residual (6 bytes per element, |error| <= 2^-33 |Y'|), reassembled in FP64 registers,
//   STORE 2: fp32 (4 bytes, 2^-25 |Y'|),
// in the same operand orders.  A wave's 64-row slice of a strip (rows padded to 64 here) is one contiguous run: four
// 1-KiB loads of the high parts -- load u, lane l: {chunk 2u .x, .y, chunk 2u+1 .x, .y} of the FP64 layout above -- and,
// STORE 1, two 1-KiB loads of the residuals behind them -- load v, lane l, word w: chunk 4v + w, .x in the low half,
// .y in the high half.  Every load is 16 bytes per lane as in the FP64 stream; measured, tools/split_read_probe.hip:
// 6.87 TB/s for the 6-byte stream reassembled to FP64 = 1.31 x the elements per second of the 8-byte stream, 2.0 x for fp32.
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int STORE>
__host__ __device__ constexpr int reduced_slice_bytes() { return STORE == 1 ? 6144 : 4096; }

template <bool NT, class T>
__device__ __forceinline__ T ldg16(const void* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const T*>(p));
    return *reinterpret_cast<const T*>(p);
}
// element (.x | .y) of chunk i of the wave's slice(s), back in FP64; NH loads of high parts (chunks 2u, 2u + 1 in load u),
// NH / 2 loads of residuals (chunks 4v .. 4v + 3 in load v) -- NH = 4: one 64-row slice, 8: the two slices of 128 rows
template <int STORE, int NH>
__device__ __forceinline__ double reduced_elem(const f4 (&hi)[NH], const u4 (&lo)[NH / 2], int i, int xy) {
    const f4 h = hi[i >> 1];
    const int e = (i & 1) * 2 + xy;
    double d = (double)(e == 0 ? h.x : e == 1 ? h.y : e == 2 ? h.z : h.w);
    if (STORE == 1) {
        const u4 l = lo[i >> 2];
        const int wsel = i & 3;
        const unsigned w = wsel == 0 ? l.x : wsel == 1 ? l.y : wsel == 2 ? l.z : l.w;
        d += (double)__uint_as_float(xy ? (w & 0xffff0000u) : (w << 16));
    }
    return d;
}
// a wave's strip in flight: FP64 chunks, or the reduced formats' loads (the members a format does not use never exist)
template <int NCH>
struct StripRegs {
    d2 v[NCH];
    f4 hi[NCH / 2];
    u4 lo[NCH / 4];
};

// row-major matrix -> reduced strip copy (centred); COLSUM selects the operand order (strip_pos / strip_pos_colsum)
template <bool COLSUM, int STORE>
__global__ __launch_bounds__(256) void k_build_strips_reduced(const double* __restrict__ Y, size_t ld, int mp, int mps64, int n,
                                                              unsigned char* __restrict__ out, int nstrips,
                                                              const double* __restrict__ center) {
    constexpr int SB = reduced_slice_bytes<STORE>();
    const size_t strip_bytes = (size_t)(mps64 / kWaveRows) * SB;
    for (int s = blockIdx.x; s < nstrips; s += gridDim.x) {
        unsigned char* dst = out + (size_t)s * strip_bytes;
        for (int p = threadIdx.x; p < mps64 * kStripCols; p += 256) {
            const int row = p >> 4, cc = p & 15;
            const size_t col = (size_t)s * kStripCols + cc;
            double v = 0.0;
            if (row < mp && col < (size_t)n) v = Y[(size_t)row * ld + col] - center[row];
            const size_t pos = COLSUM ? strip_pos_colsum(row, cc) : strip_pos(row, cc);     // ((wave 8 + chunk) 64 + lane) 2 + xy
            const int xy = (int)(pos & 1), lane = (int)((pos >> 1) & 63), chunk = (int)((pos >> 7) & 7), w = (int)(pos >> 10);
            unsigned char* sl = dst + (size_t)w * SB;
            const float hi = (float)v;
            reinterpret_cast<float*>(sl)[((chunk >> 1) * 64 + lane) * 4 + (chunk & 1) * 2 + xy] = hi;
            if (STORE == 1) {
                const float r = (float)(v - (double)hi);
                unsigned b = __float_as_uint(r);
                b += 0x7fffu + ((b >> 16) & 1u);                        // bf16, round to nearest even
                reinterpret_cast<unsigned short*>(sl + 4096)[(((chunk >> 2) * 64 + lane) * 4 + (chunk & 3)) * 2 + xy] =
                    (unsigned short)(b >> 16);
            }
        }
    }
}

struct StripArgs {
    const double* Ys;       // strip-major copy (raw matrix; the reduced formats: bytes, centred)
    const double* center;   // mp values subtracted from the rows on the way into the products (a zero vector: none)
    int mps;                // rows of a strip (multiple of 16)
    int mp;                 // rows of the operands u_c / outputs
    int nstrips;
    int n;                  // valid columns
    int K;
    int nblk;               // k_strip_adj: strip slots of the launch
    int wps, spb;           // k_strip_fwd / k_strip_adj: waves per strip slot, strips per block and iteration
    // canonical partial sets (kernels.hpp: StripSets) of the row-sum passes: k_strip_fwd, k_strip, k_strip2
    int sps, gs, tc, nch, fold, slots;
    int nslots;             // physical slots of the launch = slots x local segments (forces passes: = gs)
    int nlocal;             // local segments (forces passes: a block runs its group through all of them)
    int ilv;                // row-sum order FP64 copies: segments interleaved by this many (strip_phys); <= 1: strip order
    const double* u_c;      // [row * K + k]: forces (xy) | residuals (bt)
    const double* w0;
    double* partial;        // [block * mp K + row * K + k]  (transposed: device_utils.hpp, tiles_sum16)
    int pstride;            // k_strip_fwd: rows of the partial layout (= mp; a row panel of a taller matrix: the matrix's)
    int accumulate;         // k_strip_adj: 0 = start at `shift`, 1 = add to the outputs of the panels before this one, 2 = start at 0
    long long* stamps;      // diagnostic builds (STRIP_DIAG & 4): per wave 8 phase-cycle sums
};

// What physical slot `ps` of a row-sum pass works on (kernels.hpp: StripSets): the strips first, first + gs, ... (count
// of them) of local segment v = ps / slots -- a whole group (fold), or chunk cg of group g (slot r = cg gs + g of the
// segment: consecutive blocks read consecutive strips) -- and the set its sums go to.  count = 0: an empty chunk (a
// short group's last one) or a slot beyond the launch; its set is all zeros.
struct SlotWork {
    int first, count, set;
    int nsets, set_stride;      // forces passes: the sets the slot writes (set, set + set_stride, ...: one per chunk)
    bool live;
};
__device__ __forceinline__ SlotWork strip_slot(const StripArgs& q, int ps) {
    SlotWork w;
    w.live = ps < q.nslots;
    const int pss = w.live ? ps : 0;
    const int v = pss / q.slots, r = pss - v * q.slots;
    const int cg = r / q.gs, g = r - cg * q.gs;                 // fold: cg = 0
    const int tg = (q.sps - g + q.gs - 1) / q.gs;               // strips of group g (g < gs <= sps: at least one)
    const int t0 = q.fold ? 0 : cg * q.tc;
    const int t1 = q.fold ? tg : min(tg, t0 + q.tc);
    w.first = v * q.sps + g + q.gs * t0;
    w.count = (w.live && t1 > t0) ? t1 - t0 : 0;
    w.set = q.fold ? v * q.gs + g : (v * q.gs + g) * q.nch + cg;
    w.nsets = 1;
    w.set_stride = 0;
    return w;
}
// Forces passes: set (v, g) = the matrix-core chain over the strips g, g + gs, ... of segment v (gs = min(sps, full
// grid) groups per segment; no chunks).  Block g runs its group through the context's local segments one after the other
// -- all eight on one GPU, the 256 blocks sweeping one segment's strips side by side as they swept the whole matrix
// before r05; one on each of eight GPUs -- and writes a set at every segment's end.  Its strips as ONE sequence:
// number i is strip (i / tg) sps + g + gs (i % tg), tg = the group's strips per segment.
struct ForcesSlot {
    int g, tg, total, sps, gs;
    int flat;           // > 1 (ADJ on an interleaved copy): the slot's strips are POSITIONS g, g + gs, ... of the copy
    __device__ __forceinline__ int strip(int i) const {
        if (flat > 1) {                             // position p holds strip p / flat of local segment p % flat (strip_phys)
            const int p = g + gs * i;
            return (p % flat) * sps + p / flat;
        }
        const int v = i / tg;
        return v * sps + g + gs * (i - v * tg);
    }
};
// adj: the column-sum form on the one-copy path -- no sum over strips, so any assignment of strips to blocks gives the same
// bits; on a copy whose segments are interleaved (strip_phys) the blocks take the copy's positions in order, and at any
// moment read one contiguous window of it instead of every ilv-th strip of a window ilv times as wide
__device__ __forceinline__ ForcesSlot forces_slot(const StripArgs& q, int ps, bool adj = false) {
    ForcesSlot w;
    w.g = ps;
    w.sps = q.sps;
    w.gs = q.gs;
    w.flat = (adj && q.ilv > 1) ? q.ilv : 0;
    if (w.flat > 1) {
        w.total = (q.nstrips - ps + q.gs - 1) / q.gs;      // (ps < gs <= nstrips: at least one)
        w.tg = w.total + 1;                                 // never a segment's end: nothing is flushed in this form
        return w;
    }
    w.tg = (q.sps - ps + q.gs - 1) / q.gs;      // (ps < gs <= sps: at least one)
    w.total = w.tg * q.nlocal;
    return w;
}

// dynamic LDS: tile[mps * 16] | ul[mps * 8] | red[waves][8][16] | v[8][16] | scale[8]
//
// v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per instruction, 16 cycles; maps measured with
// one-hot operands, tools/mfma_f64_4x4_probe.hip): lane l = 16 kk + 4 blk + r holds A_blk[i = r][kk],
// B_blk[kk][j = r]; the result lane 16 i + 4 blk + j holds D_blk[i][j].  Unlike the 16x16x4 form nothing is
// padded: K <= 4 problems take one instruction per operand fetch, K <= 8 two, at 32 FLOP/clk/SIMD either way.
// DEPTH: 2 = two register sets in flight (K <= 4), 1 = one; 3 (r04, K > 4) = one register set AND the row-sum product of
// strip s deferred behind the first barrier of strip s + 1, where it runs beside P2 of that strip on the waves P2 leaves
// idle: ONE barrier per strip, P3 off the serial chain (the partial-sum, e | t and rescale buffers are doubled by strip
// parity).  Same operands, same order of every sum: the bits of a problem do not depend on which form served it.
// ADJ (r05): the column-sum half of pass 1 alone -- out_k[j] = sum_i Y'_ij u_ik + shift_k, the log-weights ADJOINT
// (k_strip_adj's product) on the ROW-sum order copy: what lets the log-weights method run with ONE strip copy of the
// matrix (ctx.hpp: one_copy).  No softmax, no row sums, no sets; instantiated with XY = true, DEPTH 2.
template <int K, bool NT, bool XY, int DEPTH = STRIP_DEPTH, int STORE = 0, bool ADJ = false>
__global__ __launch_bounds__(512, STRIP_WAVES_PER_SIMD) void k_strip(StripArgs q, ForcesRound fr) {
    constexpr int NK = (K + 3) / 4;                 // problem quads
    constexpr bool DEFER = DEPTH == 3;
    constexpr int SETS = DEPTH == 2 ? 2 : 1;        // register sets (strips in flight per wave)
    constexpr int NBUF = DEFER ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwaves = blockDim.x >> 6;
    // Every wave of the block runs the same straight-line code (a branch around the prefetch makes the compiler's
    // vmcnt bookkeeping drain BOTH register sets at every wait, see k_strip_adj): LDS regions are sized for
    // 64 rows per wave, and the second wave of a 64-row strip (blocks have at least two waves: P2 needs up to 128
    // threads) re-reads the first wave's rows against zero operands and stores nothing.
    const int lrows = nwaves * kWaveRows;
    double* tile = lds;
    double* ul = tile + (size_t)lrows * kStripCols;  // u[row][8]: forces | residuals, zero beyond K and mp
    double* red = ul + (size_t)lrows * 8;           // [parity][wave][problem 8][column 16]: the waves' partial column sums
    double* tv = red + NBUF * nwaves * 128;         // [parity] v[problem 8][column 16]: e | t of the strip
    double* scale = tv + NBUF * 128;
    double* cl = scale + NBUF * 16;                 // centre[row]
    const int rbase = wave * kWaveRows;
    const int rsrc = rbase < q.mps ? rbase : 0;     // rows the wave loads
    const int lq = lane >> 4, lr = lane & 15, lj = lane & 3;
    const ForcesSlot wk = forces_slot(q, blockIdx.x, ADJ);

    for (int i = t; i < lrows * 8; i += blockDim.x) {
        const int row = i >> 3, k = i & 7;
        ul[i] = (row < q.mp && k < K) ? q.u_c[(size_t)row * K + k] : 0.0;
    }
    for (int i = t; i < NBUF * 128; i += blockDim.x) tv[i] = 0.0;  // problems k >= K of a quad stay zero
    for (int i = t; i < lrows; i += blockDim.x) cl[i] = i < q.mp ? q.center[i] : 0.0;
    if (t < NBUF * 16) scale[t] = 1.0;

    // P3 accumulators: row block h (16 rows), problem quad kq: lane 16 i + 4 blk + j holds
    // row rbase + 16 h + 4 blk + i, problem 4 kq + j
    double acc[kWaveRows / 16][NK];
#pragma unroll
    for (int h = 0; h < kWaveRows / 16; ++h)
#pragma unroll
        for (int kq = 0; kq < NK; ++kq) acc[h][kq] = 0.0;

    // P2 state (threads t < 16 K: problem k = t / 16, column c = t % 16 -- a problem's 16 columns sit in one
    // 16-lane group, so the strip's maximum needs no LDS and no barrier)
    const bool p2 = t < kStripCols * K;
    const int pk = p2 ? t >> 4 : 0, pc = t & 15;
    double m_run = -DBL_MAX, zacc = 0.0, pxacc = 0.0;             // xy: running maximum, sum e, sum e x | bt: zacc = sum t
    double logs = 0.0, theta = 0.0, b0 = 0.0;
    // per-lane choice among the K kernel arguments by comparison (indexing the argument block with a lane value is
    // a vector load whose pending state forces vmcnt(0) -- a drain of the prefetch -- wherever the pointer is used)
    double* ak = fr.a[0];
    double* sck = fr.scal[0];
    double* pak = fr.part[0];
    double thk = fr.theta[0];
#pragma unroll
    for (int k = 1; k < K; ++k)
        if (pk == k) {
            ak = fr.a[k];
            sck = fr.scal[k];
            pak = fr.part[k];
            thk = fr.theta[k];
        }
    if (!XY && p2) {
        logs = sck[S_LOGS];
        b0 = sck[S_B0];
        theta = thk;
    }
    double shift = 0.0;                             // ADJ: sum_i u_ik (center_i - ybar_ik), k_strip_adj's constant
    if constexpr (ADJ) {                            // (accumulate: 0 = start at the shift, 1 = continue the panels before, 2 = start at 0)
        if (p2 && q.accumulate == 0) shift = sck[S_B0] - sck[S_UY];
    }

    // the wave's 8 KB of the next TWO strips travel in registers (two sets, used alternately)
    using Regs = StripRegs<kWaveRows / 8>;
    Regs preA;
    Regs preB;                                      // (SETS == 1: never touched)
    double a3old[4][kWaveRows / 16];                // DEFER: the row-sum operands of the strip before this one
    const size_t wave_off = (size_t)rsrc * kStripCols + (size_t)lane * 2;      // the wave's slice is contiguous in the copy
    int choff[kWaveRows / 8];                                                   // chunk -> chunk actually loaded (wave-uniform)
    strip_chunk_offsets(q.mps, rsrc, choff);
    auto fetch = [&](int strip, Regs& pre) {
#if !(STRIP_DIAG & 2)
        if constexpr (STORE == 0) {
            const double* src = q.Ys + (size_t)strip_phys(strip, q.sps, q.ilv) * q.mps * kStripCols + wave_off;
#pragma unroll
            for (int i = 0; i < kWaveRows / 8; ++i) pre.v[i] = ldg2<NT>(src + choff[i]);
        } else {                                    // reduced-storage experiment: centred, rows padded to 64
            constexpr int SB = reduced_slice_bytes<STORE>();
            const unsigned char* src = reinterpret_cast<const unsigned char*>(q.Ys) +
                                       ((size_t)strip * (q.mps / kWaveRows) + (size_t)(rsrc / kWaveRows)) * SB + (size_t)lane * 16;
#pragma unroll
            for (int u = 0; u < 4; ++u) pre.hi[u] = ldg16<NT, f4>(src + u * 1024);
            if constexpr (STORE == 1) {
#pragma unroll
                for (int u = 0; u < 2; ++u) pre.lo[u] = ldg16<NT, u4>(src + 4096 + u * 1024);
            }
        }
#endif
    };
#if STRIP_DIAG & 4
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_amdgcn_s_memtime();
#define STAMP(i) { const long long now_ = __builtin_amdgcn_s_memtime(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define STAMP(i)
#endif
    // ---- P3: acc[row][k] (+)= sum_c Y'[row][c] v[c][k] ----
    // A: lane (kk = lq, blk, i) = Y'[r0 + 4 blk + i = r0 + lr][c = 4 qq + lq]; B: lane (kk, blk, j) = v[4 qq + lq][4 kq + j]
    auto p3 = [&](double (&a3x)[4][kWaveRows / 16], const double* tvp, const double* scp) {
        if (XY) {
#pragma unroll
            for (int kq = 0; kq < NK; ++kq) {
                const double sc = scp[4 * kq + lj];
#pragma unroll
                for (int h = 0; h < kWaveRows / 16; ++h) acc[h][kq] *= sc;
            }
        }
#if !(STRIP_DIAG & 1)
        double bv[4][NK];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int kq = 0; kq < NK; ++kq) bv[qq][kq] = tvp[(4 * kq + lj) * 16 + 4 * qq + lq];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int h = 0; h < kWaveRows / 16; ++h)
#pragma unroll
                for (int kq = 0; kq < NK; ++kq)
                    acc[h][kq] = __builtin_amdgcn_mfma_f64_4x4x4f64(a3x[qq][h], bv[qq][kq], acc[h][kq], 0, 0, 0);
#else
        acc[0][0] += a3x[0][0] * tvp[lj * 16];
#endif
    };
    // A segment's sums leave as one set: the row sums and the set statistics; then everything starts from zero again
    // (pass 1: running maximum -DBL_MAX, so the next strip rescales by exp(-DBL_MAX - m) = 0 exactly as a block's first).
    // WHERE it is issued matters (r05, measured): the wave has two strips of loads in flight and waits for them with
    // vmcnt(N), N = the loads the compiler counts behind the one it needs.  Stores it does not count (they sit under a
    // block-uniform branch) issued BEHIND those loads make every such wait longer by as many of the YOUNGER loads -- the
    // strip after next -- as there are stores: 3-4 us per flush and block, 5-10 % of a pass with eight segments per block.
    // So the set of a finished segment leaves at the start of the NEXT segment's first strip, behind the wait for that
    // strip's own data and in front of its prefetch: then only the (fast) stores themselves stand in the way of a wait.
    auto flush = [&](int set) {
        // result lane 16 i + 4 blk + j: row rbase + 16 h + 4 blk + i, problem 4 kq + j.  Row block by row block (16 rows x K
        // sums = one run of <= 128 doubles of the set) through the wave's own slice of `red` -- free here: the column sums of
        // the strip before have been consumed -- so that the sums leave as contiguous stores: as 8-byte stores 8 K bytes apart
        // (r02-r04, once per launch) eight flushes per block cost 7 % of a pass at K = 4 (partial-line writes: 0.75 TB/s)
        int lz = lane;
        asm volatile("" : "+v"(lz));      // opaque: keeps the address arithmetic of these stores out of the strip loop's registers
        double* const stg = red + wave * 128;
        const int rl = 4 * ((lz >> 2) & 3) + (lz >> 4);
#pragma unroll
        for (int h = 0; h < kWaveRows / 16; ++h) {
#pragma unroll
            for (int kq = 0; kq < NK; ++kq) {
                const int k = 4 * kq + (lz & 3);
                // rows between the strip's last row block and mp exist only in the M-vectors: their sums are zero (the
                // wave computed a redirected row block's there)
                if (k < K) stg[rl * K + k] = rbase + 16 * h + rl < q.mps ? acc[h][kq] : 0.0;
                acc[h][kq] = 0.0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int nrun = min(16, q.mp - (rbase + 16 * h)) * K;          // (<= 0: rows beyond the operands; the idle second wave of a 64-row strip)
            double* const dst = q.partial + ((size_t)set * q.mp + rbase + 16 * h) * K;
            for (int i = lz; i < nrun; i += 64) __builtin_nontemporal_store(stg[i], dst + i);   // streamed: see k_strip_adj's outputs
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // set statistics per problem: sums over the problem's 16 columns (its 16-lane group)
        double z = zacc, px = pxacc;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            z += __shfl_xor(z, o, 64);
            px += __shfl_xor(px, o, 64);
        }
        if (p2 && pc == 0) {
            double* pa = pak;
            if (XY) {
                pa[(size_t)P_MAX * kPartStride + set] = m_run;
                pa[(size_t)P_SUM * kPartStride + set] = z;
                pa[(size_t)P_PP * kPartStride + set] = px;
            } else {
                pa[(size_t)P_KL * kPartStride + set] = z;        // this set's share of sum_j t_j
            }
        }
        m_run = -DBL_MAX;
        zacc = 0.0;
        pxacc = 0.0;
    };
    auto one_strip = [&](int si, Regs& pre, int par, bool first, int flush_set) {      // si: number of the strip in the slot's sequence
        const int s = wk.strip(si);
        double* const redp = red + (DEFER ? par * nwaves * 128 : 0);
        double* const tvp = tv + (DEFER ? par * 128 : 0);
        double* const scp = scale + (DEFER ? par * 16 : 0);
        // registers -> LDS image (row-major, swizzled) for the column sums; the same 16 rows x 2 columns per
        // 32 lanes as the row-sum operand fetch used to read: conflict-free.  The registers themselves ARE the
        // row-sum operands: kept in a3 until P3 (the prefetch below reuses `pre`).
        double a3[4][kWaveRows / 16];
        {
            const int sw3 = strip_swz(lr);          // row rbase + 16 h + lr: only bit 0 of its swizzle depends on h
            double* img = tile + (size_t)(rbase + lr) * kStripCols;
#pragma unroll
            for (int i = 0; i < kWaveRows / 8; ++i) {
                const int h = i >> 1, qp = i & 1;
                double vx, vy;
                if constexpr (STORE == 0) {
                    const double ch = cl[rsrc + 16 * h + lr];
#if STRIP_PRECENTERED
                    vx = pre.v[i].x, vy = pre.v[i].y;
#else
                    vx = pre.v[i].x - ch, vy = pre.v[i].y - ch;                   // the centring (r02: stored in the copy)
#endif
                } else {
                    vx = reduced_elem<STORE, 4>(pre.hi, pre.lo, i, 0);
                    vy = reduced_elem<STORE, 4>(pre.hi, pre.lo, i, 1);
                }
                a3[2 * qp][h] = vx;
                a3[2 * qp + 1][h] = vy;
                img[h * 256 + (((8 * qp + lq) ^ sw3) ^ (h & 1))] = vx;
                img[h * 256 + (((8 * qp + 4 + lq) ^ sw3) ^ (h & 1))] = vy;
            }
        }
        // wave-private slice of the tile: the wave's own program order is the synchronisation
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        STAMP(0)    // waited for the strip, copied it to LDS
        if constexpr (!ADJ) {
            if (flush_set >= 0) {                   // (block-uniform) the first strip of a segment: the set of the one before it
                if constexpr (DEFER) {              // ... whose last strip's deferred row sums run now
                    __syncthreads();                // its e | t are in place
                    p3(a3old, tv + (par ^ 1) * 128, scale + (par ^ 1) * 16);
                }
                flush(flush_set);
            }
        }
        // P2's operands first, THEN the prefetch: vmcnt retires in order, a load issued behind the
        // prefetch would wait for the whole strip after next
        const size_t col = (size_t)s * kStripCols + pc;
        double w0v = 0.0, xv = 0.0;
        if constexpr (!ADJ) {
            if (p2) {
                w0v = q.w0[col];
                if (!XY) xv = ak[col];
            }
        } else {
            if (p2 && q.accumulate == 1) xv = ak[col];            // the column sums of the row panels before this one
        }
        fetch(si + SETS < wk.total ? wk.strip(si + SETS) : s, pre);   // unconditional, see k_strip_adj
        // ---- P1: D1[c][k] = sum_{i in the wave's rows} Y'[i][c] u[i][k] ----
        // A: lane (kk = lq, blk, i) = Y'[r0 + lq][c = 4 blk + i = lr]; B: lane (kk = lq, blk, j) = u[r0 + lq][4 kq + j]
        {
            double d[4][NK];                        // four chains over the row groups: the result latency is 3 issues
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                for (int kq = 0; kq < NK; ++kq) d[ch][kq] = 0.0;
            // row rbase + 4 g + lq: its swizzle has period 8 in g (rbase is a multiple of 64): compile-time
            // offsets on four per-lane bases address all of P1
            const double* p1 = tile + (size_t)(rbase + lq) * kStripCols;
            const double* pu = ul + (size_t)(rbase + lq) * 8 + lj;
#if !(STRIP_DIAG & 1)
            // all operand fetches of a batch first (one LDS round trip instead of sixteen), then its matrix
            // instructions; K > 4 takes two batches of eight row groups: one batch of 16 x 3 operands does not fit
            // beside the two register sets in flight (the compiler spilled 2..22 values per strip)
            constexpr int NB = NK > 1 ? 2 : 1, GB = kWaveRows / 4 / NB;
#pragma unroll
            for (int hb = 0; hb < NB; ++hb) {
                double a1[GB], b1[GB][NK];
#pragma unroll
                for (int gg = 0; gg < GB; ++gg) {
                    const int g = hb * GB + gg;
                    a1[gg] = p1[g * 64 + (lr ^ (((((lq >> 1) + 2 * g) & 7) << 1) ^ ((g >> 2) & 1)))];
#pragma unroll
                    for (int kq = 0; kq < NK; ++kq) b1[gg][kq] = pu[g * 32 + 4 * kq];
                }
#pragma unroll
                for (int gg = 0; gg < GB; ++gg)
#pragma unroll
                    for (int kq = 0; kq < NK; ++kq)
                        d[gg & 3][kq] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1[gg], b1[gg][kq], d[gg & 3][kq], 0, 0, 0);
            }
#else
            d[0][0] = p1[lr] * pu[0];
#endif
            // result lane 16 i + 4 blk + j: column c = 4 blk + i, problem 4 kq + j
            const int c = 4 * ((lane >> 2) & 3) + lq;
            double* redw = redp;
#pragma unroll
            for (int kq = 0; kq < NK; ++kq)
                redw[wave * 128 + (4 * kq + lj) * 16 + c] = (d[0][kq] + d[1][kq]) + (d[2][kq] + d[3][kq]);
        }
        STAMP(1)    // issued the prefetch, P1
        __syncthreads();
        STAMP(2)    // first barrier
        // ---- P2 ----
        if constexpr (ADJ) {
            if (p2) {                               // the waves' partial column sums in wave order, the constant, out
                double colsum = 0.0;
                const int nown = (q.mps + kWaveRows - 1) / kWaveRows;
                for (int wv = 0; wv < nown; ++wv) colsum += redp[wv * 128 + pk * 16 + pc];
                __builtin_nontemporal_store(col < (size_t)q.n ? colsum + (q.accumulate == 1 ? xv : shift) : 0.0, ak + col);
            }
        } else if (t < kStripCols * K || (XY && wave < (kStripCols * K + 63) / 64)) {      // whole waves: the shuffles below
            double colsum = 0.0;
            if (p2) {
                const int nown = (q.mps + kWaveRows - 1) / kWaveRows;
                for (int wv = 0; wv < nown; ++wv) colsum += redp[wv * 128 + pk * 16 + pc];
            }
            if (XY) {
                const bool valid = p2 && col < (size_t)q.n;
                if (p2) __builtin_nontemporal_store(valid ? colsum : 0.0, ak + col);   // streamed, see k_strip_adj (plain: +3..5 %)
                double smax = valid ? colsum : -DBL_MAX;          // the strip's maximum: over the 16 lanes of the problem
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) smax = fmax(smax, __shfl_xor(smax, o, 64));
                const double m_new = fmax(m_run, smax);
                // the running maximum rarely moves after the first strips: exp(0) = 1 exactly, skip it wave-wide
                double sc = 1.0;
                if (__any(m_new != m_run)) sc = exp(m_run - m_new);   // 0 the first time
                const double e = valid ? w0v * exp(colsum - m_new) : 0.0;
                zacc = fma(zacc, sc, e);
                pxacc = fma(pxacc, sc, valid ? e * colsum : 0.0);
                m_run = m_new;
                if (p2) {
                    if (pc == 0) scp[pk] = sc;
                    tvp[pk * 16 + pc] = e;
                }
            } else if (p2) {
                const double lrat = xv - logs;                    // log(w / w0)
                const double wv = w0v * exp(lrat);
                double dd = 1.0;
                if (wv >= DBL_MIN && w0v >= DBL_MIN) dd += lrat;  // c_bioen_kernels_forces.c:320-328
                const double tval = (dd * theta + (colsum + b0)) * wv;
                tvp[pk * 16 + pc] = tval;
                zacc += tval;
            }
        }
        STAMP(3)    // P2
        if constexpr (DEFER) {
            // no second barrier: the row sums of the strip BEFORE this one run here, beside P2 of this strip (its e | t and
            // rescale factors sit in the other parity's buffers, complete since the barrier above); this strip's own
            // row-sum operands wait in a3old for the next turn
            if (!first) p3(a3old, tv + (par ^ 1) * 128, scale + (par ^ 1) * 16);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int h = 0; h < kWaveRows / 16; ++h) a3old[qq][h] = a3[qq][h];
            STAMP(4)
        } else {
            __syncthreads();
            STAMP(4)    // second barrier (ADJ: `red` may be rewritten)
            if constexpr (!ADJ) p3(a3, tvp, scp);
        }
        STAMP(5)    // P3
        // no barrier here: the next strip's copy goes to the wave's own slice; red is rewritten only after
        // every wave has finished P2 of this strip (second barrier above), v only after the next first barrier.
    };
    // The slot's strips, number 0 .. total - 1, straight through the local segments.  The two register sets of DEPTH 2
    // alternate strip by strip whatever the segments' lengths; the first strip of a segment carries the flush of the
    // segment before it (above), the last segment's set leaves behind the loop.
    int cnt = 0, vloc = 0;
    auto pending = [&](int i) { return (i > 0 && cnt == 0) ? (vloc - 1) * q.gs + wk.g : -1; };   // the set to flush at strip i
    auto count = [&]() {
        if (++cnt == wk.tg) {
            cnt = 0;
            ++vloc;
        }
    };
    fetch(wk.strip(0), preA);                                      // total >= 1: the slot has a first strip
    if constexpr (DEPTH == 2) fetch(wk.total > 1 ? wk.strip(1) : wk.strip(0), preB);
    __syncthreads();                                              // ul / tv / scale initialised
    if constexpr (DEPTH == 2) {
        // both halves and both prefetches unconditional inside the loop (see k_strip_adj); an odd last strip is peeled
        int i = 0;
        for (; i + 1 < wk.total; i += 2) {
            one_strip(i, preA, 0, false, pending(i));
            count();
            one_strip(i + 1, preB, 1, false, pending(i + 1));
            count();
        }
        if (i < wk.total) {
            one_strip(i, preA, 0, false, pending(i));
            count();
        }
    } else if constexpr (DEFER) {
        int par = 0;
        for (int i = 0; i < wk.total; ++i, par ^= 1) {
            const int fs = pending(i);
            one_strip(i, preA, par, i == 0 || fs >= 0, fs);       // (a segment's first strip: the deferred row sums of the strip
            count();                                              //  before it have run with the flush)
        }
        __syncthreads();                                          // the last strip's e | t are in place
        p3(a3old, tv + (par ^ 1) * 128, scale + (par ^ 1) * 16);
    } else {
        for (int i = 0, par = 0; i < wk.total; ++i, par ^= 1) {
            one_strip(i, preA, par, false, pending(i));
            count();
        }
    }
    if constexpr (!ADJ) flush((vloc - 1) * q.gs + wk.g);           // the last segment's set
#if STRIP_DIAG & 4
    if (q.stamps && lane == 0)
        for (int i = 0; i < 8; ++i) q.stamps[((size_t)blockIdx.x * 16 + wave) * 8 + i] = tacc[i];
#endif
}

// ---- the same two passes for 512 < M <= 1024 (r03; until then the r01 kernels on the row-major matrix: 4.8 TB/s at K = 1,
// spilling at K = 8).  Sixteen waves of 64 rows would leave 128 registers per wave and need a 128-KB image beside the
// 64-KB operand table; instead EIGHT waves own 128 rows each (256 registers, as k_strip): a wave keeps its 16 KB of
// the strip in registers as the row-sum operands (a3) and passes it through its 8-KB slice of the LDS image in two
// halves of 64 rows for the column sums (P1 over the first half, rewrite, P1 over the second half, one chain of
// accumulators).  One register set in flight (the prefetch is issued as soon as a3 holds the strip): 8 waves x 16 KB =
// the 128 KB per CU the other strip kernels keep in flight.  Everything else -- operand maps, swizzle, P2, P3, the
// straight-line rules about vmcnt -- is k_strip's; kept as a kernel of its own so that the tuned M <= 512 code does not
// move by an instruction.
#ifndef STRIP2_GB2
#define STRIP2_GB2 2
#endif
template <int K, bool NT, bool XY, int STORE = 0, bool ADJ = false>        // ADJ: as in k_strip
__global__ __launch_bounds__(512, 2) void k_strip2(StripArgs q, ForcesRound fr) {
    constexpr int RH = 2;                           // 64-row halves per wave
    constexpr int WR = 64 * RH;                     // rows per wave
    constexpr int NK = (K + 3) / 4;                 // problem quads
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwaves = blockDim.x >> 6;
    const int lrows = nwaves * WR;
    double* tile = lds;                              // [wave][64 rows][16]: one half of a wave's rows at a time
    double* ul = tile + (size_t)nwaves * 64 * kStripCols;  // u[row][8]: forces | residuals, zero beyond K and mp
    double* red = ul + (size_t)lrows * 8;           // [wave][problem 8][column 16]: the waves' partial column sums
    double* tv = red + nwaves * 128;                // v[problem 8][column 16]: e | t of the strip
    double* scale = tv + 128;
    double* cl = scale + 16;                        // centre[row]
    const int rbase = wave * WR;
    const int rsrc = rbase < q.mps ? rbase : 0;     // rows the wave loads
    const int lq = lane >> 4, lr = lane & 15, lj = lane & 3;
    const ForcesSlot wk = forces_slot(q, blockIdx.x, ADJ);

    for (int i = t; i < lrows * 8; i += blockDim.x) {
        const int row = i >> 3, k = i & 7;
        ul[i] = (row < q.mp && k < K) ? q.u_c[(size_t)row * K + k] : 0.0;
    }
    for (int i = t; i < 128; i += blockDim.x) tv[i] = 0.0;         // problems k >= K of a quad stay zero
    for (int i = t; i < lrows; i += blockDim.x) cl[i] = i < q.mp ? q.center[i] : 0.0;
    if (t < 8) scale[t] = 1.0;

    // P3 accumulators: row block h (16 rows), problem quad kq: lane 16 i + 4 blk + j holds
    // row rbase + 16 h + 4 blk + i, problem 4 kq + j
    double acc[WR / 16][NK];
#pragma unroll
    for (int h = 0; h < WR / 16; ++h)
#pragma unroll
        for (int kq = 0; kq < NK; ++kq) acc[h][kq] = 0.0;

    // P2 state (threads t < 16 K: problem k = t / 16, column c = t % 16 -- a problem's 16 columns sit in one
    // 16-lane group, so the strip's maximum needs no LDS and no barrier)
    const bool p2 = t < kStripCols * K;
    const int pk = p2 ? t >> 4 : 0, pc = t & 15;
    double m_run = -DBL_MAX, zacc = 0.0, pxacc = 0.0;             // xy: running maximum, sum e, sum e x | bt: zacc = sum t
    double logs = 0.0, theta = 0.0, b0 = 0.0;
    // per-lane choice among the K kernel arguments by comparison (indexing the argument block with a lane value is
    // a vector load whose pending state forces vmcnt(0) -- a drain of the prefetch -- wherever the pointer is used)
    double* ak = fr.a[0];
    double* sck = fr.scal[0];
    double* pak = fr.part[0];
    double thk = fr.theta[0];
#pragma unroll
    for (int k = 1; k < K; ++k)
        if (pk == k) {
            ak = fr.a[k];
            sck = fr.scal[k];
            pak = fr.part[k];
            thk = fr.theta[k];
        }
    if (!XY && p2) {
        logs = sck[S_LOGS];
        b0 = sck[S_B0];
        theta = thk;
    }
    double shift = 0.0;                             // ADJ: k_strip_adj's constant (k_strip)
    if constexpr (ADJ) {                            // (accumulate: 0 = start at the shift, 1 = continue the panels before, 2 = start at 0)
        if (p2 && q.accumulate == 0) shift = sck[S_B0] - sck[S_UY];
    }

    // the wave's 16 KB of the next strip travel in registers
    StripRegs<WR / 8> pre;
    const size_t wave_off = (size_t)rsrc * kStripCols + (size_t)lane * 2;      // the wave's slice is contiguous in the copy
    int choff[WR / 8];                                                          // chunk -> chunk actually loaded (wave-uniform)
    {
        const int nh = min(WR / 16, (q.mps - rsrc) / 16);                       // row blocks of this wave's slice
#pragma unroll
        for (int i = 0; i < WR / 8; ++i) choff[i] = __builtin_amdgcn_readfirstlane(((i >> 1) < nh ? i : (i & 1)) * 128);
    }
    auto fetch_part = [&](int strip, int lo, int hi) {
        if constexpr (STORE == 0) {
            const double* src = q.Ys + (size_t)strip_phys(strip, q.sps, q.ilv) * q.mps * kStripCols + wave_off;
#pragma unroll
            for (int i = 0; i < WR / 8; ++i)
                if (i >= lo && i < hi) pre.v[i] = ldg2<NT>(src + choff[i]);
        } else {      // reduced-storage experiment: centred, rows padded to 128 here; a wave's two 64-row slices are adjacent
            constexpr int SB = reduced_slice_bytes<STORE>();
            const unsigned char* src = reinterpret_cast<const unsigned char*>(q.Ys) +
                                       ((size_t)strip * (q.mps / 64) + (size_t)(rsrc / 64)) * SB + (size_t)lane * 16;
#pragma unroll
            for (int u = 0; u < WR / 16; ++u)
                if (2 * u >= lo && 2 * u < hi) pre.hi[u] = ldg16<NT, f4>(src + (u >> 2) * SB + (u & 3) * 1024);
            if constexpr (STORE == 1) {
#pragma unroll
                for (int u = 0; u < WR / 32; ++u)
                    if (4 * u >= lo && 4 * u < hi) pre.lo[u] = ldg16<NT, u4>(src + (u >> 1) * SB + 4096 + (u & 1) * 1024);
            }
        }
    };
    auto fetch = [&](int strip) { fetch_part(strip, 0, WR / 8); };
    // K > 4: the operand batches of the column-sum product do not fit beside a3 AND the whole prefetch (7-12 registers
    // spilled per strip); the second half of the prefetch is issued behind that product instead
    // (r06) The ADJ form has no row-sum accumulators and no second operand table: the WHOLE prefetch fits in front of the
    // product at every K (174 registers, no spill) -- K = 5 / 8 at N = 1e6 x M = 1024: 1.167 / 1.19 ms per launch against
    // 1.222 / 1.235 with the split (three alternations, one box: profiles/r06_adj_prefetch_ab.txt); -DSTRIP2_ADJ_SPLIT=1
    // brings the split back for an A/B.  The order of the loads changes no bit.
#ifndef STRIP2_ADJ_SPLIT
#define STRIP2_ADJ_SPLIT 0
#endif
    constexpr int SPLIT = (NK > 1 && (!ADJ || STRIP2_ADJ_SPLIT)) ? WR / 16 : WR / 8;
    auto flush = [&](int set) {                                    // a segment's sums leave as one set; then everything starts from zero (k_strip)
        // result lane 16 i + 4 blk + j: row rbase + 16 h + 4 blk + i, problem 4 kq + j.  Row block by row block (16 rows x K
        // sums = one run of <= 128 doubles of the set) through the wave's own slice of `red` -- free here: the column sums of
        // the strip before have been consumed -- so that the sums leave as contiguous stores: as 8-byte stores 8 K bytes apart
        // (r02-r04, once per launch) eight flushes per block cost 7 % of a pass at K = 4 (partial-line writes: 0.75 TB/s)
        int lz = lane;
        asm volatile("" : "+v"(lz));      // opaque: keeps the address arithmetic of these stores out of the strip loop's registers
        double* const stg = red + wave * 128;
        const int rl = 4 * ((lz >> 2) & 3) + (lz >> 4);
#pragma unroll
        for (int h = 0; h < WR / 16; ++h) {
#pragma unroll
            for (int kq = 0; kq < NK; ++kq) {
                const int k = 4 * kq + (lz & 3);
                // rows between the strip's last row block and mp exist only in the M-vectors: their sums are zero (the
                // wave computed a redirected row block's there)
                if (k < K) stg[rl * K + k] = rbase + 16 * h + rl < q.mps ? acc[h][kq] : 0.0;
                acc[h][kq] = 0.0;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int nrun = min(16, q.mp - (rbase + 16 * h)) * K;          // (<= 0: rows beyond the operands; the idle second wave of a 64-row strip)
            double* const dst = q.partial + ((size_t)set * q.mp + rbase + 16 * h) * K;
            for (int i = lz; i < nrun; i += 64) __builtin_nontemporal_store(stg[i], dst + i);   // streamed: see k_strip_adj's outputs
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        double z = zacc, px = pxacc;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            z += __shfl_xor(z, o, 64);
            px += __shfl_xor(px, o, 64);
        }
        if (p2 && pc == 0) {
            double* pa = pak;
            if (XY) {
                pa[(size_t)P_MAX * kPartStride + set] = m_run;
                pa[(size_t)P_SUM * kPartStride + set] = z;
                pa[(size_t)P_PP * kPartStride + set] = px;
            } else {
                pa[(size_t)P_KL * kPartStride + set] = z;        // this set's share of sum_j t_j
            }
        }
        m_run = -DBL_MAX;
        zacc = 0.0;
        pxacc = 0.0;
    };
    auto one_strip = [&](int si, int flush_set) {   // si: number of the strip in the slot's sequence (k_strip)
        const int s = wk.strip(si);
        // the strip's centred values: the row-sum operands of P3, and the source of the LDS image below
        double a3[4][WR / 16];
        {
#pragma unroll
            for (int i = 0; i < WR / 8; ++i) {
                const int h = i >> 1, qp = i & 1;
                if constexpr (STORE == 0) {
                    const double ch = cl[rsrc + 16 * h + lr];
                    a3[2 * qp][h] = pre.v[i].x - ch;                              // the centring
                    a3[2 * qp + 1][h] = pre.v[i].y - ch;
                } else {
                    a3[2 * qp][h] = reduced_elem<STORE, WR / 16>(pre.hi, pre.lo, i, 0);
                    a3[2 * qp + 1][h] = reduced_elem<STORE, WR / 16>(pre.hi, pre.lo, i, 1);
                }
            }
        }
        if constexpr (!ADJ)
            if (flush_set >= 0) flush(flush_set);  // (block-uniform) a segment's first strip: the set of the segment before it,
                                                   // behind the wait for this strip's data and in front of its prefetch (k_strip)
        // P2's operands first, THEN the prefetch: vmcnt retires in order, a load issued behind the
        // prefetch would wait for the whole strip after next
        const size_t col = (size_t)s * kStripCols + pc;
        double w0v = 0.0, xv = 0.0;
        if constexpr (!ADJ) {
            if (p2) {
                w0v = q.w0[col];
                if (!XY) xv = ak[col];
            }
        } else {
            if (p2 && q.accumulate == 1) xv = ak[col];            // the column sums of the row panels before this one
        }
        const int nxt = si + 1 < wk.total ? wk.strip(si + 1) : s;
        fetch_part(nxt, 0, SPLIT);                 // unconditional, see k_strip_adj; `pre` is free: a3 holds the strip
        // ---- P1: D1[c][k] = sum_{i in the wave's rows} Y'[i][c] u[i][k], half by half through the LDS image ----
        {
            // four chains over the row groups (the result latency is 3 issues) for EVERY K: a problem's sums must not
            // depend on the width of its batch
            double d[4][NK];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                for (int kq = 0; kq < NK; ++kq) d[ch][kq] = 0.0;
            const int sw3 = strip_swz(lr);          // row 16 h + lr of the half: only bit 0 of its swizzle depends on h
            double* img = tile + (size_t)(wave * 64 + lr) * kStripCols;
            const double* p1 = tile + (size_t)(wave * 64 + lq) * kStripCols;
#pragma unroll
            for (int half = 0; half < RH; ++half) {
                // registers -> image (row-major, swizzled: the same 16 rows x 2 columns per 32 lanes as a row-sum operand
                // fetch: conflict-free); the wave's own program order is the synchronisation of its private slice
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int hl = 0; hl < 4; ++hl)
#pragma unroll
                    for (int qp = 0; qp < 2; ++qp) {
                        img[hl * 256 + (((8 * qp + lq) ^ sw3) ^ (hl & 1))] = a3[2 * qp][4 * half + hl];
                        img[hl * 256 + (((8 * qp + 4 + lq) ^ sw3) ^ (hl & 1))] = a3[2 * qp + 1][4 * half + hl];
                    }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                // A: lane (kk = lq, blk, i) = Y'[r0 + lq][c = 4 blk + i = lr]; B: lane (kk = lq, blk, j) = u[r0 + lq][4 kq + j]
                const double* pu = ul + (size_t)(rbase + 64 * half + lq) * 8 + lj;
                // operand batches that fit beside a3 and the prefetch (K > 4 with the WHOLE prefetch in flight here, batches of
                // 4 | 2 | 1: 27-31 | 7-12 | 8-9 registers spilled; pass 1 / pass 2 at N = 1e6 x M = 1024, K = 8: 1.66 / 1.45 |
                // 1.50 / 1.39 | 1.51 / 1.41 ms; with the prefetch split around this product (SPLIT): none, 1.30 / 1.19 ms)
                constexpr int GB = NK > 1 ? STRIP2_GB2 : 8, NB = 16 / GB;
#pragma unroll
                for (int hb = 0; hb < NB; ++hb) {
                    double a1[GB], b1[GB][NK];
#pragma unroll
                    for (int gg = 0; gg < GB; ++gg) {
                        const int g = hb * GB + gg;
                        a1[gg] = p1[g * 64 + (lr ^ (((((lq >> 1) + 2 * g) & 7) << 1) ^ ((g >> 2) & 1)))];
#pragma unroll
                        for (int kq = 0; kq < NK; ++kq) b1[gg][kq] = pu[g * 32 + 4 * kq];
                    }
#pragma unroll
                    for (int gg = 0; gg < GB; ++gg)
#pragma unroll
                        for (int kq = 0; kq < NK; ++kq)
                            d[(hb * GB + gg) & 3][kq] =
                                __builtin_amdgcn_mfma_f64_4x4x4f64(a1[gg], b1[gg][kq], d[(hb * GB + gg) & 3][kq], 0, 0, 0);
                }
            }
            // result lane 16 i + 4 blk + j: column c = 4 blk + i, problem 4 kq + j
            const int c = 4 * ((lane >> 2) & 3) + lq;
#pragma unroll
            for (int kq = 0; kq < NK; ++kq)
                red[wave * 128 + (4 * kq + lj) * 16 + c] = (d[0][kq] + d[1][kq]) + (d[2][kq] + d[3][kq]);
        }
        if (SPLIT < WR / 8) fetch_part(nxt, SPLIT, WR / 8);
        __syncthreads();
        // ---- P2 ----
        if constexpr (ADJ) {
            if (p2) {                               // the waves' partial column sums in wave order, the constant, out
                double colsum = 0.0;
                const int nown = (q.mps + WR - 1) / WR;
                for (int wv = 0; wv < nown; ++wv) colsum += red[wv * 128 + pk * 16 + pc];
                __builtin_nontemporal_store(col < (size_t)q.n ? colsum + (q.accumulate == 1 ? xv : shift) : 0.0, ak + col);
            }
        } else if (t < kStripCols * K || (XY && wave < (kStripCols * K + 63) / 64)) {      // whole waves: the shuffles below
            double colsum = 0.0;
            if (p2) {
                const int nown = (q.mps + WR - 1) / WR;
                for (int wv = 0; wv < nown; ++wv) colsum += red[wv * 128 + pk * 16 + pc];
            }
            if (XY) {
                const bool valid = p2 && col < (size_t)q.n;
                if (p2) __builtin_nontemporal_store(valid ? colsum : 0.0, ak + col);   // streamed, see k_strip_adj (plain: +3..5 %)
                double smax = valid ? colsum : -DBL_MAX;          // the strip's maximum: over the 16 lanes of the problem
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) smax = fmax(smax, __shfl_xor(smax, o, 64));
                const double m_new = fmax(m_run, smax);
                // the running maximum rarely moves after the first strips: exp(0) = 1 exactly, skip it wave-wide
                double sc = 1.0;
                if (__any(m_new != m_run)) sc = exp(m_run - m_new);   // 0 the first time
                const double e = valid ? w0v * exp(colsum - m_new) : 0.0;
                zacc = fma(zacc, sc, e);
                pxacc = fma(pxacc, sc, valid ? e * colsum : 0.0);
                m_run = m_new;
                if (p2) {
                    if (pc == 0) scale[pk] = sc;
                    tv[pk * 16 + pc] = e;
                }
            } else if (p2) {
                const double lrat = xv - logs;                    // log(w / w0)
                const double wv = w0v * exp(lrat);
                double dd = 1.0;
                if (wv >= DBL_MIN && w0v >= DBL_MIN) dd += lrat;  // c_bioen_kernels_forces.c:320-328
                const double tval = (dd * theta + (colsum + b0)) * wv;
                tv[pk * 16 + pc] = tval;
                zacc += tval;
            }
        }
        __syncthreads();
        // ---- P3: acc[row][k] (+)= sum_c Y'[row][c] v[c][k] ----
        // A: lane (kk = lq, blk, i) = Y'[r0 + 4 blk + i = r0 + lr][c = 4 qq + lq]; B: lane (kk, blk, j) = v[4 qq + lq][4 kq + j]
        if constexpr (!ADJ) {
            if (XY) {
#pragma unroll
                for (int kq = 0; kq < NK; ++kq) {
                    const double sc = scale[4 * kq + lj];
#pragma unroll
                    for (int h = 0; h < WR / 16; ++h) acc[h][kq] *= sc;
                }
            }
            double bv[4][NK];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int kq = 0; kq < NK; ++kq) bv[qq][kq] = tv[(4 * kq + lj) * 16 + 4 * qq + lq];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int h = 0; h < WR / 16; ++h)
#pragma unroll
                    for (int kq = 0; kq < NK; ++kq)
                        acc[h][kq] = __builtin_amdgcn_mfma_f64_4x4x4f64(a3[qq][h], bv[qq][kq], acc[h][kq], 0, 0, 0);
        }
        // no barrier here: the next strip's copy goes to the wave's own slice; red is rewritten only after
        // every wave has finished P2 of this strip (second barrier above), v only after the next first barrier.
    };
    fetch(wk.strip(0));                                            // total >= 1: the slot has a first strip
    __syncthreads();                                              // ul / tv / scale / cl initialised
    // straight through the local segments; a segment's first strip carries the flush of the segment before it (k_strip)
    int cnt = 0, vloc = 0;
    for (int i = 0; i < wk.total; ++i) {
        one_strip(i, (i > 0 && cnt == 0) ? (vloc - 1) * q.gs + wk.g : -1);
        if (++cnt == wk.tg) {
            cnt = 0;
            ++vloc;
        }
    }
    if constexpr (!ADJ) flush((vloc - 1) * q.gs + wk.g);           // the last segment's set
}

// ---- log-weights forward pass on the strip copy: partial[set mp K + row K + k] = sum_{j in the set's strips} Y'[row][j] e_k[j]
// (A4, c_bioen_common.c:70-108; replaces k_fwd_partial for M <= 1024).  The copy is stored in the operand
// order of this product, so the matrix goes HBM -> registers -> matrix cores: no LDS image, no shuffles, and the
// K vectors e_k enter once per BLOCK and strip (16 K doubles through LDS, loaded one strip ahead) instead of once
// per wave and KiB as in the streaming kernel, whose K = 8 launch took 1.28 x its K = 1 time for that reason.
// Geometry: a wave owns 64 rows of one strip (128 registers: 16 waves per CU).  A block is `spb` strip slots of
// `wps` waves each -- the whole CU for every M (M = 1024: 1 x 16, 512: 2 x 8, 256: 4 x 4, <= 128: 4 x 2) -- and does
// `spb` strips per iteration behind ONE barrier: with one strip per block a 256-row problem ran 4 blocks of 4 waves
// per CU, four times the barriers and per-strip bookkeeping per byte, at 4.8 TB/s instead of 6.9.  Slot `sub` of
// block B is partial set B spb + sub and takes the strips set + it * (sets): the assignment, and therefore every
// bit of the result, is that of one block per set.  A 64-row strip keeps a second, idle wave per slot (the 16 K
// threads that stage e need up to two waves): it re-reads the first one's rows and stores nothing.
// One register set (one strip in flight per wave, 128 KB per CU).  r03 tried two (110-126 VGPRs, no spill, loop
// unrolled by two as in k_strip): 2418-2443 vs 2432-2438 us of matrix kernels per headline round, 84.7 vs 82 us at
// N = 1e5 x M = 256 -- no gain: the ~7 TB/s these passes reach is the memory system's rate for this stream, not a
// shortage of bytes in flight.
template <int K, bool NT, int STORE = 0>
__global__ __launch_bounds__(1024) void k_strip_fwd(StripArgs q, Vec8 v) {
    constexpr int NK = (K + 3) / 4;
    __shared__ double tv[2][4][8 * kStripCols];                   // [parity][slot][problem][column]
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int sub = wave / q.wps, rw = wave - sub * q.wps;
    const int rbase = rw * kWaveRows;
    const int rsrc = rbase < q.mps ? rbase : 0;
    const int lq = lane >> 4, lj = lane & 3;

    double acc[kWaveRows / 16][NK];
#pragma unroll
    for (int h = 0; h < kWaveRows / 16; ++h)
#pragma unroll
        for (int kq = 0; kq < NK; ++kq) acc[h][kq] = 0.0;
    for (int i = t; i < 2 * 4 * 8 * kStripCols; i += blockDim.x) (&tv[0][0][0])[i] = 0.0;   // problems k >= K of a quad stay zero

    const bool p2 = t < q.spb * kStripCols * K;                   // slot t / 16 K, problem (t % 16 K) / 16, column t % 16
    const int psub = p2 ? t / (kStripCols * K) : 0;
    const int pk = p2 ? (t - psub * kStripCols * K) >> 4 : 0, pc = t & 15;
    const double* vk = v.p[0];                                    // chosen by comparison, not by a lane-indexed (vector) load
#pragma unroll
    for (int k = 1; k < K; ++k)
        if (pk == k) vk = v.p[k];
    d2 pre[kWaveRows / 8];
    f4 rhi[4];                                                     // reduced formats (STORE != 0): the wave's slice as it is stored
    u4 rlo[2];
    double cen[kWaveRows / 16];                                    // centre of the lane's row in each of its four row blocks
#pragma unroll
    for (int h = 0; h < kWaveRows / 16; ++h) {
        const int row = rsrc + 16 * h + (lane & 15);
        cen[h] = (STORE == 0 && row < q.mp) ? q.center[row] : 0.0;     // the reduced copies hold the centred operand
    }
    const size_t wave_off = (size_t)rsrc * kStripCols + (size_t)lane * 2;
    int choff[kWaveRows / 8];
    strip_chunk_offsets(q.mps, rsrc, choff);
    auto fetch = [&](int strip) {
        if constexpr (STORE == 0) {
            const double* src = q.Ys + (size_t)strip_phys(strip, q.sps, q.ilv) * q.mps * kStripCols + wave_off;
#pragma unroll
            for (int i = 0; i < kWaveRows / 8; ++i) pre[i] = ldg2<NT>(src + choff[i]);
        } else {
            constexpr int SB = reduced_slice_bytes<STORE>();
            const unsigned char* src = reinterpret_cast<const unsigned char*>(q.Ys) +
                                       ((size_t)strip * (q.mps / kWaveRows) + (size_t)(rsrc / kWaveRows)) * SB + (size_t)lane * 16;
#pragma unroll
            for (int u = 0; u < 4; ++u) rhi[u] = ldg16<NT, f4>(src + u * 1024);
            if constexpr (STORE == 1) {
#pragma unroll
                for (int u = 0; u < 2; ++u) rlo[u] = ldg16<NT, u4>(src + 4096 + u * 1024);
            }
        }
    };
    // canonical sets (kernels.hpp: StripSets): the wave's slot and the slot this thread stages e for
    const SlotWork mw = strip_slot(q, blockIdx.x * q.spb + sub);
    const SlotWork pw = strip_slot(q, blockIdx.x * q.spb + psub);
    int trips = 0;                                                 // the block's iterations: its longest slot's (block-uniform)
    for (int i = 0; i < q.spb; ++i) trips = max(trips, strip_slot(q, blockIdx.x * q.spb + i).count);
    const int safe = mw.count > 0 ? mw.first : 0;                  // a strip the wave may touch when it has none of its own left
    double ecur = (p2 && pw.count > 0) ? vk[(size_t)pw.first * kStripCols + pc] : 0.0;
    fetch(safe);
    __syncthreads();
    // fold: the slot runs a whole group and adds up its chunks (tc strips each) in turn, from +0.0 -- what the consumer
    // does with the sets of a launch that runs one chunk per slot (k_fwd_rows_local_t): the same bits either way
    double tot[kWaveRows / 16][NK];
#pragma unroll
    for (int h = 0; h < kWaveRows / 16; ++h)
#pragma unroll
        for (int kq = 0; kq < NK; ++kq) tot[h][kq] = 0.0;
    int tcnt = 0;
    for (int it = 0, par = 0; it < trips; ++it, par ^= 1) {
        if (p2) tv[par][psub][pk * 16 + pc] = ecur;               // loaded during the previous strip
        ecur = (p2 && it + 1 < pw.count) ? vk[(size_t)(pw.first + q.gs * (it + 1)) * kStripCols + pc] : 0.0;   // (without these loads: -2 %; nontemporal: +1 %)
        __syncthreads();                                           // this strip's e is in place; the buffer of parity
                                                                   // `par` is rewritten two strips on, behind another barrier
        double bv[4][NK];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int kq = 0; kq < NK; ++kq) bv[qq][kq] = tv[par][sub][(4 * kq + lj) * 16 + 4 * qq + lq];
        // consecutive instructions go to different accumulators (a result is ready three issue slots later)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int h = 0; h < kWaveRows / 16; ++h) {
                double a;
                if constexpr (STORE == 0) {
                    const d2 y = pre[2 * h + (qq >> 1)];
#if STRIP_PRECENTERED
                    a = (qq & 1) ? y.y : y.x;
#else
                    a = ((qq & 1) ? y.y : y.x) - cen[h];               // the centring (r02: stored in the copy)
#endif
                } else {
                    a = reduced_elem<STORE, 4>(rhi, rlo, 2 * h + (qq >> 1), qq & 1);
                }
#pragma unroll
                for (int kq = 0; kq < NK; ++kq)
                    acc[h][kq] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bv[qq][kq], acc[h][kq], 0, 0, 0);
            }
        // unconditional (see k_strip_adj); the operands are consumed at issue.  Past its last strip a wave re-reads a
        // strip it may touch (its last one, or strip 0) against e = 0
#if FWD_DIAG & 1
        fetch(min((int)(blockIdx.x * q.spb + sub + (it + 1) * gridDim.x * q.spb), q.nstrips - 1));
#else
        fetch(it + 1 < mw.count ? mw.first + q.gs * (it + 1) : (mw.count > 0 ? mw.first + q.gs * (mw.count - 1) : 0));
#endif
        if (!(FWD_DIAG & 2) && q.fold && ++tcnt == q.tc) {                            // a chunk ends (block-uniform; registers only)
            tcnt = 0;
#pragma unroll
            for (int h = 0; h < kWaveRows / 16; ++h)
#pragma unroll
                for (int kq = 0; kq < NK; ++kq) {
                    tot[h][kq] += acc[h][kq];
                    acc[h][kq] = 0.0;
                }
        }
    }
    if (q.fold) {                                                  // the group's last, shorter chunk (or + 0.0)
#pragma unroll
        for (int h = 0; h < kWaveRows / 16; ++h)
#pragma unroll
            for (int kq = 0; kq < NK; ++kq) acc[h][kq] = tot[h][kq] + acc[h][kq];
    }
    {
        // result lane 16 i + 4 blk + j: row rbase + 16 h + 4 blk + i, problem 4 kq + j
        const int rr = rbase + 4 * ((lane >> 2) & 3) + lq;
#pragma unroll
        for (int h = 0; h < kWaveRows / 16; ++h)
#pragma unroll
            for (int kq = 0; kq < NK; ++kq) {
                const int row = rr + 16 * h, k = 4 * kq + lj;
                if (mw.live && row < q.mp && k < K)     // transposed: a set's sums are one run; rows beyond the strip: zero
                    q.partial[(size_t)mw.set * q.pstride * K + (size_t)row * K + k] = row < q.mps ? acc[h][kq] : 0.0;
            }
    }
}

// ---- log-weights adjoint pass on the strip copy (column-sum operand order):
//   out_k[j] = sum_i u_ik (Y_ij - ybar_ik) = sum_i Y'_ij u_ik + shift_k,   shift_k = sum_i u_ik (center_i - ybar_ik)
// (A6, c_bioen_kernels_logw.c:185-205; replaces k_adj for M <= 1024).  HBM -> registers -> matrix cores and the block
// geometry as in the forward pass; u = r (compact [row K + k]) sits in an LDS table in B-operand reach, the partial
// column sums of a slot's waves meet in LDS (two buffers by strip parity: one barrier per iteration), 16 K threads per
// slot add the shift and store.
// One register set, as in the forward pass.  r06 tried two here (the strip after next requested before the products of the
// next one start, loop unrolled by two, same bits; profiles/r06_adj_depth_ab.txt): K <= 4: 1.18-1.21 ms per launch at
// N = 1e6 x M = 1024 against 1.17-1.21 (nothing), K > 4: 1.71 ms against 1.21-1.26 (the second set does not fit beside
// 2 x 16 operand registers of u).  What a launch takes moves by 4-5 % with the process and the box (the first 0.2 s of a
// process, where the copy landed in HBM: tools/pass_probe.py shows it for both passes and for the plain read probe alike),
// not with bytes in flight.  Nor does it help at K > 4 to move the 16 centred operands out of the load registers first and
// request the next strip before the 32 products (what the one-copy form k_strip2<ADJ> gains 4 % from, below): 1.231-1.258 ms
// either way at N = 1e6 x M = 1024, K = 8 (three alternations, one box).
template <int K, bool NT, int STORE = 0>
__global__ __launch_bounds__(1024) void k_strip_adj(StripArgs q, MVec8 out, MVec8 scal) {
    constexpr int NK = (K + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nwaves = blockDim.x >> 6;
    const int lrows = q.wps * kWaveRows;                          // the table holds 64 rows per wave of a slot
    double* ul = lds;                                             // u[row][8], zero beyond K and mp
    double* cl = ul + (size_t)lrows * 8;                          // centre[row] (16 per lane in registers would spill at K = 8)
    double* red = cl + lrows;                                     // [parity][wave][problem 8][column 16]
    const int sub = wave / q.wps, rw = wave - sub * q.wps;
    const int rbase = rw * kWaveRows;
    const int rsrc = rbase < q.mps ? rbase : 0;                   // the idle second wave of a 64-row strip re-reads the first one's rows
    const int lq = lane >> 4, lj = lane & 3;
    const int stride = gridDim.x * q.spb;
    for (int i = t; i < lrows * 8; i += blockDim.x) {
        const int row = i >> 3, k = i & 7;
        ul[i] = (row < q.mp && k < K) ? q.u_c[(size_t)row * K + k] : 0.0;
    }
    for (int i = t; i < lrows; i += blockDim.x) cl[i] = i < q.mp ? q.center[i] : 0.0;
    const bool p2 = t < q.spb * kStripCols * K;                   // slot t / 16 K, problem (t % 16 K) / 16, column t % 16
    const int psub = p2 ? t / (kStripCols * K) : 0;
    const int pk = p2 ? (t - psub * kStripCols * K) >> 4 : 0, pc = t & 15;
    double shift = 0.0;
    double* outk = out.p[0];                                      // chosen by comparison, not by a lane-indexed (vector) load
    const double* sck = scal.p[0];
#pragma unroll
    for (int k = 1; k < K; ++k)
        if (pk == k) {
            outk = out.p[k];
            sck = scal.p[k];
        }
    if (p2 && q.accumulate == 0) shift = sck[S_B0] - sck[S_UY];
    d2 pre[kWaveRows / 8];
    f4 rhi[4];                                                     // reduced formats (STORE != 0)
    u4 rlo[2];
    const size_t wave_off = (size_t)rsrc * kStripCols + (size_t)lane * 2;
    int choff[kWaveRows / 8];
    strip_chunk_offsets(q.mps, rsrc, choff);
    auto fetch = [&](int strip) {
        if constexpr (STORE == 0) {
            const double* src = q.Ys + (size_t)strip * q.mps * kStripCols + wave_off;
#pragma unroll
            for (int i = 0; i < kWaveRows / 8; ++i) pre[i] = ldg2<NT>(src + choff[i]);
        } else {
            constexpr int SB = reduced_slice_bytes<STORE>();
            const unsigned char* src = reinterpret_cast<const unsigned char*>(q.Ys) +
                                       ((size_t)strip * (q.mps / kWaveRows) + (size_t)(rsrc / kWaveRows)) * SB + (size_t)lane * 16;
#pragma unroll
            for (int u = 0; u < 4; ++u) rhi[u] = ldg16<NT, f4>(src + u * 1024);
            if constexpr (STORE == 1) {
#pragma unroll
                for (int u = 0; u < 2; ++u) rlo[u] = ldg16<NT, u4>(src + 4096 + u * 1024);
            }
        }
    };
    int base = blockIdx.x * q.spb;                                 // strip of slot 0: < nstrips for every block
    int sw = base + sub, sp = base + psub;                         // this wave's strip / the strip this thread stores for
    fetch(sw < q.nstrips ? sw : base);
    __syncthreads();                                              // ul in place
    const double* pu = ul + (size_t)(rbase + lq) * 8 + lj;
    const double* pc_ = cl + (rsrc + lq);                         // the centre of row group g: pc_[4 g]
    const int nown = (q.mps + kWaveRows - 1) / kWaveRows;
    for (int par = 0; base < q.nstrips; base += stride, sw += stride, sp += stride, par ^= 1) {
        double* redw = red + (size_t)par * nwaves * 128;
        {
            double d[4][NK];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                for (int kq = 0; kq < NK; ++kq) d[ch][kq] = 0.0;
            double b1[kWaveRows / 4][NK];
#pragma unroll
            for (int g = 0; g < kWaveRows / 4; ++g)
#pragma unroll
                for (int kq = 0; kq < NK; ++kq) b1[g][kq] = pu[g * 32 + 4 * kq];
#pragma unroll
            for (int g = 0; g < kWaveRows / 4; ++g) {
                double a;
                if constexpr (STORE == 0) {
#if STRIP_PRECENTERED
                    a = (g & 1) ? pre[g >> 1].y : pre[g >> 1].x;
#else
                    a = ((g & 1) ? pre[g >> 1].y : pre[g >> 1].x) - pc_[4 * g];   // the centring (r02: stored in the copy)
#endif
                } else {
                    a = reduced_elem<STORE, 4>(rhi, rlo, g >> 1, g & 1);
                }
#pragma unroll
                for (int kq = 0; kq < NK; ++kq)
                    d[g & 3][kq] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b1[g][kq], d[g & 3][kq], 0, 0, 0);
            }
            // unconditional (past its last strip a wave re-reads one it may touch): a conditional prefetch makes the
            // compiler's vmcnt bookkeeping assume the loads may not exist, and every wait then drains them all
            const int nxt = sw + stride;
            fetch(nxt < q.nstrips ? nxt : (sw < q.nstrips ? sw : base));   // the operands are consumed at issue
            const int c = 4 * ((lane >> 2) & 3) + lq;             // result lane 16 i + 4 blk + j: column 4 blk + i, problem 4 kq + j
#pragma unroll
            for (int kq = 0; kq < NK; ++kq)
                redw[wave * 128 + (4 * kq + lj) * 16 + c] = (d[0][kq] + d[1][kq]) + (d[2][kq] + d[3][kq]);
        }
        __syncthreads();            // the buffer of this parity is rewritten two strips on, behind the next barrier
        if (p2 && sp < q.nstrips) {
            double colsum = 0.0;
            for (int wv = 0; wv < nown; ++wv) colsum += redw[(psub * q.wps + wv) * 128 + pk * 16 + pc];
#if ADJ_DIAG & 1
            if (colsum == 1.2345e300) outk[(size_t)sp * kStripCols + pc] = colsum + shift;
#else
            // streamed past the L2 (no write-allocate): a plain store here cost 3 % of the pass -- the 8 K bytes per
            // column are 0.8 % of the traffic, without any store the pass runs at the forward pass's time.  (r03, tried
            // and dropped: contiguous runs of strips per block with the outputs staged in LDS and written as KiB runs --
            // which strip a block takes changes no bit here.  One run per block: 1 % SLOWER at N = 1e6 x M = 1024, the
            // blocks' reads no longer sweep the HBM channels together; runs of 8 strips dealt round-robin: +0.2..0.7 %,
            // within the noise, and the uneven last run costs as much.)
            // (a row panel behind the first one of a matrix taller than 1024 rows continues its predecessors' sums)
            const double start = q.accumulate == 1 ? outk[(size_t)sp * kStripCols + pc] : shift;     // (2: shift stayed 0)
            __builtin_nontemporal_store(colsum + start, outk + (size_t)sp * kStripCols + pc);
#endif
        }
    }
}

// ---- geometry ------------------------------------------------------------------------------------
// Rows of a strip: M padded to the 16-row block of a matrix-core operand (r02 padded to a wave's 64 rows: M = 205
// streamed 256 rows, 20 % of the traffic for nothing; M = 28 streamed 64).  A wave still owns 64 rows of a strip; the
// last wave of a strip owns the 1..4 row blocks that exist, and the chunks of the others are redirected to its first
// row block (cache hits, multiplied with zero operands or never stored), so that every wave runs the same
// straight-line code and every sum is formed from the same terms in the same order as before.
static int strip_rows(const bioen_hip_ctx* c) { return (int)round_up((size_t)c->m, 16); }
// Matrices taller than 1024 rows (r03): the matrix passes run over row PANELS of <= 1024 rows -- the same two
// kernels once per panel, the forward pass writing its panel's rows of the partial sums, the adjoint pass continuing
// the column sums of the panels before it.  Until r03 that range ran the r01 streaming kernels (K = 8 forward pass 1.28 x
// its K = 1 time).  The panels are cut from the row-major matrix, which is freed once they exist (read-back and the
// fallback kernels gather it back from them: gather_block / ensure_rowmajor).
constexpr int kPanelRows = 1024;
static int panel_count(const bioen_hip_ctx* c) { return c->mp <= kPanelRows ? 1 : (c->m + kPanelRows - 1) / kPanelRows; }
static int panel_m(const bioen_hip_ctx* c, int p) { return c->mp <= kPanelRows ? c->m : std::min(kPanelRows, c->m - p * kPanelRows); }   // valid rows
static int panel_mp(const bioen_hip_ctx* c, int p) { return c->mp <= kPanelRows ? c->mp : std::min(kPanelRows, c->mp - p * kPanelRows); } // operand rows
static int panel_mps(const bioen_hip_ctx* c, int p) { return (int)round_up((size_t)panel_m(c, p), 16); }                                  // strip rows
static bool paneled(const bioen_hip_ctx* c) { return c->mp > kPanelRows; }
bool strip_panels(const bioen_hip_ctx* c) { return paneled(c); }

static int strip_waves(const bioen_hip_ctx* c) { return (strip_rows(c) + kWaveRows - 1) / kWaveRows; }
// forces kernels: k_strip (a wave owns 64 rows) for M <= 512, k_strip2 (128 rows per wave) for 512 < M <= 1024
static bool strip_tall(const bioen_hip_ctx* c) { return c->mp > 512; }
static int strip_threads(const bioen_hip_ctx* c) {
    return 64 * std::max(2, strip_tall(c) ? (strip_rows(c) + 2 * kWaveRows - 1) / (2 * kWaveRows) : strip_waves(c));
}
static size_t strip_lds_bytes(const bioen_hip_ctx* c) {
    const size_t waves = strip_threads(c) / 64;
    if (strip_tall(c))      // image: 64 rows per wave (one half at a time); operand table and centres: 128 rows per wave
        return (waves * kWaveRows * kStripCols + waves * 2 * kWaveRows * (8 + 1) + waves * 128 + 128 + 16) * sizeof(double);
    return (waves * kWaveRows * (kStripCols + 8 + 1) + waves * 128 + 128 + 16) * sizeof(double);   // 64 rows per wave
}

static int env_flag(const char* name, int dflt);
// segments interleaved in the row-sum order FP64 copies of this context (strip_phys); BIOEN_HIP_STRIP_INTERLEAVE=0: strip
// order as until r05 (A/B; read once per context at the first copy)
static int strip_ilv(const bioen_hip_ctx* c) { return std::max(1, c->strip_ilv); }      // the layout the copies ARE in
// the layout a method wants: the log-weights passes (k_strip_fwd's folded groups, one slot per (segment, group)) the
// interleaved one; the forces passes (a block runs its group through the segments one after the other: already one window)
// strip order -- interleaved they read every ilv-th strip of an ilv times wider window, 2-4 % slower at K >= 6
static int strip_ilv_wanted(const bioen_hip_ctx* c, bool forces) {
    if (forces) return 1;
    return env_flag("BIOEN_HIP_STRIP_INTERLEAVE", 1) != 0 ? std::max(1, c->vr) : 1;
}
static int strip_sps(const bioen_hip_ctx* c) { return c->segcols / kStripCols; }

static int env_flag(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}

// canonical sets of a row-sum pass whose full grid is `gs_nominal` slots per segment (kernels.hpp: StripSets)
static StripSets make_sets(const bioen_hip_ctx* c, int gs_nominal, bool may_fold) {
    StripSets ss{};
    ss.sps = c->segcols / kStripCols;
    ss.gs = std::max(1, std::min(ss.sps, gs_nominal));
    const int tmax = (ss.sps + ss.gs - 1) / ss.gs;
    ss.tc = (tmax + 7) / 8;
    ss.nch = (tmax + ss.tc - 1) / ss.tc;
    // a context that holds all eight segments runs whole groups per slot (8 x gs slots: the full grid) and adds up the
    // chunks in registers; BIOEN_HIP_STRIP_FOLD=0: one chunk per slot there too (tests: the same bits)
    ss.fold = (may_fold && c->vr >= 8 && env_flag("BIOEN_HIP_STRIP_FOLD", 1) != 0) ? 1 : 0;
    ss.slots = ss.gs * (ss.fold ? 1 : ss.nch);
    ss.sets = ss.slots;
    return ss;
}

static int forces_per_cu(const bioen_hip_ctx* c) {
    // blocks per CU: LDS (160 KiB) and the waves per SIMD the kernel's register budget admits
    const int by_lds = (int)((size_t)160 * 1024 / strip_lds_bytes(c));
    const int by_waves = 4 * STRIP_WAVES_PER_SIMD / (strip_threads(c) / 64);
    return std::max(1, std::min(std::min(by_lds, by_waves), 4));
}

StripSets forces_sets(const bioen_hip_ctx* c) {        // gs = 0: the strip passes do not apply to this context
    static int tall_off = -1;                          // BIOEN_HIP_STRIP_TALL=0: the streaming kernels for 512 < M <= 1024 (A/B)
    if (tall_off < 0) tall_off = env_flag("BIOEN_HIP_STRIP_TALL", 1) == 0 ? 1 : 0;
    if (c->mp > 1024 || (c->mp > 512 && tall_off) || c->strips_unavailable) return StripSets{};
    // one set per (segment, group): gs = the full grid of a GPU holding ONE segment (or every strip of the segment); a
    // block runs its group through the local segments (kernels_strip.hip: ForcesSlot)
    StripSets ss{};
    ss.sps = c->segcols / kStripCols;
    ss.gs = std::max(1, std::min(ss.sps, 256 * forces_per_cu(c)));
    ss.tc = (ss.sps + ss.gs - 1) / ss.gs;
    ss.nch = 1;
    ss.fold = 0;
    ss.slots = ss.gs;
    ss.sets = ss.gs;
    return ss;
}

int forces_fused_blocks(const bioen_hip_ctx* c) {      // sets per SEGMENT of the forces strip passes; 0: not applicable
    return forces_sets(c).sets;
}

// Every allocation of a strip copy goes through here.  Tests: BIOEN_HIP_TEST_FAIL_STRIP_ALLOC=k makes the k-th one of a
// context fail as an exhausted device would (the fallback to the streaming kernels is otherwise never exercised).
static hipError_t strip_malloc(bioen_hip_ctx* c, double** p, size_t bytes) {
    ++c->strip_allocs;
    if (const char* e = std::getenv("BIOEN_HIP_TEST_FAIL_STRIP_ALLOC"))
        if (std::atoi(e) == c->strip_allocs) return hipErrorOutOfMemory;
    return hipMalloc(reinterpret_cast<void**>(p), bytes);
}

// A failed allocation leaves the context WITHOUT strip copies (strips_unavailable): fwd_strip_blocks /
// forces_fused_blocks then answer 0 and every caller takes the streaming kernels on the row-major matrix, which
// need no extra memory.  The copy pointers are published only after the build kernel is enqueued.
static int strip_copy_failed(bioen_hip_ctx* c, double* ys, hipError_t e, const char* what) {
    if (ys) (void)hipFree(ys);
    (void)hipGetLastError();                       // an out-of-memory error must not surface at the next launch check
    c->strips_unavailable = 1;
    return hip_fail(e, what, __FILE__, __LINE__);
}

// The row-major matrix is the form data ARRIVE in (upload, device-side assembly, generator) and the operand of the
// streaming kernels (M > 1024).  Once the row-sum strip copy exists it is
// redundant for every other path -- the copy holds the same numbers -- and is freed (bioen_hip_ctx_read_ytilde,
// bioen_hip_chi_squared and the column-sum copy are served by the strip copy); a later call that needs it gets it back
// from the strip copy (ensure_rowmajor) and then keeps it.  Footprint of the matrix, M <= 1024: log-weights 2 x (both
// strip copies), forces method 1 x.
int ensure_rowmajor(bioen_hip_ctx* c) {
    if (c->Y) return 0;
    if (!c->Ys && !(paneled(c) && c->Yp[0])) return BIOEN_HIP_ESTATE;
    int rc = 0;
    double* y = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&y), (size_t)c->mp * c->ld * sizeof(double));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        hip_fail(e, "hipMalloc (row-major matrix back from the strip copy)", __FILE__, __LINE__);
        return BIOEN_HIP_ENOMEM;
    }
    e = hipMemsetAsync(y, 0, (size_t)c->mp * c->ld * sizeof(double), c->stream);
    if (e != hipSuccess) rc = hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__);
    for (int p = 0; p < panel_count(c) && !rc; ++p) {
        const int mps = paneled(c) ? panel_mps(c, p) : strip_rows(c);
        const double* src = paneled(c) ? c->Yp[p] : c->Ys;
        hipLaunchKernelGGL(k_gather_strips, dim3(4096), dim3(256), 0, c->stream, src, mps, 0, std::min(mps, panel_mp(c, p)),
                           (size_t)0, (int)c->ld, y + (size_t)p * kPanelRows * c->ld, c->ld, strip_sps(c), strip_ilv(c));
        e = hipGetLastError();
        if (e != hipSuccess) rc = hip_fail(e, "k_gather_strips", __FILE__, __LINE__);
    }
    if (rc) {
        (void)hipFree(y);
        return rc;
    }
    c->Y = y;
    c->rowmajor_rebuilt = 1;
    return 0;
}

// block of the matrix -> device buffer out[rows][cols], whichever form is resident
int gather_block(bioen_hip_ctx* c, int row0, int rows, size_t col0, int cols, double* out) {
    if (c->Y) {
        hipError_t e = hipMemcpy2DAsync(out, (size_t)cols * sizeof(double), c->Y + (size_t)row0 * c->ld + col0,
                                        c->ld * sizeof(double), (size_t)cols * sizeof(double), (size_t)rows,
                                        hipMemcpyDeviceToDevice, c->stream);
        return e == hipSuccess ? 0 : hip_fail(e, "hipMemcpy2DAsync", __FILE__, __LINE__);
    }
    if (paneled(c)) {                                   // the rows of [row0, row0 + rows) panel by panel
        if (!c->Yp[0]) return BIOEN_HIP_ESTATE;
        for (int p = row0 / kPanelRows; p < panel_count(c) && p * kPanelRows < row0 + rows; ++p) {
            const int lo = std::max(row0, p * kPanelRows), hi = std::min(row0 + rows, (p + 1) * kPanelRows);
            const size_t total = (size_t)(hi - lo) * cols;
            hipLaunchKernelGGL(k_gather_strips, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0,
                               c->stream, c->Yp[p], panel_mps(c, p), lo - p * kPanelRows, hi - lo, col0, cols,
                               out + (size_t)(lo - row0) * cols, (size_t)cols, strip_sps(c), strip_ilv(c));
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) return hip_fail(e, "k_gather_strips", __FILE__, __LINE__);
        }
        return 0;
    }
    if (!c->Ys) return BIOEN_HIP_ESTATE;
    const size_t total = (size_t)rows * cols;
    hipLaunchKernelGGL(k_gather_strips, dim3((unsigned)std::min<size_t>(4096, (total + 255) / 256)), dim3(256), 0, c->stream,
                       c->Ys, strip_rows(c), row0, rows, col0, cols, out, (size_t)cols, strip_sps(c), strip_ilv(c));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "k_gather_strips", __FILE__, __LINE__);
}

static int ensure_zero_center(bioen_hip_ctx* c) {
    if (c->zero_center) return 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->zero_center), (size_t)c->mp * sizeof(double));
    if (e != hipSuccess) {
        c->zero_center = nullptr;
        return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
    }
    e = hipMemsetAsync(c->zero_center, 0, (size_t)c->mp * sizeof(double), c->stream);
    return e == hipSuccess ? 0 : hip_fail(e, "hipMemsetAsync", __FILE__, __LINE__);
}

static int ensure_center(bioen_hip_ctx* c) {
    if (c->strip_center) return 0;
    double* cen = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&cen), (size_t)c->mp * sizeof(double));
    if (e != hipSuccess) return strip_copy_failed(c, nullptr, e, "hipMalloc (strip centre)");
    e = hipMemcpyAsync(cen, c->YT, (size_t)c->mp * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
    if (e != hipSuccess) {
        (void)hipFree(cen);
        return strip_copy_failed(c, nullptr, e, "hipMemcpyAsync (strip centre)");
    }
    c->strip_center = cen;
    return 0;
}

// reduced-byte storage experiment: the centred copies of c->storage's format, both operand orders, from the row-major
// FP64 matrix (which stays resident: read-back, chi_squared and the forces method keep using it)
static int reduced_rows(const bioen_hip_ctx* c) {       // rows of a reduced strip: whole 64-row slices; 512 < M <= 1024: pairs of
    return (int)round_up((size_t)c->m, c->mp > 512 ? 2 * kWaveRows : kWaveRows);     // them (k_strip2's waves own 128 rows)
}
static size_t reduced_copy_bytes(const bioen_hip_ctx* c) {
    const size_t slices = (size_t)(c->ld / kStripCols) * (reduced_rows(c) / kWaveRows);
    return slices * (c->storage == 1 ? reduced_slice_bytes<1>() : reduced_slice_bytes<2>());
}
static int ensure_reduced_copy(bioen_hip_ctx* c, bool colsum) {
    void*& slot = colsum ? c->Yr1 : c->Yr;
    if (slot) return 0;
    if (paneled(c)) return BIOEN_HIP_ESTATE;
    int rc = ensure_rowmajor(c);                        // (gathered back exactly if the FP64 strip copy had replaced it)
    if (rc) return rc;
    if ((rc = ensure_center(c))) return rc;
    if (ensure_zero_center(c)) return BIOEN_HIP_ENOMEM;
    void* buf = nullptr;
    hipError_t e = hipMalloc(&buf, reduced_copy_bytes(c));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        hip_fail(e, "hipMalloc (reduced-storage strip copy)", __FILE__, __LINE__);
        return BIOEN_HIP_ENOMEM;
    }
    const int nstrips = (int)(c->ld / kStripCols), mps64 = reduced_rows(c);
    const dim3 grid(std::min(nstrips, 4096)), block(256);
    unsigned char* out = static_cast<unsigned char*>(buf);
    if (c->storage == 1 && !colsum) hipLaunchKernelGGL((k_build_strips_reduced<false, 1>), grid, block, 0, c->stream, c->Y, c->ld, c->mp, mps64, c->n, out, nstrips, c->strip_center);
    if (c->storage == 1 && colsum) hipLaunchKernelGGL((k_build_strips_reduced<true, 1>), grid, block, 0, c->stream, c->Y, c->ld, c->mp, mps64, c->n, out, nstrips, c->strip_center);
    if (c->storage == 2 && !colsum) hipLaunchKernelGGL((k_build_strips_reduced<false, 2>), grid, block, 0, c->stream, c->Y, c->ld, c->mp, mps64, c->n, out, nstrips, c->strip_center);
    if (c->storage == 2 && colsum) hipLaunchKernelGGL((k_build_strips_reduced<true, 2>), grid, block, 0, c->stream, c->Y, c->ld, c->mp, mps64, c->n, out, nstrips, c->strip_center);
    e = hipGetLastError();
    if (e != hipSuccess) {
        (void)hipFree(buf);
        return hip_fail(e, "k_build_strips_reduced", __FILE__, __LINE__);
    }
    slot = buf;
    return 0;
}

// bioen_hip_ctx_set_storage: switch the format of the log-weights passes' copies (0 = FP64).  Copies of another format
// are dropped; the row-major FP64 matrix is made resident again and kept from now on.
int set_storage_format(bioen_hip_ctx* c, int fmt) {
    if (fmt == c->storage) return 0;
    if (fmt != 0 && paneled(c)) return BIOEN_HIP_ESTATE;
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
    if (fmt != 0) {
        const int rc = ensure_rowmajor(c);
        if (rc && !(rc == BIOEN_HIP_ESTATE && c->Y)) return rc;
        c->keep_rowmajor = 1;
    }
    if (c->Yr) (void)hipFree(c->Yr);
    if (c->Yr1) (void)hipFree(c->Yr1);
    c->Yr = c->Yr1 = nullptr;
    c->storage = fmt;
    return 0;
}

// The existing row-sum order copies into the layout `want` (strip_phys), best effort: new buffers, whole strips moved, the old
// ones freed -- 2 x the copy's bytes of traffic (3 ms at the headline) and, for the moment of the move, a second copy's
// memory; if that is not to be had the copies stay as they are (every kernel reads either layout).
static void relayout_strip_copies(bioen_hip_ctx* c, int want) {
    if (strip_ilv(c) == want) return;
    const int nstrips = (int)(c->ld / kStripCols);
    const int np = paneled(c) ? panel_count(c) : 1;
    double* made[bioen_hip_ctx::kMaxPanels] = {};
    for (int p = 0; p < np; ++p) {
        const int mps = paneled(c) ? panel_mps(c, p) : strip_rows(c);
        const double* from = paneled(c) ? c->Yp[p] : c->Ys;
        hipError_t e = strip_malloc(c, &made[p], (size_t)nstrips * mps * kStripCols * sizeof(double));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_relayout, dim3(std::min(nstrips, 4096)), dim3(256), 0, c->stream, from, made[p], mps, nstrips,
                               strip_sps(c), strip_ilv(c), want);
            e = hipGetLastError();
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(c->stream);
            for (int q = 0; q <= p; ++q)
                if (made[q]) (void)hipFree(made[q]);
            return;
        }
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) {          // (reported by the next launch check)
        for (int p = 0; p < np; ++p) (void)hipFree(made[p]);
        return;
    }
    for (int p = 0; p < np; ++p) {
        double*& slot = paneled(c) ? c->Yp[p] : c->Ys;
        (void)hipFree(slot);
        slot = made[p];
    }
    c->strip_ilv = want;
    ++c->strip_relayouts;
}

// method: 0 = the log-weights passes are about to run on the copy, 1 = the forces passes, -1 = any layout will do
int ensure_strip_copy(bioen_hip_ctx* c, int method) {
    if (c->storage) return ensure_reduced_copy(c, false);
    if (method >= 0 && (paneled(c) ? c->Yp[0] : c->Ys) != nullptr)
        relayout_strip_copies(c, strip_ilv_wanted(c, method == 1));
    if (!(paneled(c) ? c->Yp[0] : c->Ys)) c->strip_ilv = strip_ilv_wanted(c, method == 1);     // the layout they are built in
    if (paneled(c)) {                                   // row panels of a matrix taller than 1024 rows; Y stays
        if (c->Yp[0]) return 0;
        if (c->strips_unavailable) return BIOEN_HIP_ENOMEM;
        if (!c->Y) return BIOEN_HIP_ESTATE;
        int rc = ensure_center(c);
        if (rc) return rc;
        if (ensure_zero_center(c)) return strip_copy_failed(c, nullptr, hipErrorOutOfMemory, "zero centre");
        const int nstrips = (int)(c->ld / kStripCols);
        double* made[bioen_hip_ctx::kMaxPanels] = {};
        for (int p = 0; p < panel_count(c); ++p) {
            const int mps = panel_mps(c, p);
            hipError_t e = strip_malloc(c, &made[p], (size_t)nstrips * mps * kStripCols * sizeof(double));
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_build_strips<false>, dim3(std::min(nstrips, 4096)), dim3(256), 0, c->stream,
                                   c->Y + (size_t)p * kPanelRows * c->ld, c->ld, panel_mp(c, p), mps, c->n, made[p], nstrips,
                                   c->strip_center + (size_t)p * kPanelRows, strip_sps(c), strip_ilv(c));
                e = hipGetLastError();
            }
            if (e != hipSuccess) {
                for (int q = 0; q <= p; ++q)
                    if (made[q]) (void)hipFree(made[q]);
                return strip_copy_failed(c, nullptr, e, "strip copies of the row panels");
            }
        }
        for (int p = 0; p < panel_count(c); ++p) c->Yp[p] = made[p];
        if (!c->keep_rowmajor) {                        // the panels hold the same numbers: the row-major form has served
            hipError_t e = hipStreamSynchronize(c->stream);
            if (e == hipSuccess) e = hipFree(c->Y);
            if (e != hipSuccess) return hip_fail(e, "release of the row-major matrix", __FILE__, __LINE__);
            c->Y = nullptr;
        }
        return 0;
    }
    if (c->Ys) return 0;
    if (c->strips_unavailable) return BIOEN_HIP_ENOMEM;
    const int mps = strip_rows(c);
    const int nstrips = (int)(c->ld / kStripCols);
    double* ys = nullptr;
    hipError_t e = strip_malloc(c, &ys, (size_t)nstrips * mps * kStripCols * sizeof(double));
    if (e != hipSuccess) return strip_copy_failed(c, nullptr, e, "hipMalloc (strip-major copy of yTilde)");
    if (!c->strip_center) {
        double* cen = nullptr;
        e = hipMalloc(reinterpret_cast<void**>(&cen), (size_t)c->mp * sizeof(double));
        if (e != hipSuccess) return strip_copy_failed(c, ys, e, "hipMalloc (strip centre)");
        e = hipMemcpyAsync(cen, c->YT, (size_t)c->mp * sizeof(double), hipMemcpyDeviceToDevice, c->stream);
        if (e != hipSuccess) {
            (void)hipFree(cen);
            return strip_copy_failed(c, ys, e, "hipMemcpyAsync (strip centre)");
        }
        c->strip_center = cen;
    }
    if (ensure_zero_center(c)) return strip_copy_failed(c, ys, hipErrorOutOfMemory, "zero centre");
    hipLaunchKernelGGL(k_build_strips<false>, dim3(std::min(nstrips, 4096)), dim3(256), 0, c->stream, c->Y, c->ld, c->mp, mps,
                       c->n, ys, nstrips, c->strip_center, strip_sps(c), strip_ilv(c));
    e = hipGetLastError();
    if (e != hipSuccess) return strip_copy_failed(c, ys, e, "k_build_strips");
    c->Ys = ys;
    // the row-major form has served: every path of this context that still wants it (forces_weights' streaming
    // kernels, the r01 kernels of an A/B run) gets it back through ensure_rowmajor.  BIOEN_HIP_KEEP_ROWMAJOR=1 keeps it (A/B).
    if (!c->keep_rowmajor) {
        e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipFree(c->Y);
        if (e != hipSuccess) return hip_fail(e, "release of the row-major matrix", __FILE__, __LINE__);
        c->Y = nullptr;
    }
    return 0;
}

// More than 64 KB of dynamic LDS needs an opt-in -- on the CURRENT device's copy of the kernel: once per (kernel, device),
// contexts of one process may sit on different devices
template <auto Kernel>
static void allow_big_lds(const bioen_hip_ctx* c) {
    static std::atomic<unsigned long long> done{0};
    const unsigned long long bit = 1ull << (c->device & 63);
    if (done.load(std::memory_order_relaxed) & bit) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done.fetch_or(bit, std::memory_order_relaxed);
}

template <int K, bool NT, bool XY, int DEPTH>
static void strip_launch_kd(bioen_hip_ctx* c, const StripArgs& q, const ForcesRound& fr, dim3 block, size_t lds) {
    allow_big_lds<&k_strip<K, NT, XY, DEPTH>>(c);
    BIOEN_LAUNCH_TIMED(c, (k_strip<K, NT, XY, DEPTH>), dim3(q.nblk), block, lds, q, fr);
}
// the same on the reduced-storage copies (experiment)
template <int K, bool NT, bool XY, int DEPTH, int STORE>
static void strip_launch_kds(bioen_hip_ctx* c, const StripArgs& q, const ForcesRound& fr, dim3 block, size_t lds) {
    allow_big_lds<&k_strip<K, NT, XY, DEPTH, STORE>>(c);
    BIOEN_LAUNCH_TIMED(c, (k_strip<K, NT, XY, DEPTH, STORE>), dim3(q.nblk), block, lds, q, fr);
}

// strips in flight per wave: two register sets, except where the second one does not fit (K > 4, pass 1: the
// compiler spilled; experiment knob BIOEN_HIP_STRIP_DEPTH5)
template <int K, bool NT, bool XY>
static void strip_launch_k(bioen_hip_ctx* c, const StripArgs& q, const ForcesRound& fr, dim3 block, size_t lds) {
    static int depth5 = -1;
    if (depth5 < 0) {
        const char* e = std::getenv("BIOEN_HIP_STRIP_DEPTH5");
        depth5 = e ? std::atoi(e) : 3;
    }
    // K > 4 (r04): one register set + the row-sum product deferred behind the next strip's barrier (DEPTH 3); the LDS
    // request carries the doubled partial-sum / e | t / rescale buffers.  BIOEN_HIP_STRIP_DEPTH5=2 / 1: the r03 forms (A/B)
    if (c->storage) {               // reduced-storage experiment: the default forms only
        constexpr int D = K > 4 ? 3 : 2;
        const size_t l2 = lds + (D == 3 ? ((block.x / 64) * 128 + 128 + 16) * sizeof(double) : 0);
        if (c->storage == 1) strip_launch_kds<K, NT, XY, D, 1>(c, q, fr, block, l2);
        else strip_launch_kds<K, NT, XY, D, 2>(c, q, fr, block, l2);
        return;
    }
    if constexpr (K > 4) {
        if (depth5 == 1) { strip_launch_kd<K, NT, XY, 1>(c, q, fr, block, lds); return; }
        if (depth5 != 2) { strip_launch_kd<K, NT, XY, 3>(c, q, fr, block, lds + ((block.x / 64) * 128 + 128 + 16) * sizeof(double)); return; }
    }
    strip_launch_kd<K, NT, XY, 2>(c, q, fr, block, lds);
}

template <int K, bool NT, bool XY>
static void strip2_launch_k(bioen_hip_ctx* c, const StripArgs& q, const ForcesRound& fr, dim3 block, size_t lds) {
    allow_big_lds<&k_strip2<K, NT, XY>>(c);
    if (c->storage) {
        if (c->storage == 1) {
            allow_big_lds<&k_strip2<K, NT, XY, 1>>(c);
            BIOEN_LAUNCH_TIMED(c, (k_strip2<K, NT, XY, 1>), dim3(q.nblk), block, lds, q, fr);
        } else {
            allow_big_lds<&k_strip2<K, NT, XY, 2>>(c);
            BIOEN_LAUNCH_TIMED(c, (k_strip2<K, NT, XY, 2>), dim3(q.nblk), block, lds, q, fr);
        }
        return;
    }
    BIOEN_LAUNCH_TIMED(c, (k_strip2<K, NT, XY>), dim3(q.nblk), block, lds, q, fr);
}

// ADJ forms (the log-weights adjoint on the row-sum order copy: ctx.hpp, one_copy)
template <int K, bool NT>
static void strip_adj_launch_k(bioen_hip_ctx* c, const StripArgs& q, const ForcesRound& fr, dim3 block, size_t lds, bool tall) {
    if (tall) {
        allow_big_lds<&k_strip2<K, NT, true, 0, true>>(c);
        BIOEN_LAUNCH_TIMED(c, (k_strip2<K, NT, true, 0, true>), dim3(q.nblk), block, lds, q, fr);
    } else {
        allow_big_lds<&k_strip<K, NT, true, 2, 0, true>>(c);
        BIOEN_LAUNCH_TIMED(c, (k_strip<K, NT, true, 2, 0, true>), dim3(q.nblk), block, lds, q, fr);
    }
}
template <bool NT>
static void strip_adj_launch_nt(bioen_hip_ctx* c, const StripArgs& q, const ForcesRound& fr, dim3 block, size_t lds, bool tall) {
    switch (fr.n) {
        case 1: strip_adj_launch_k<1, NT>(c, q, fr, block, lds, tall); break;
        case 2: strip_adj_launch_k<2, NT>(c, q, fr, block, lds, tall); break;
        case 3: strip_adj_launch_k<3, NT>(c, q, fr, block, lds, tall); break;
        case 4: strip_adj_launch_k<4, NT>(c, q, fr, block, lds, tall); break;
        case 5: strip_adj_launch_k<5, NT>(c, q, fr, block, lds, tall); break;
        case 6: strip_adj_launch_k<6, NT>(c, q, fr, block, lds, tall); break;
        case 7: strip_adj_launch_k<7, NT>(c, q, fr, block, lds, tall); break;
        default: strip_adj_launch_k<8, NT>(c, q, fr, block, lds, tall); break;
    }
}

template <bool NT, bool XY>
static void strip_launch_nt(bioen_hip_ctx* c, const StripArgs& q, const ForcesRound& fr, dim3 block, size_t lds) {
    if (strip_tall(c)) {
        switch (fr.n) {
            case 1: strip2_launch_k<1, NT, XY>(c, q, fr, block, lds); break;
            case 2: strip2_launch_k<2, NT, XY>(c, q, fr, block, lds); break;
            case 3: strip2_launch_k<3, NT, XY>(c, q, fr, block, lds); break;
            case 4: strip2_launch_k<4, NT, XY>(c, q, fr, block, lds); break;
            case 5: strip2_launch_k<5, NT, XY>(c, q, fr, block, lds); break;
            case 6: strip2_launch_k<6, NT, XY>(c, q, fr, block, lds); break;
            case 7: strip2_launch_k<7, NT, XY>(c, q, fr, block, lds); break;
            default: strip2_launch_k<8, NT, XY>(c, q, fr, block, lds); break;
        }
        return;
    }
    switch (fr.n) {
        case 1: strip_launch_k<1, NT, XY>(c, q, fr, block, lds); break;
        case 2: strip_launch_k<2, NT, XY>(c, q, fr, block, lds); break;
        case 3: strip_launch_k<3, NT, XY>(c, q, fr, block, lds); break;
        case 4: strip_launch_k<4, NT, XY>(c, q, fr, block, lds); break;
        case 5: strip_launch_k<5, NT, XY>(c, q, fr, block, lds); break;
        case 6: strip_launch_k<6, NT, XY>(c, q, fr, block, lds); break;
        case 7: strip_launch_k<7, NT, XY>(c, q, fr, block, lds); break;
        default: strip_launch_k<8, NT, XY>(c, q, fr, block, lds); break;
    }
}

// geometry of the 16-wave kernels (k_strip_fwd / k_strip_adj): waves per strip slot, slots per block
static int fa_wps_rows(int mps) { return std::max(2, (mps + kWaveRows - 1) / kWaveRows); }
static int fa_spb_rows(int mps) { return std::max(1, std::min(16 / fa_wps_rows(mps), 4)); }
static int fa_spb(const bioen_hip_ctx* c) { return fa_spb_rows(paneled(c) ? kPanelRows : strip_rows(c)); }

// forward pass of the log-weights method on the strip copy (all K <= 8): the number of partial sets
StripSets strip_sets(const bioen_hip_ctx* c) {        // paneled: every panel uses the sets of a full 1024-row panel
    return make_sets(c, 32 * fa_spb(c), true);
}

// > 0: the log-weights matrix passes run on the strip copies; the value = sets of the forward pass that reach memory
// on this context (all local segments: what a consumer that totals them as one run is given)
int fwd_strip_blocks(const bioen_hip_ctx* c) {
    if (c->fwd_stream || c->strips_unavailable) return 0;
    if (paneled(c) && (c->panel_off || panel_count(c) > bioen_hip_ctx::kMaxPanels)) return 0;
    return strip_sets(c).sets * c->vr;
}

template <int K, bool NT>
static void fwd_strip_launch_k(bioen_hip_ctx* c, const StripArgs& q, const Vec8& v, dim3 block) {
    const dim3 grid((q.nslots + q.spb - 1) / q.spb);
    if (c->storage == 1) { BIOEN_LAUNCH_TIMED(c, (k_strip_fwd<K, NT, 1>), grid, block, 0, q, v); }
    else if (c->storage == 2) { BIOEN_LAUNCH_TIMED(c, (k_strip_fwd<K, NT, 2>), grid, block, 0, q, v); }
    else { BIOEN_LAUNCH_TIMED(c, (k_strip_fwd<K, NT>), grid, block, 0, q, v); }
}

template <bool NT>
static void fwd_strip_launch_nt(bioen_hip_ctx* c, const StripArgs& q, const Vec8& v, dim3 block) {
    switch (q.K) {
        case 1: fwd_strip_launch_k<1, NT>(c, q, v, block); break;
        case 2: fwd_strip_launch_k<2, NT>(c, q, v, block); break;
        case 3: fwd_strip_launch_k<3, NT>(c, q, v, block); break;
        case 4: fwd_strip_launch_k<4, NT>(c, q, v, block); break;
        case 5: fwd_strip_launch_k<5, NT>(c, q, v, block); break;
        case 6: fwd_strip_launch_k<6, NT>(c, q, v, block); break;
        case 7: fwd_strip_launch_k<7, NT>(c, q, v, block); break;
        default: fwd_strip_launch_k<8, NT>(c, q, v, block); break;
    }
}

// partial[(row K + a) nblk + block] of Y' . v_a; the caller adds the centre back (k_rows_combine's `center`);
// plain = true: Y . v_a itself (no centring: bioen_hip_chi_squared takes any w, not only normalised ones)
void launch_fwd_strip(bioen_hip_ctx* c, int K, const Vec8& v, int nblk, bool plain) {
    const double* center = plain ? c->zero_center : c->strip_center;
    for (int p = 0; p < panel_count(c); ++p) {
        TimedLaunch tl(c, 0, K);
        const int row0 = p * kPanelRows;
        StripArgs q{};
        q.Ys = paneled(c) ? c->Yp[p] : c->Ys;
        q.center = center + row0;
        q.mps = paneled(c) ? panel_mps(c, p) : strip_rows(c);
        if (c->storage) {                               // reduced-storage experiment: centred copies, rows padded to 64
            q.Ys = static_cast<const double*>(c->Yr);
            q.mps = reduced_rows(c);
        }
        q.mp = panel_mp(c, p);
        q.nstrips = (int)(c->ld / kStripCols);
        q.n = c->n;
        q.K = K;
        (void)nblk;
        const StripSets ss = strip_sets(c);
        q.sps = ss.sps; q.gs = ss.gs; q.tc = ss.tc; q.nch = ss.nch; q.fold = ss.fold; q.slots = ss.slots;
        q.ilv = c->storage ? 1 : strip_ilv(c);          // (the reduced-format copies are kept in strip order)
        q.nslots = ss.slots * c->vr;
        q.partial = c->fwd_partial + (size_t)row0 * K;
        q.pstride = c->mp;
        q.wps = fa_wps_rows(q.mps);
        q.spb = fa_spb(c);                               // the sets are those of the context's geometry in every panel
        if (q.wps * q.spb > 16) q.wps = 16 / q.spb;     // (cannot happen: a shorter last panel needs fewer waves per slot)
        const dim3 block(64 * q.wps * q.spb);
        if (c->nontemporal) fwd_strip_launch_nt<true>(c, q, v, block);
        else fwd_strip_launch_nt<false>(c, q, v, block);
    }
}

// ONE strip copy or two?  (r06: the default depends on the size.)  one_copy_wanted: 1 / 0 = asked for / refused
// (BIOEN_HIP_ONE_COPY, bioen_hip_ctx_set_one_copy), -1 = by size: a second copy of more than 1 GiB is not made.  With the
// one-copy adjoint taking the copy's positions in order (forces_slot: flat) it runs at the two-copy kernel's time wherever
// the matrix is large (profiles/r06_onecopy_ab.txt: headline sweep 1.306-1.316 s on one copy against 1.307-1.315 s on two,
// adjoint 1.203-1.209 against 1.201-1.212 ms; M = 512 x 1e6, 1024 x 1.25e5, 256 x 1e5: equal; M <= 128: a launch of the
// LDS-image kernel costs 12 us against 6), so the 8.2 GB the headline's second copy took are no longer spent by default;
// small problems keep the faster dedicated kernel.  Decided on the GLOBAL matrix (every rank of a sharded context takes
// the same form: the bits of a result must not depend on the GPU count).
static bool one_copy_by_default(const bioen_hip_ctx* c) {
    if (c->one_copy_wanted >= 0) return c->one_copy_wanted == 1;
    const double global_copy_bytes = (double)round_up((size_t)c->m, 16) * (double)c->n_global * sizeof(double);
    return global_copy_bytes > 1024.0 * 1024.0 * 1024.0;
}

// adjoint pass of the log-weights method on the column-sum copy (built on first use)
int ensure_strip_copy_colsum(bioen_hip_ctx* c) {
    if (c->storage) {
        const int rc = ensure_reduced_copy(c, false);
        return rc ? rc : ensure_reduced_copy(c, true);
    }
    if (paneled(c)) {
        if (c->Y1p[0]) return 0;
        int rc = ensure_strip_copy(c);
        if (rc) return rc;
        if (c->one_copy) return 0;                                // (below: the one-copy form, asked for or taken)
        if (one_copy_by_default(c)) {
            c->one_copy = 1;
            return 0;
        }
        const int nstrips = (int)(c->ld / kStripCols);
        double* made[bioen_hip_ctx::kMaxPanels] = {};
        for (int p = 0; p < panel_count(c); ++p) {
            const int mps = panel_mps(c, p);
            hipError_t e = strip_malloc(c, &made[p], (size_t)nstrips * mps * kStripCols * sizeof(double));
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_restripe, dim3(std::min(nstrips, 4096)), dim3(256), 0, c->stream, c->Yp[p], mps, made[p], nstrips, strip_sps(c), strip_ilv(c));
                e = hipGetLastError();
            }
            if (e != hipSuccess) {
                for (int q = 0; q <= p; ++q)
                    if (made[q]) (void)hipFree(made[q]);
                (void)hipGetLastError();
                c->one_copy = 1;                                  // the row-sum order panels serve both products
                return 0;
            }
        }
        for (int p = 0; p < panel_count(c); ++p) c->Y1p[p] = made[p];
        return 0;
    }
    if (c->Ys1) return 0;
    int rc = ensure_strip_copy(c);                               // the centre is shared; the column-sum copy is cut from the row-sum one
    if (rc) return rc;
    // ONE strip copy (r05; ctx.hpp: one_copy): asked for (BIOEN_HIP_ONE_COPY=1), the default of a large matrix (r06:
    // one_copy_by_default), or taken when the second copy does not
    // fit -- the adjoint then runs on the row-sum order copy (launch_adj_strip) at the forces kernels' rate, instead of the
    // whole context falling back to the streaming kernels on the row-major matrix
    const bool can_one = true;                                   // (the ADJ forms of k_strip / k_strip2 serve every strip height)
    if (c->one_copy) return 0;
    if (one_copy_by_default(c) && can_one) {
        c->one_copy = 1;
        return 0;
    }
    const int mps = strip_rows(c);
    const int nstrips = (int)(c->ld / kStripCols);
    double* ys = nullptr;
    hipError_t e = strip_malloc(c, &ys, (size_t)nstrips * mps * kStripCols * sizeof(double));
    if (e != hipSuccess && can_one) {
        (void)hipGetLastError();                                 // an out-of-memory error must not surface at the next launch check
        c->one_copy = 1;
        return 0;
    }
    if (e != hipSuccess) return strip_copy_failed(c, nullptr, e, "hipMalloc (column-sum strip copy of yTilde)");
    hipLaunchKernelGGL(k_restripe, dim3(std::min(nstrips, 4096)), dim3(256), 0, c->stream, c->Ys, mps, ys, nstrips, strip_sps(c), strip_ilv(c));
    e = hipGetLastError();
    if (e != hipSuccess) return strip_copy_failed(c, ys, e, "k_restripe");
    c->Ys1 = ys;
    return 0;
}

static size_t adj_strip_lds_bytes(const bioen_hip_ctx* c, int wps) {
    const int waves = wps * fa_spb(c);          // u table of one slot's rows | their centres | two parity buffers of partial sums
    return ((size_t)wps * kWaveRows * 9 + (size_t)2 * waves * 128) * sizeof(double);
}

template <int K, bool NT>
static void adj_strip_launch_k(bioen_hip_ctx* c, const StripArgs& q, const MVec8& out, const MVec8& scal, dim3 block, size_t lds) {
    allow_big_lds<&k_strip_adj<K, NT>>(c);
    const dim3 grid((q.nblk + q.spb - 1) / q.spb);
    if (c->storage == 1) {
        allow_big_lds<&k_strip_adj<K, NT, 1>>(c);
        BIOEN_LAUNCH_TIMED(c, (k_strip_adj<K, NT, 1>), grid, block, lds, q, out, scal);
    } else if (c->storage == 2) {
        allow_big_lds<&k_strip_adj<K, NT, 2>>(c);
        BIOEN_LAUNCH_TIMED(c, (k_strip_adj<K, NT, 2>), grid, block, lds, q, out, scal);
    } else {
        BIOEN_LAUNCH_TIMED(c, (k_strip_adj<K, NT>), grid, block, lds, q, out, scal);
    }
}

template <bool NT>
static void adj_strip_launch_nt(bioen_hip_ctx* c, const StripArgs& q, const MVec8& out, const MVec8& scal, dim3 block, size_t lds) {
    switch (q.K) {
        case 1: adj_strip_launch_k<1, NT>(c, q, out, scal, block, lds); break;
        case 2: adj_strip_launch_k<2, NT>(c, q, out, scal, block, lds); break;
        case 3: adj_strip_launch_k<3, NT>(c, q, out, scal, block, lds); break;
        case 4: adj_strip_launch_k<4, NT>(c, q, out, scal, block, lds); break;
        case 5: adj_strip_launch_k<5, NT>(c, q, out, scal, block, lds); break;
        case 6: adj_strip_launch_k<6, NT>(c, q, out, scal, block, lds); break;
        case 7: adj_strip_launch_k<7, NT>(c, q, out, scal, block, lds); break;
        default: adj_strip_launch_k<8, NT>(c, q, out, scal, block, lds); break;
    }
}

// out_a[j] = sum_i u_c[i K + a] (Y_ij - ybar_c[i K + a]) with the RAW ybar in ybar_c; needs S_B0 / S_UY of this
// round in the problems' scalars (k_rows_combine with the strip centre)
void launch_adj_strip(bioen_hip_ctx* c, int K, const double* u_c, const MVec8& out, const MVec8& scal, int nblk, bool plain) {
    if (c->one_copy && !c->storage) {
        // ONE strip copy (r05): the product runs on the row-sum order copy through the forces kernels' LDS image (k_strip /
        // k_strip2 in their ADJ form; a matrix taller than 1024 rows: panel by panel, continuing the column sums) -- no sum
        // over columns: any assignment of strips to blocks gives the same bits
        for (int p = 0; p < panel_count(c); ++p) {
            TimedLaunch tl(c, 1, K);
            const int row0 = p * kPanelRows;
            StripArgs q{};
            q.Ys = paneled(c) ? c->Yp[p] : c->Ys;
            q.center = (plain ? c->zero_center : c->strip_center) + row0;
            q.mps = paneled(c) ? panel_mps(c, p) : strip_rows(c);
            q.mp = panel_mp(c, p);
            q.nstrips = (int)(c->ld / kStripCols);
            q.n = c->n;
            q.K = K;
            q.sps = c->segcols / kStripCols;
            // as many blocks per CU as the forces passes run (M <= 128: four of two waves, <= 256: two of four, else one)
            q.gs = std::max(1, std::min(q.sps, 256 * (paneled(c) ? 1 : forces_per_cu(c))));
            q.tc = (q.sps + q.gs - 1) / q.gs;
            q.nch = 1; q.fold = 0; q.slots = q.gs;
            q.nslots = q.gs;
            q.nlocal = c->vr;
            q.ilv = strip_ilv(c);
            q.nblk = q.nslots;
            q.u_c = u_c + (size_t)row0 * K;
            q.accumulate = p > 0 ? 1 : (plain ? 2 : 0);
            q.w0 = c->fixed;
            q.partial = c->fwd_partial;
            ForcesRound fr{};
            fr.n = K;
            for (int a = 0; a < K; ++a) {
                fr.a[a] = out.p[a];
                fr.scal[a] = scal.p[a];
            }
            const bool tall = q.mp > 512;                       // k_strip2: 128 rows per wave
            const int waves = std::max(2, tall ? (q.mps + 2 * kWaveRows - 1) / (2 * kWaveRows) : (q.mps + kWaveRows - 1) / kWaveRows);
            const dim3 block(64 * waves);
            const size_t lds = tall ? ((size_t)waves * kWaveRows * kStripCols + (size_t)waves * 2 * kWaveRows * (8 + 1) + (size_t)waves * 128 + 128 + 16) * sizeof(double)
                                    : ((size_t)waves * kWaveRows * (kStripCols + 8 + 1) + (size_t)waves * 128 + 128 + 16) * sizeof(double);
            if (c->nontemporal) strip_adj_launch_nt<true>(c, q, fr, block, lds, tall);
            else strip_adj_launch_nt<false>(c, q, fr, block, lds, tall);
        }
        return;
    }
    for (int p = 0; p < panel_count(c); ++p) {
        TimedLaunch tl(c, 1, K);
        const int row0 = p * kPanelRows;
        StripArgs q{};
        q.Ys = paneled(c) ? c->Y1p[p] : c->Ys1;
        q.center = (plain ? c->zero_center : c->strip_center) + row0;      // plain: out = Y^T u itself (forces method, M > 1024)
        q.mps = paneled(c) ? panel_mps(c, p) : strip_rows(c);
        if (c->storage) {
            q.Ys = static_cast<const double*>(c->Yr1);
            q.mps = reduced_rows(c);
        }
        q.mp = panel_mp(c, p);
        q.nstrips = (int)(c->ld / kStripCols);
        q.n = c->n;
        q.K = K;
        (void)nblk;                                      // no sum over columns here: any assignment of strips to slots gives the same bits
        q.nblk = std::min(256 * fa_spb(c), q.nstrips);
        q.u_c = u_c + (size_t)row0 * K;
        q.accumulate = p > 0 ? 1 : (plain ? 2 : 0);
        q.wps = fa_wps_rows(q.mps);
        q.spb = fa_spb(c);
        const dim3 block(64 * q.wps * q.spb);
        const size_t lds = adj_strip_lds_bytes(c, q.wps);
        if (c->nontemporal) adj_strip_launch_nt<true>(c, q, out, scal, block, lds);
        else adj_strip_launch_nt<false>(c, q, out, scal, block, lds);
    }
}

template <bool XY>
static void strip_launch(bioen_hip_ctx* c, const ForcesRound& fr, int nblk, const double* u_c) {
    StripArgs q{};
    q.Ys = c->Ys;
    q.center = c->strip_center;
    q.mps = strip_rows(c);
    if (c->storage) {                                   // reduced-storage experiment: the centred row-sum order copy
        q.Ys = static_cast<const double*>(c->Yr);
        q.mps = reduced_rows(c);
    }
    q.mp = c->mp;
    q.nstrips = (int)(c->ld / kStripCols);
    q.n = c->n;
    q.K = fr.n;
    (void)nblk;
    const StripSets ss = forces_sets(c);
    q.sps = ss.sps; q.gs = ss.gs; q.tc = ss.tc; q.nch = 1; q.fold = 0; q.slots = ss.gs;
    q.nslots = ss.gs;                                  // one block per group; it runs the local segments in turn
    q.nlocal = c->vr;
    q.ilv = c->storage ? 1 : strip_ilv(c);
    q.nblk = q.nslots;
    q.u_c = u_c;
    q.w0 = c->fixed;
    q.partial = c->fwd_partial;
    q.stamps = reinterpret_cast<long long*>(c->strip_stamps);
    const dim3 block(strip_threads(c));
    const size_t lds = strip_lds_bytes(c);
    if (c->nontemporal) strip_launch_nt<true, XY>(c, q, fr, block, lds);
    else strip_launch_nt<false, XY>(c, q, fr, block, lds);
}

// Merge the sets of the xy pass per SEGMENT (one block per problem and local segment; set order): m_v = max_b m_b,
// Z_v = sum_b e^{m_b - m_v} Z_b, likewise sum e x; P_MAX[b] <- e^{m_b - m_v}, the weight of set
// b's raw sums.  The segment totals {Z_v, sum e x, m_v} go to the tail of the segment's part of X_YBAR --
// the layout of the log-weights rounds -- so k_rows_combine<true> finishes both methods alike:
//   scal[S_LOGS] = M + log Z  (w_j = w0_j exp(x_j - S_LOGS)),  scal[S_P] = sum_j w_j x_j,
//   KL = sum_j w_j log(w_j / w0_j) = S_P - S_LOGS   (c_bioen_kernels_forces.c:246-258, with
//   log w_j - log w0_j = x_j - S_LOGS; the prior constant S_LOGS0 is zero for this method).
__global__ __launch_bounds__(kBlock) void k_forces_blockstats(ForcesRound fr, int seg_sets, int mp, int K, Xch xo) {
    __shared__ double sh[kWaves];
    const int a = blockIdx.y, v = blockIdx.z;
    double* pa = fr.part[a];
    double* pm = pa + (size_t)P_MAX * kPartStride + (size_t)v * seg_sets;
    const double* ps = pa + (size_t)P_SUM * kPartStride + (size_t)v * seg_sets;
    const double* pp = pa + (size_t)P_PP * kPartStride + (size_t)v * seg_sets;
    const double mr = max_partials(pm, seg_sets, sh);
    double z = 0.0, px = 0.0;
    for (int b = threadIdx.x; b < seg_sets; b += kBlock) {
        const double fb = exp(pm[b] - mr);
        z = fma(fb, ps[b], z);
        px = fma(fb, pp[b], px);
    }
    z = block_sum(z, sh);
    px = block_sum(px, sh);
    __syncthreads();
    for (int b = threadIdx.x; b < seg_sets; b += kBlock) pm[b] = exp(pm[b] - mr);
    if (threadIdx.x == 0) {
        double* tail = xo.base + (size_t)(xo.rank + v) * xo.payload + (size_t)mp * K + 3 * a;
        tail[0] = z;
        tail[1] = px;
        tail[2] = mr;
        fr.scal[a][S_LOGS0] = 0.0;
    }
}

// a segment's share of ybar' : sum_b weight_b raw_i,b over the segment's sets, on transposed partials (the strip kernels
// of this file); blockIdx.y = local segment
struct TermWeighted {
    const double* wb;
    __device__ __forceinline__ double operator()(int b, double v, double s) const { return fma(wb[b], v, s); }
};

__global__ __launch_bounds__(kBlock) void k_forces_rows_weighted_t(const double* __restrict__ partial, int seg_sets, int mp,
                                                                   int K, ForcesRound fr, Xch xo) {
    __shared__ double lds[8][32];
    const int v = blockIdx.y;
    const size_t n = (size_t)mp * K;
    const size_t idx = (size_t)blockIdx.x * 32 + (threadIdx.x & 31);
    const bool valid = idx < n;
    const int a = (int)(idx % K);
    const double* wb = fr.part[0];
#pragma unroll
    for (int k = 1; k < kMaxBatch; ++k)
        if (k == a) wb = fr.part[k];
    const double s = sets_sum8(partial + (size_t)v * seg_sets * n + (valid ? idx : 0), n, seg_sets, 1, lds,
                               TermWeighted{wb + (size_t)P_MAX * kPartStride + (size_t)v * seg_sets});
    if (threadIdx.x < 32 && valid) (xo.base + (size_t)(xo.rank + v) * xo.payload)[idx] = s;
}

// w_j = w0_j exp(x_j - S_LOGS): the weights themselves, when a result is handed out
__global__ __launch_bounds__(kBlock) void k_forces_w_from_x(ForcesRound fr, const double* __restrict__ w0, int n) {
    const int a = blockIdx.y;
    const double* __restrict__ x = fr.a[a];
    double* __restrict__ w = fr.w[a];
    const double logz = fr.scal[a][S_LOGS];
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) w[j] = w0[j] * exp(x[j] - logz);
}

void launch_forces_blockmerge(bioen_hip_ctx* c, const ForcesRound& fr, int seg_sets, bool tposed) {
    (void)tposed;
    const Xch xo = make_xch(c, X_YBAR, ybar_payload(c, fr.n, true));
    hipLaunchKernelGGL(k_forces_blockstats, dim3(1, fr.n, c->vr), dim3(kBlock), 0, c->stream, fr, seg_sets, c->mp, fr.n, xo);
    hipLaunchKernelGGL(k_forces_rows_weighted_t, dim3((c->mp * fr.n + 31) / 32, c->vr), dim3(kBlock), 0, c->stream,
                       c->fwd_partial, seg_sets, c->mp, fr.n, fr, xo);
}

// pass 1: x' = Y'^T f, online softmax, raw ybar' per block; then the block merge and ybar' -> X_YBAR
void launch_forces_xy(bioen_hip_ctx* c, const ForcesRound& fr, int nblk) {
    {
        TimedLaunch tl(c, 1, fr.n);
        strip_launch<true>(c, fr, nblk, c->um);
    }
    launch_forces_blockmerge(c, fr, nblk, true);
}

// pass 2: b' = Y'^T r, t, Y' . t
void launch_forces_bt(bioen_hip_ctx* c, const ForcesRound& fr, int nblk) {
    TimedLaunch tl(c, 0, fr.n);
    strip_launch<false>(c, fr, nblk, c->r_c);
}

void launch_forces_w_from_x(bioen_hip_ctx* c, const ForcesRound& fr) {
    hipLaunchKernelGGL(k_forces_w_from_x, dim3(vec_blocks(c), fr.n), dim3(kBlock), 0, c->stream, fr, c->fixed, c->n);
}

}  // namespace bioen
