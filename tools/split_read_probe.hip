// Feasibility probe for the reduced-byte storage experiment: what rate does the
// memory system give a read stream of 6-byte (fp32 + bf16 residual) or 4-byte (fp32) elements that are reassembled to
// FP64 in registers, in the strip kernels' access pattern (a wave reads contiguous KiB, 16 bytes per lane), compared
// with the plain 8-byte stream?  hipcc --offload-arch=gfx950 -O3 tools/split_read_probe.hip -o build/split_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <class T>
__device__ __forceinline__ T ldnt(const T* p) { return __builtin_nontemporal_load(p); }

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 8-byte elements: a wave reads 8 KiB per trip (1024 elements)
__global__ __launch_bounds__(1024) void k_f64(const double* __restrict__ p, size_t nelem, double* out) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t base = wave * 1024; base + 1024 <= nelem; base += nw * 1024) {
        d2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ldnt(reinterpret_cast<const d2*>(p + base) + u * 64 + lane);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += v[u].x + v[u].y;
    }
    double t = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    t = wave_sum(t);
    if (lane == 0 && t == 123.456) out[blockIdx.x] = t;
}

// 6-byte elements: per 1024 elements a 4-KiB block of fp32 high parts then a 2-KiB block of bf16 residuals
__global__ __launch_bounds__(1024) void k_split(const unsigned char* __restrict__ p, size_t nelem, double* out) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t base = wave * 1024; base + 1024 <= nelem; base += nw * 1024) {
        const unsigned char* blk = p + base * 6;
        f4 hi[4];
        u4 lo[2];
#pragma unroll
        for (int u = 0; u < 4; ++u) hi[u] = ldnt(reinterpret_cast<const f4*>(blk) + u * 64 + lane);
#pragma unroll
        for (int u = 0; u < 2; ++u) lo[u] = ldnt(reinterpret_cast<const u4*>(blk + 4096) + u * 64 + lane);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u4 l = lo[u >> 1];
            const unsigned w0 = (u & 1) ? l.z : l.x, w1 = (u & 1) ? l.w : l.y;     // two bf16 per word
            const double e0 = (double)hi[u].x + (double)__uint_as_float(w0 << 16);
            const double e1 = (double)hi[u].y + (double)__uint_as_float(w0 & 0xffff0000u);
            const double e2 = (double)hi[u].z + (double)__uint_as_float(w1 << 16);
            const double e3 = (double)hi[u].w + (double)__uint_as_float(w1 & 0xffff0000u);
            acc[2 * u] += e0 + e1;
            acc[2 * u + 1] += e2 + e3;
        }
    }
    double t = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    t = wave_sum(t);
    if (lane == 0 && t == 123.456) out[blockIdx.x] = t;
}

// 4-byte elements
__global__ __launch_bounds__(1024) void k_f32(const float* __restrict__ p, size_t nelem, double* out) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t base = wave * 2048; base + 2048 <= nelem; base += nw * 2048) {
        f4 hi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) hi[u] = ldnt(reinterpret_cast<const f4*>(p + base) + u * 64 + lane);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += ((double)hi[u].x + (double)hi[u].y) + ((double)hi[u].z + (double)hi[u].w);
    }
    double t = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    t = wave_sum(t);
    if (lane == 0 && t == 123.456) out[blockIdx.x] = t;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const size_t nelem = (size_t)1024 * 1000000;          // the headline matrix
    void* buf = nullptr;
    double* out = nullptr;
    CK(hipMalloc(&buf, nelem * 8));
    CK(hipMalloc(&out, 4096 * 8));
    CK(hipMemset(buf, 0, nelem * 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int reps = 10;
    for (int variant = 0; variant < 3; ++variant) {
        float best = 1e30f;
        for (int trial = 0; trial < 3; ++trial) {
            CK(hipEventRecord(a));
            for (int r = 0; r < reps; ++r) {
                if (variant == 0) hipLaunchKernelGGL(k_f64, dim3(256), dim3(1024), 0, 0, (const double*)buf, nelem, out);
                if (variant == 1) hipLaunchKernelGGL(k_split, dim3(256), dim3(1024), 0, 0, (const unsigned char*)buf, nelem, out);
                if (variant == 2) hipLaunchKernelGGL(k_f32, dim3(256), dim3(1024), 0, 0, (const float*)buf, nelem, out);
            }
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        const double bytes = (variant == 0 ? 8.0 : variant == 1 ? 6.0 : 4.0) * nelem;
        std::printf("%s: %.3f ms per pass over %.2e elements, %.2f TB/s, %.3f Telem/s\n",
                    variant == 0 ? "f64 (8 B)" : variant == 1 ? "fp32 + bf16 (6 B)" : "fp32 (4 B)", best / reps, (double)nelem,
                    bytes / (best / reps * 1e-3) / 1e12, nelem / (best / reps * 1e-3) / 1e12);
    }
    return 0;
}
