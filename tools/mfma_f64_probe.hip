// Probe for v_mfma_f64_16x16x4_f64 on gfx950: (1) operand / result lane maps checked with exact
// integer data (asymmetric operands), (2) issue rate on one SIMD: back-to-back MFMAs on 1, 2 and 4
// accumulators, one and two waves per SIMD, against a v_fma_f64 stream of the same FLOP count.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_probe tools/mfma_f64_probe.hip && /tmp/mfma_f64_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// D = A (16x4) . B (4x16): A[i][k] = 1 + i + 16 k, B[k][j] = (k + 1) * (j + 3)  (asymmetric)
__global__ void k_layout(double* out /* 16 x 16 row-major, by the documented map */) {
    const int l = threadIdx.x;
    const double a = 1.0 + (l & 15) + 16.0 * (l >> 4);          // A[i = l&15][k = l>>4]
    const double b = ((l >> 4) + 1.0) * ((l & 15) + 3.0);        // B[k = l>>4][j = l&15]
    v4d c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) out[((l >> 4) + 4 * v) * 16 + (l & 15)] = c[v];   // row = (l>>4) + 4 v, col = l&15
}

template <int ACC>
__global__ __launch_bounds__(512) void k_rate(double* sink, long long* cycles, int iters) {
    const int l = threadIdx.x & 63;
    double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
    v4d c[ACC];
    for (int q = 0; q < ACC; ++q) c[q] = v4d{0.0, 0.0, 0.0, 0.0};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < ACC; ++q) c[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[q], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
    for (int q = 0; q < ACC; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (l == 0) cycles[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

__global__ __launch_bounds__(512) void k_fma_rate(double* sink, long long* cycles, int iters) {
    const int l = threadIdx.x & 63;
    double a = 1.0 + 1e-9 * l;
    double c[16];
    for (int q = 0; q < 16; ++q) c[q] = 1e-3 * q;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) c[q] = __builtin_fma(c[q], a, 1e-7);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
    for (int q = 0; q < 16; ++q) s += c[q];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (l == 0) cycles[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    double* d_out;
    CHECK(hipMalloc(&d_out, 256 * sizeof(double)));
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, d_out);
    std::vector<double> out(256);
    CHECK(hipMemcpy(out.data(), d_out, 256 * sizeof(double), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0.0;
            for (int k = 0; k < 4; ++k) ref += (1.0 + i + 16.0 * k) * ((k + 1.0) * (j + 3.0));
            if (out[i * 16 + j] != ref) ++bad;
        }
    std::printf("layout check (A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4v col=l&15): %d wrong of 256\n", bad);

    const int blocks = 256, iters = 4096;
    double* sink;
    long long* cyc;
    CHECK(hipMalloc(&sink, (size_t)blocks * 512 * sizeof(double)));
    CHECK(hipMalloc(&cyc, (size_t)blocks * 8 * sizeof(long long)));
    std::vector<long long> h((size_t)blocks * 8);
    auto report = [&](const char* name, int waves, int acc_or_ops, double flops_per_iter) {
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), cyc, (size_t)blocks * waves * sizeof(long long), hipMemcpyDeviceToHost));
        double mean = 0.0;
        for (int i = 0; i < blocks * waves; ++i) mean += (double)h[i];
        mean /= blocks * waves;
        // s_memtime ticks at 100 MHz-derived constant rate?  report ticks per instruction; the ratio to the fma row is what matters
        std::printf("%-34s waves/block %d : %.2f ticks per wave-instruction, %.1f FLOP per tick per wave\n", name, waves,
                    mean / ((double)iters * acc_or_ops), flops_per_iter * iters / mean);
        return 0;
    };
    for (int threads : {256, 512}) {
        const int waves = threads / 64;
        hipLaunchKernelGGL((k_rate<1>), dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters);
        report("mfma_f64_16x16x4, 1 accumulator", waves, 1, 2048.0 * 1);
        hipLaunchKernelGGL((k_rate<2>), dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters);
        report("mfma_f64_16x16x4, 2 accumulators", waves, 2, 2048.0 * 2);
        hipLaunchKernelGGL((k_rate<4>), dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters);
        report("mfma_f64_16x16x4, 4 accumulators", waves, 4, 2048.0 * 4);
        hipLaunchKernelGGL(k_fma_rate, dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters);
        report("v_fma_f64 x16 independent", waves, 16, 128.0 * 16);
    }
    return 0;
}
