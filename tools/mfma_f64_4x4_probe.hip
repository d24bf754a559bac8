// Probe for v_mfma_f64_4x4x4_4b_f64 on gfx950: lane maps (searched over the plausible index assignments with
// exact integer data) and issue rate.   hipcc --offload-arch=gfx950 -O3 -o build/mfma_f64_4x4_probe tools/mfma_f64_4x4_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_once(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], 0.0, 0, 0, 0);
}

template <int ACC>
__global__ __launch_bounds__(512) void k_rate(double* sink, long long* cycles, int iters) {
    const int l = threadIdx.x & 63;
    double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
    double c[ACC];
    for (int q = 0; q < ACC; ++q) c[q] = 0.0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < ACC; ++q) c[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[q], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
    for (int q = 0; q < ACC; ++q) s += c[q];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (l == 0) cycles[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    std::vector<double> a(64), b(64), d(64);
    for (int l = 0; l < 64; ++l) { a[l] = 1 + (l * 7) % 61; b[l] = 2 + (l * 11) % 59; }
    double *da, *db, *dd;
    CHECK(hipMalloc(&da, 512)); CHECK(hipMalloc(&db, 512)); CHECK(hipMalloc(&dd, 512));
    CHECK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_once, dim3(1), dim3(64), 0, 0, da, db, dd);
    CHECK(hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost));
    // lane l = 16 blk + 4 p + r.  hypotheses: A element of (blk, i, k) sits in lane 16 blk + 4 X + Y with (X, Y) = (i, k) or (k, i);
    // B element (blk, k, j) in lane 16 blk + 4 X + Y with (X, Y) = (k, j) or (j, k); D (blk, i, j) in lane 16 blk + 4 X + Y, (i, j) or (j, i)
    int found = 0;
    for (int ha = 0; ha < 2; ++ha) for (int hb = 0; hb < 2; ++hb) for (int hd = 0; hd < 2; ++hd) {
        int bad = 0;
        for (int blk = 0; blk < 4; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            double ref = 0.0;
            for (int k = 0; k < 4; ++k) {
                const int la = 16 * blk + (ha ? 4 * k + i : 4 * i + k);
                const int lb = 16 * blk + (hb ? 4 * j + k : 4 * k + j);
                ref += a[la] * b[lb];
            }
            const int ld = 16 * blk + (hd ? 4 * j + i : 4 * i + j);
            if (d[ld] != ref) ++bad;
        }
        if (!bad) { std::printf("layout: A lane = 16 blk + %s, B lane = 16 blk + %s, D lane = 16 blk + %s\n", ha ? "4 k + i" : "4 i + k", hb ? "4 j + k" : "4 k + j", hd ? "4 j + i" : "4 i + j"); ++found; }
    }
    if (!found) { std::printf("no hypothesis matched; D ="); for (int l = 0; l < 64; ++l) std::printf(" %.0f", d[l]); std::printf("\n"); }

    const int blocks = 256, iters = 4096;
    double* sink; long long* cyc;
    CHECK(hipMalloc(&sink, (size_t)blocks * 512 * sizeof(double)));
    CHECK(hipMalloc(&cyc, (size_t)blocks * 8 * sizeof(long long)));
    std::vector<long long> h((size_t)blocks * 8);
    for (int threads : {256, 512}) {
        const int waves = threads / 64;
        for (int acc : {1, 4}) {
            if (acc == 1) hipLaunchKernelGGL((k_rate<1>), dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters);
            else hipLaunchKernelGGL((k_rate<4>), dim3(blocks), dim3(threads), 0, 0, sink, cyc, iters);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h.data(), cyc, (size_t)blocks * waves * sizeof(long long), hipMemcpyDeviceToHost));
            double mean = 0.0;
            for (int i = 0; i < blocks * waves; ++i) mean += (double)h[i];
            mean /= blocks * waves;
            std::printf("mfma_f64_4x4x4_4b, %d accumulator(s), %d waves per block: %.2f ticks per wave-instruction (512 FLOP each)\n", acc, waves, mean / ((double)iters * acc));
        }
    }
    return 0;
}
