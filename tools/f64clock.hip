// f64clock.hip -- what the fp64 matrix cores deliver in a bare loop (the ceiling the float64 kernels are priced against: 78.6 TFLOP/s
// = 64 cycles per v_mfma_f64_16x16x4_f64 and SIMD at 2.4 GHz), and what the shapes of the fused KL tiles cost:
//   mode 0  16 independent accumulators (the NT / TN kernels' MFMA blocks)
//   mode 1  chains of 16 dependent MFMAs on ONE accumulator, one chain after the other (the first product of a KL tile)
//   mode 2  two such chains interleaved
//   mode 3  mode 0 plus four IEEE divisions per lane every 32 MFMAs (the quotient of a KL tile), independent of the MFMAs
//   mode 4  as 3, the divisions fed by an MFMA result and feeding the next MFMAs (the real dependency)
// random operands in registers, one wave per SIMD (and two), every CU busy, ~1 s per measurement.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/_build/f64clock tools/f64clock.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(const double* __restrict__ in, long trips, double* out) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    double a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = in[(tid * 16 + j) & 0xfffff]; b[j] = in[(tid * 16 + 8 + j) & 0xfffff]; }
    f64x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
    double u[4] = {a[0], a[1], a[2], a[3]};
    for (long t = 0; t < trips; ++t) {           // 32 MFMAs per trip
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i & 15] = MFMA64(a[i & 7], b[(i + 3) & 7], acc[i & 15]);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[c] = MFMA64(a[i & 7], b[(i + 3) & 7], acc[c]);
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[0] = MFMA64(a[i & 7], b[(i + 3) & 7], acc[0]);
                acc[1] = MFMA64(a[(i + 1) & 7], b[(i + 4) & 7], acc[1]);
            }
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i & 15] = MFMA64(a[i & 7], b[(i + 3) & 7], acc[i & 15]);
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = a[e] / (u[e] + 1.0e-3);
        } else {
            f64x4 s4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < 16; ++i) s4 = MFMA64(a[i & 7], b[(i + 3) & 7], s4);
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = a[e] / (s4[e] + 1.0e-3);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) acc[tt] = MFMA64(u[e], b[(e + tt) & 7], acc[tt]);
        }
    }
    double s = u[0] + u[1] + u[2] + u[3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) out[0] = s;
}

template <int MODE>
double run(const double* in, double* out, int waves_per_simd, long trips) {
    const int threads = 256 * waves_per_simd, blocks = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, in, trips / 8, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, in, trips, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 32.0 * 2048.0 * (double)trips * (threads / 64) * blocks;
    return flop / (ms * 1e-3) / 1e12;
}

int main() {
    double *in, *out;
    const size_t n = 1 << 20;
    double* h = (double*)malloc(n * sizeof(double));
    srand(7);
    for (size_t i = 0; i < n; ++i) h[i] = (double)rand() / RAND_MAX + 0.01;
    CK(hipMalloc(&in, n * sizeof(double))); CK(hipMalloc(&out, 8));
    CK(hipMemcpy(in, h, n * sizeof(double), hipMemcpyHostToDevice));
    for (int w = 1; w <= 2; ++w) {
        const long trips = 1200000 / w;          // ~1 s at the full rate
        printf("{\"waves_per_simd\": %d, \"independent\": %.1f, \"chain16\": %.1f, \"chains2x16\": %.1f, \"indep_plus_div\": %.1f, \"kl_tile\": %.1f, \"unit\": \"TFLOP/s\", \"peak\": 78.6}\n",
               w, run<0>(in, out, w, trips), run<1>(in, out, w, trips), run<2>(in, out, w, trips), run<3>(in, out, w, trips), run<4>(in, out, w, trips));
    }
    return 0;
}
