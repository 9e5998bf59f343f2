// shapeclock.hip -- does the fp32 MFMA SHAPE change the clock the chip holds under load?  (MI355X_MICROARCH.md "DVFS give-back" item 7
// measured it for the bf16 shapes: 16x16x32 delivered 1.15 x the FLOP/s of 32x32x16 at equal cycles per FLOP, on random data.)
// Bare loops on random operands held in registers, one or two waves per SIMD, every CU busy, >= 1 s per measurement (wall clock by HIP
// events): TFLOP/s of v_mfma_f32_32x32x2_f32 against v_mfma_f32_16x16x4_f32.  Both retire 64 FLOP per cycle and SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/_build/shapeclock tools/shapeclock.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int SHAPE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ in, long trips, float* out) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = in[(tid * 16 + j) & 0xfffff]; b[j] = in[(tid * 16 + 8 + j) & 0xfffff]; }
    f32x4 acc[8];
    f32x16 acc32[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    for (long t = 0; t < trips; ++t) {
        if constexpr (SHAPE == 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 7], b[(i + 3) & 7], acc[i & 7], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc32[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 7], b[(i + 3) & 7], acc32[i & 3], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc32[i][r];
    if (s == 123.456f) out[0] = s;
}

template <int SHAPE>
double run(const float* in, float* out, int waves_per_simd, long trips) {
    const int threads = 256 * waves_per_simd, blocks = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(threads), 0, 0, in, trips / 8, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(threads), 0, 0, in, trips, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // per trip and wave: 16 x 2048 flop (16x16x4) or 8 x 4096 flop (32x32x2) = 32768 flop
    const double flop = 32768.0 * (double)trips * (threads / 64) * blocks;
    return flop / (ms * 1e-3) / 1e12;
}

int main() {
    float *in, *out;
    const size_t n = 1 << 20;
    float* h = (float*)malloc(n * sizeof(float));
    srand(7);
    for (size_t i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX + 0.01f;
    CK(hipMalloc(&in, n * sizeof(float))); CK(hipMalloc(&out, 4));
    CK(hipMemcpy(in, h, n * sizeof(float), hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep)
        for (int w = 1; w <= 2; ++w) {
            const long trips = 6000000 / w;          // ~1.5 s at the full rate
            const double t32 = run<32>(in, out, w, trips);
            const double t16 = run<16>(in, out, w, trips);
            printf("{\"waves_per_simd\": %d, \"tflops_32x32x2\": %.1f, \"tflops_16x16x4\": %.1f, \"ratio_16_over_32\": %.3f}\n", w, t32, t16, t16 / t32);
        }
    return 0;
}
