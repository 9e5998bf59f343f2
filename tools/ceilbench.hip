// ceilbench.hip -- measured ceilings of one MI355X for the MU kernels' instruction mix ("record the measured
// ceilings next to the datasheet ones"):
//   mfma      : v_mfma_f32_32x32x2_f32 issue rate with nothing else going on (8 waves/CU, 4 independent accumulators each)
//   stream    : nontemporal 16-B loads of a large buffer, no MFMA
//   mfma+hbm  : the same MFMA loop with every wave also streaming R bytes of HBM per MFMA -- the power-shared regime
//               the A.H^T / W^T.A kernels run in (k = 64: 32 B of A per 32x32x2 MFMA of 4096 flop)
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ceilbench tools/ceilbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// Each trip: LOADS 16-byte nontemporal loads (64 lanes x 16 B = 1 KiB per instruction) and MFMAS MFMAs.
template <int MFMAS, int LOADS>
__global__ __launch_bounds__(256) void mix_kernel(const f32x4* __restrict__ buf, long n4, long trips, float* out) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long gw = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    float a = 1.0f + lane * 1e-6f, b = 0.5f;
    f32x4 v[LOADS > 0 ? LOADS : 1];
    long idx = (gw * 64 + lane) % n4;
    const long stride = nw * 64;
    if constexpr (LOADS > 0) {
#pragma unroll
        for (int l = 0; l < LOADS; ++l) { v[l] = __builtin_nontemporal_load(buf + idx); idx += stride; if (idx >= n4) idx -= n4; }
    }
    for (long t = 0; t < trips; ++t) {
        f32x4 nv[LOADS > 0 ? LOADS : 1];
        if constexpr (LOADS > 0) {
#pragma unroll
            for (int l = 0; l < LOADS; ++l) { nv[l] = __builtin_nontemporal_load(buf + idx); idx += stride; if (idx >= n4) idx -= n4; }
        }
#pragma unroll
        for (int i = 0; i < MFMAS; ++i) {
            float aa = a;
            if constexpr (LOADS > 0) aa += v[i % LOADS][i & 3] * 1e-30f;   // consume the loaded data
            acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa, b, acc[i & 3], 0, 0, 0);
        }
        if constexpr (LOADS > 0) {
            if constexpr (MFMAS == 0) {
#pragma unroll
                for (int l = 0; l < LOADS; ++l) a += v[l][0] + v[l][1] + v[l][2] + v[l][3];
            }
#pragma unroll
            for (int l = 0; l < LOADS; ++l) v[l] = nv[l];
        }
    }
    float s = a;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}

// the same mix on v_mfma_f32_16x16x4_f32 (2048 flop per instruction, 4 accumulator registers: per flop 40 % less register-file
// traffic than 32x32x2 -- does the power-shared ceiling move?  round 4)
template <int MFMAS, int LOADS>
__global__ __launch_bounds__(256) void mix16_kernel(const f32x4* __restrict__ buf, long n4, long trips, float* out) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long gw = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nw = ((long)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    float a = 1.0f + lane * 1e-6f, b = 0.5f;
    f32x4 v[LOADS > 0 ? LOADS : 1];
    long idx = (gw * 64 + lane) % n4;
    const long stride = nw * 64;
    if constexpr (LOADS > 0) {
#pragma unroll
        for (int l = 0; l < LOADS; ++l) { v[l] = __builtin_nontemporal_load(buf + idx); idx += stride; if (idx >= n4) idx -= n4; }
    }
    for (long t = 0; t < trips; ++t) {
        f32x4 nv[LOADS > 0 ? LOADS : 1];
        if constexpr (LOADS > 0) {
#pragma unroll
            for (int l = 0; l < LOADS; ++l) { nv[l] = __builtin_nontemporal_load(buf + idx); idx += stride; if (idx >= n4) idx -= n4; }
        }
#pragma unroll
        for (int i = 0; i < MFMAS; ++i) {
            float aa = a;
            if constexpr (LOADS > 0) aa += v[i % LOADS][i & 3] * 1e-30f;
            acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(aa, b, acc[i & 7], 0, 0, 0);
        }
        if constexpr (LOADS > 0) {
#pragma unroll
            for (int l = 0; l < LOADS; ++l) v[l] = nv[l];
        }
    }
    float s = a;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
}

template <int MFMAS, int LOADS>
void run16(const char* name, const f32x4* buf, long n4, float* out, long trips, int wg_per_cu = 2) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wg_per_cu, block = 256;
    trips = trips * 2 / wg_per_cu;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((mix16_kernel<MFMAS, LOADS>), dim3(grid), dim3(block), 0, 0, buf, n4, trips / 8, out);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((mix16_kernel<MFMAS, LOADS>), dim3(grid), dim3(block), 0, 0, buf, n4, trips, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double waves = (double)grid * block / 64;
    const double flops = waves * trips * MFMAS * 2048.0, bytes = waves * trips * LOADS * 1024.0;
    printf("%-30s %2d waves/CU %8.3f ms  %7.1f TFLOP/s (%3.0f %% of 157.3)  %6.2f TB/s  [16x16x4: %d B of HBM per 4096 flop]\n", name, 4 * wg_per_cu, best,
           flops / best / 1e9, 100.0 * flops / best / 1e9 / 157.3, bytes / best / 1e9, MFMAS ? LOADS * 2048 / MFMAS : 0);
}

template <int MFMAS, int LOADS>
void run(const char* name, const f32x4* buf, long n4, float* out, long trips, int wg_per_cu = 2) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wg_per_cu, block = 256;   // 4 waves per workgroup
    trips = trips * 2 / wg_per_cu;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((mix_kernel<MFMAS, LOADS>), dim3(grid), dim3(block), 0, 0, buf, n4, trips / 8, out);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((mix_kernel<MFMAS, LOADS>), dim3(grid), dim3(block), 0, 0, buf, n4, trips, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double waves = (double)grid * block / 64;
    const double flops = waves * trips * MFMAS * 4096.0, bytes = waves * trips * LOADS * 1024.0;
    printf("%-30s %2d waves/CU %8.3f ms  %7.1f TFLOP/s (%3.0f %% of 157.3)  %6.2f TB/s  [%d B of HBM per MFMA]\n", name, 4 * wg_per_cu, best,
           flops / best / 1e9, 100.0 * flops / best / 1e9 / 157.3, bytes / best / 1e9, MFMAS ? LOADS * 1024 / MFMAS : 0);
}

int main() {
    const long bytes = 8L << 30;
    f32x4* buf; float* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(buf, 0, bytes));
    const long n4 = bytes / 16;
    run<32, 0>("mfma only", buf, n4, out, 20000);
    run<0, 4>("stream only", buf, n4, out, 4000);
    run<0, 4>("stream only", buf, n4, out, 4000, 4);
    for (int w = 2; w <= 4; ++w) {
        run<32, 2>("mfma + stream (k=128 ratio)", buf, n4, out, 20000, w);
        run<32, 4>("mfma + stream (k=64 ratio)", buf, n4, out, 20000, w);
        run<16, 4>("mfma + stream (k=32 ratio)", buf, n4, out, 20000, w);
        run<8, 4>("mfma + stream (k=16 ratio)", buf, n4, out, 20000, w);
    }
    // round 4: the same ceilings on the 16x16x4 instruction (twice the instructions for the same flops and bytes)
    run16<64, 0>("16x16x4 mfma only", buf, n4, out, 20000);
    for (int w = 2; w <= 3; ++w) {
        run16<64, 2>("16x16x4 + stream (k=128)", buf, n4, out, 20000, w);
        run16<64, 4>("16x16x4 + stream (k=64)", buf, n4, out, 20000, w);
        run16<32, 4>("16x16x4 + stream (k=32)", buf, n4, out, 20000, w);
    }
    return 0;
}
