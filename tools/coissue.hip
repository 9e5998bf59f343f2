// coissue.hip -- do fp32 VALU instructions overlap with fp32 MFMAs on one SIMD?  (MI355X_MICROARCH.md quantifies the overlap
// for the bf16 MFMA shapes; the fp32-input MFMA runs at the fp32 VALU rate and the KL kernels put a division between two
// fp32 products, so the answer decides what their ceiling is.)
// Each trip: M independent v_mfma_f32_16x16x4_f32 (4 accumulator chains) and V independent VALU ops (v_fma_f32, or v_rcp_f32
// for every fourth one with RCP = 1); W waves per SIMD; cycles per trip come from s_memtime.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/coissue tools/coissue.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int M, int V, int RCP, int SHAPE>
__global__ __launch_bounds__(256) void k(long trips, float* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    float a = 1.0f + lane * 1e-6f, b = 0.5f;
    f32x4 acc[4];
    f32x16 acc32[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    float x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = 1.0f + j + lane * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (long t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < (M > V ? M : V); ++i) {
            if (i < M) {
                if constexpr (SHAPE == 16) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 3], 0, 0, 0);
                else acc32[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32[i & 1], 0, 0, 0);
            }
            if (i < V) {
                if (RCP && (i & 3) == 3) x[i & 15] = __builtin_amdgcn_rcpf(x[i & 15]);
                else x[i & 15] = fmaf(x[i & 15], 0.999f, 0.001f);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc32[i][r];
#pragma unroll
    for (int j = 0; j < 16; ++j) s += x[j];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// The same question for LDS reads (ds_read_b128, conflict free) and for global loads that hit the L2 (16 B per lane from a
// 64 KiB buffer): M MFMAs (32x32x2) + NDS LDS reads + NGL global loads per trip, the loaded values folded into a register that
// is consumed after the loop (the loads complete, nothing waits on them inside the trip beyond hipcc's own counters).
template <int M, int NDS, int NGL>
__global__ __launch_bounds__(256) void k2(long trips, const f32x4* __restrict__ gbuf, float* out) {
    __shared__ f32x4 lds[1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = f32x4{1.f * i, 2.f, 3.f, 4.f};
    __syncthreads();
    float a = 1.0f + lane * 1e-6f, b = 0.5f;
    f32x16 acc32[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
    f32x4 sink = {0.f, 0.f, 0.f, 0.f};
    int off = threadIdx.x;
    for (long t = 0; t < trips; ++t) {
        f32x4 dv[NDS > 0 ? NDS : 1], gv[NGL > 0 ? NGL : 1];
#pragma unroll
        for (int i = 0; i < NDS; ++i) dv[i] = lds[(off + 64 * i) & 1023];
#pragma unroll
        for (int i = 0; i < NGL; ++i) gv[i] = gbuf[(off + 256 * i) & 4095];
#pragma unroll
        for (int i = 0; i < M; ++i) acc32[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32[i & 1], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NDS; ++i) sink[i & 3] += dv[i][i & 3];
#pragma unroll
        for (int i = 0; i < NGL; ++i) sink[i & 3] += gv[i][i & 3];
        off = (off + 7) & 1023;
    }
    float s = sink[0] + sink[1] + sink[2] + sink[3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc32[i][r];
    if (s == 123.456f) out[0] = s;
}

template <int M, int NDS, int NGL>
void run2(int wps, const f32x4* gbuf, float* out) {
    const long trips = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wps;
    hipLaunchKernelGGL((k2<M, NDS, NGL>), dim3(grid), dim3(256), 0, 0, trips / 10, gbuf, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k2<M, NDS, NGL>), dim3(grid), dim3(256), 0, 0, trips, gbuf, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("32x32x2  M=%2d ds_read_b128=%2d global_load_dwordx4=%2d waves/SIMD=%d : %8.3f ms (%6.1f ns/trip; each trip also has the %d vector adds that consume the loads)\n",
           M, NDS, NGL, wps, ms, ms * 1e6 / trips, NDS + NGL);
}


// What clock does the chip hold under fp32 MFMAs alone?  k3: M v_mfma_f32_32x32x2_f32 per trip on operands that are random per
// lane and change from MFMA to MFMA (RND = 1) or constants (RND = 0), 4 waves per SIMD; the clock = s_memtime ticks per 10 ns of
// wall_clock64 (block 0).  The NT / TN kernels of the library hold 2.0 GHz at 93-97 % MFMA busy on random data.
template <int RND>
__global__ __launch_bounds__(256) void k3(long trips, float* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    float a[8], b[8];
    unsigned x = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        a[i] = RND ? (x >> 8) * (1.0f / 16777216.0f) + 1e-3f : 1.0f;
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        b[i] = RND ? (x >> 8) * (1.0f / 16777216.0f) + 1e-3f : 0.5f;
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (long t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 7], b[(i + (i >> 3)) & 7], acc[i & 3], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
    (void)lane;
}

template <int RND>
void run3(float* out, unsigned long long* cyc) {
    const long trips = 200000;                                 // 16 x 64 cycles x 4 waves per SIMD = 4096 cycles per trip: ~0.4 s
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k3<RND>), dim3(1024), dim3(256), 0, 0, trips / 10, out, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k3<RND>), dim3(1024), dim3(256), 0, 0, trips, out, cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c[2]; CK(hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost));
    printf("fp32 MFMA 32x32x2 only, %s operands, 4 waves/SIMD: %8.2f ms, %.1f cycles per MFMA and SIMD, clock held %.3f GHz (%.1f TFLOP/s)\n",
           RND ? "random" : "constant", ms, (double)c[0] / trips / 16 / 4 * 4, (double)c[0] / (double)c[1] * 0.1,
           1024.0 * 4 * trips * 16 * 4096 * 2 / (ms * 1e-3) * 1e-12);
}

template <int M, int V, int RCP, int SHAPE>
void run(int wps, float* out, unsigned long long* cyc) {
    const long trips = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wps;      // 4-wave workgroups: wps workgroups per CU = wps waves per SIMD
    hipLaunchKernelGGL((k<M, V, RCP, SHAPE>), dim3(grid), dim3(256), 0, 0, trips / 10, out, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<M, V, RCP, SHAPE>), dim3(grid), dim3(256), 0, 0, trips, out, cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    // s_memtime counts at 100 MHz steps of the shader clock? (MI355X_MICROARCH.md: tick = shader cycle)
    printf("shape %2d  M=%2d V=%2d rcp=%d waves/SIMD=%d : %8.3f ms  %7.1f memtime ticks/trip  (%5.1f ns/trip)  MFMA-only floor %d cycles\n",
           SHAPE, M, V, RCP, wps, ms, (double)c / trips, ms * 1e6 / trips, M * (SHAPE == 16 ? 32 : 64) * wps);
}

int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 64));
    if (getenv("COISSUE_CLOCK_ONLY")) { run3<0>(out, cyc); run3<1>(out, cyc); run3<0>(out, cyc); run3<1>(out, cyc); return 0; }
    for (int wps = 1; wps <= 4; wps += (wps == 1 ? 1 : 2)) {     // 1, 2, 4
        run<8, 0, 0, 16>(wps, out, cyc);
        run<8, 8, 0, 16>(wps, out, cyc);
        run<8, 16, 0, 16>(wps, out, cyc);
        run<8, 32, 0, 16>(wps, out, cyc);
        run<8, 64, 0, 16>(wps, out, cyc);
        run<8, 32, 1, 16>(wps, out, cyc);
        run<0, 32, 0, 16>(wps, out, cyc);
        run<0, 64, 0, 16>(wps, out, cyc);
        run<4, 0, 0, 32>(wps, out, cyc);
        run<4, 32, 0, 32>(wps, out, cyc);
        run<4, 64, 0, 32>(wps, out, cyc);
    }
    f32x4* gbuf; CK(hipMalloc(&gbuf, 4096 * sizeof(f32x4))); CK(hipMemset(gbuf, 0, 4096 * sizeof(f32x4)));
    for (int wps = 1; wps <= 4; wps *= 2) {
        run2<8, 0, 0>(wps, gbuf, out);
        run2<8, 8, 0>(wps, gbuf, out);
        run2<8, 16, 0>(wps, gbuf, out);
        run2<8, 0, 4>(wps, gbuf, out);
        run2<8, 0, 8>(wps, gbuf, out);
        run2<0, 16, 0>(wps, gbuf, out);
        run2<0, 0, 8>(wps, gbuf, out);
    }
    return 0;
}
