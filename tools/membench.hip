// membench.hip -- HBM access-pattern microbenchmark for the two streaming patterns of the MU kernels (no MFMA).
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/membench tools/membench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// P0: linear stream
__global__ void p_linear(const f32x4* A, long n4, float* out) {
    f32x4 s = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) s += A[i];
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1;
}

// P1: TN pattern. wave -> (chunk, colblk of CW floats); per step the wave loads D row-pairs (lane (li,h): row r+2u+h, 16 B at col0 + 4 li)
template <int D>
__global__ void p_tn(const float* A, long m, long ld, int ncolblk, long rows_per_chunk, float* out) {
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const long gw = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long chunk = gw / ncolblk, colblk = gw % ncolblk;
    const long rbeg = chunk * rows_per_chunk, rend = rbeg + rows_per_chunk;
    if (rend > m) return;
    const float* base = A + colblk * 128 + 4 * li + h * ld;
    f32x4 s = {0, 0, 0, 0};
    for (long r = rbeg; r < rend; r += 2 * D) {
        f32x4 v[D];
#pragma unroll
        for (int u = 0; u < D; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + (r + 2 * u) * ld);
#pragma unroll
        for (int u = 0; u < D; ++u) s += v[u];
    }
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1;
}

// P2: NT pattern. WG of 256 threads owns BM rows; per k-tile (32 floats = 128 B per row) thread t loads rows (t>>3)+32*it, chunk t&7
template <int BM, int D>   // D k-tiles in flight
__global__ void p_nt(const float* A, long m, long n, long ld, float* out) {
    const int t = threadIdx.x;
    const long row0 = (long)blockIdx.x * BM;
    const float* base = A + (row0 + (t >> 3)) * ld + (t & 7) * 4;
    f32x4 s = {0, 0, 0, 0};
    for (long c = 0; c < n; c += 32 * D) {
        f32x4 v[D][BM / 32];
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int it = 0; it < BM / 32; ++it) v[d][it] = *reinterpret_cast<const f32x4*>(base + (long)it * 32 * ld + c + 32 * d);
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int it = 0; it < BM / 32; ++it) s += v[d][it];
    }
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1;
}

// P3: row-slab pattern: wave reads 1 row x 1 KiB per instruction (lane i: 16 B at col0 + 4 i), D rows in flight, walks down rows
template <int D>
__global__ void p_rows1k(const float* A, long m, long ld, int ncolblk, long rows_per_chunk, float* out) {
    const int lane = threadIdx.x & 63;
    const long gw = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long chunk = gw / ncolblk, colblk = gw % ncolblk;
    const long rbeg = chunk * rows_per_chunk, rend = rbeg + rows_per_chunk;
    if (rend > m) return;
    const float* base = A + colblk * 256 + 4 * lane;
    f32x4 s = {0, 0, 0, 0};
    for (long r = rbeg; r < rend; r += D) {
        f32x4 v[D];
#pragma unroll
        for (int u = 0; u < D; ++u) v[u] = *reinterpret_cast<const f32x4*>(base + (r + u) * ld);
#pragma unroll
        for (int u = 0; u < D; ++u) s += v[u];
    }
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1;
}

template <typename F>
double timeit(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}

int main(int argc, char** argv) {
    long m = argc > 1 ? atol(argv[1]) : 262144, n = argc > 2 ? atol(argv[2]) : 8192;
    long ld = argc > 3 ? atol(argv[3]) : n;
    float *A, *out; CK(hipMalloc(&A, m * ld * 4)); CK(hipMalloc(&out, 4));
    CK(hipMemset(A, 0, m * ld * 4));
    const double gb = (double)m * n * 4 / 1e9;
    printf("m=%ld n=%ld ld=%ld (%.2f GB)\n", m, n, ld, gb);
    double ms;
    ms = timeit([&] { hipLaunchKernelGGL(p_linear, dim3(2048), dim3(256), 0, 0, (const f32x4*)A, m * n / 4, out); });
    printf("P0 linear                 : %7.3f ms %7.1f GB/s\n", ms, gb / ms * 1e3);
#define RUN_TN(D, RPC) { int ncb = n / 128; long waves = (m / RPC) * ncb; ms = timeit([&] { hipLaunchKernelGGL(p_tn<D>, dim3(waves / 4), dim3(256), 0, 0, A, m, ld, ncb, (long)RPC, out); }); \
    printf("P1 tn   D=%2d rows/chunk=%5d waves=%6ld: %7.3f ms %7.1f GB/s\n", D, RPC, waves, ms, gb / ms * 1e3); }
    RUN_TN(2, 4096) RUN_TN(4, 4096) RUN_TN(8, 4096) RUN_TN(16, 4096) RUN_TN(4, 1024) RUN_TN(8, 1024) RUN_TN(8, 16384)
#define RUN_NT(BM, D) { ms = timeit([&] { hipLaunchKernelGGL((p_nt<BM, D>), dim3(m / BM), dim3(256), 0, 0, A, m, n, ld, out); }); \
    printf("P2 nt   BM=%3d ktiles_in_flight=%d          : %7.3f ms %7.1f GB/s\n", BM, D, ms, gb / ms * 1e3); }
    RUN_NT(128, 1) RUN_NT(128, 2) RUN_NT(128, 4) RUN_NT(256, 1) RUN_NT(256, 2) RUN_NT(64, 2) RUN_NT(64, 4)
#define RUN_R1(D, RPC) { int ncb = n / 256; long waves = (m / RPC) * ncb; ms = timeit([&] { hipLaunchKernelGGL(p_rows1k<D>, dim3(waves / 4), dim3(256), 0, 0, A, m, ld, ncb, (long)RPC, out); }); \
    printf("P3 1KiB rows D=%2d rows/chunk=%5d waves=%6ld: %7.3f ms %7.1f GB/s\n", D, RPC, waves, ms, gb / ms * 1e3); }
    RUN_R1(4, 2048) RUN_R1(8, 2048) RUN_R1(16, 2048) RUN_R1(8, 512)
    return 0;
}
