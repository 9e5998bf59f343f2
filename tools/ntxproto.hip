// ntxproto.hip -- prototype bench for the bf16x6 A H^T kernel (ntx_kernel of csrc/dnmf_split.h) against candidate main loops.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -Ipydnmfk_amd/csrc -Itools -o tools/_build/ntxproto tools/ntxproto.hip
// Run on the GPU box: tools/_build/ntxproto [m n]   (k = 64; outputs are compared bit for bit with the library kernel)
#define NT2_CLOCKS 1
#define DNMF_NTX2 1      // pulls tools/dnmf_split_nt2.h into dnmf_split.h (not part of the library any more)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cmath>
#include "dnmf.h"
#include "dnmf_host.h"
#include "dnmf_split.h"

namespace {
template <int KT, int MODE, int AUX, int NSET, int ABL = 0>
__global__ __launch_bounds__(256, 2) void ntx2_kernel(NtArgs p, SplitOperand ys) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long row0 = (long)blockIdx.x * 128;
    f32x16 acc[1][KT];
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][jt][r] = 0.f;
#ifdef NT2_CLOCKS
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
#endif
    ntx2_mainloop<KT, NSET, AUX != 0, ABL>(acc, static_cast<const float*>(p.X), p.ldx, row0, ys, 0, p.ncols, smem);
    float* out = p.out;
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = row0 + wave * 32 + crow(r, h);
            out[row * p.ldo + jt * 32 + li] = acc[0][jt][r];
        }
#ifdef NT2_CLOCKS      // shader clock held over the main loop (tools/ntxproto.hip): s_memtime ticks per 100 MHz wall tick
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) {
        unsigned long long* c = reinterpret_cast<unsigned long long*>(const_cast<float*>(p.G)) + 2 * (blockIdx.x >> 6);
        c[0] = __builtin_amdgcn_s_memtime() - c0;
        c[1] = wall_clock64() - w0;
    }
#endif
}

}  // namespace

char* dnmf_errbuf_() { static char b[DNMF_ERRBUF]; return b; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, long n, unsigned seed) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i < n; i += stride) {
        unsigned x = (unsigned)(i * 2654435761u) ^ seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (x >> 8) * (1.0f / 16777216.0f) + 1e-3f;
    }
}

template <typename F>
float time_ms(F&& f, int reps = 7) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv) {
    const long m = argc > 2 ? atol(argv[1]) : 262144, n = argc > 2 ? atol(argv[2]) : 8192;
    const int k = 64, KT = 2, kp = 64;
    float *X, *H, *o0, *o1;
    CK(hipMalloc(&X, m * n * 4)); CK(hipMalloc(&H, (long)k * n * 4)); CK(hipMalloc(&o0, m * kp * 4)); CK(hipMalloc(&o1, m * kp * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, X, m * n, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, H, (long)k * n, 2u);
    bf16_t* img; CK(hipMalloc(&img, 3L * kp * n * 2));
    SplitOperand ys{img, (long)kp * n, n};
    hipLaunchKernelGGL(split3_rows_kernel, dim3((unsigned)((kp * (n / 8) + 255) / 256)), dim3(256), 0, 0, H, n, k, n, img, n, ys.split_stride, kp);
    CK(hipDeviceSynchronize());
    NtArgs a{};
    a.X = X; a.ldx = n; a.nrows = m; a.ncols = n; a.yrows = k; a.cols_per_split = n; a.ldo = kp; a.split_stride = 0; a.store_all = 1; a.k = k;
    const dim3 grid((unsigned)(m / 128));
    unsigned long long* clk; CK(hipMalloc(&clk, 16 * (m / 128 / 64 + 1)));
    a.G = reinterpret_cast<const float*>(clk);
    constexpr size_t lds0 = 2 * (32 * 4 * 128 + 3 * 32 * KT * 64);
    allow_lds(ntx_kernel<KT, NT_STORE, 2, float, 4, 4, 0>, lds0);
    auto ref = [&] { NtArgs b = a; b.out = o0; hipLaunchKernelGGL((ntx_kernel<KT, NT_STORE, 2, float, 4, 4, 0>), grid, dim3(256), lds0, 0, b, ys); };
    std::vector<float> h0(m * kp), h1(m * kp);
    ref();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h0.data(), o0, m * kp * 4, hipMemcpyDeviceToHost));
    printf("ntx_kernel (library)         : %.3f ms\n", time_ms(ref));

    auto run = [&](const char* name, auto kernel, size_t lds) {
        allow_lds(kernel, lds);
        CK(hipMemset(o1, 0xff, m * kp * 4));
        auto f = [&] { NtArgs b = a; b.out = o1; hipLaunchKernelGGL(kernel, grid, dim3(256), lds, 0, b, ys); };
        f();
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        CK(hipMemcpy(h1.data(), o1, m * kp * 4, hipMemcpyDeviceToHost));
        long bad = 0; double maxrel = 0;
        for (long i = 0; i < m * kp; ++i) {
            if (memcmp(&h0[i], &h1[i], 4) != 0) { ++bad; const double r = fabs((double)h0[i] - h1[i]) / fabs((double)h0[i]); if (!(r <= maxrel)) maxrel = r; }
        }
        const float ms = time_ms(f);
        std::vector<unsigned long long> hc(2 * (m / 128 / 64));
        CK(hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost));
        double ghz = 0;
        for (size_t i = 0; i < hc.size(); i += 2) ghz += (double)hc[i] / (double)hc[i + 1] * 0.1;
        printf("%-30s: %.3f ms   differing words %ld (max rel %.3g)   memtime/wall = %.3f GHz\n", name, ms, bad, maxrel, ghz / (hc.size() / 2));
    };
    run("ntx_kernel, second main loop", ntx_kernel<KT, NT_STORE, 2, float, 4, 4, 1>, lds0);
    run("nt2 NSET=4", ntx2_kernel<KT, NT_STORE, 2, 4>, nt2_lds_bytes<KT>());
    run("nt2 NSET=2", ntx2_kernel<KT, NT_STORE, 2, 2>, nt2_lds_bytes<KT>());
    run("nt2 NSET=2 half the H reads", ntx2_kernel<KT, NT_STORE, 2, 2, 4>, nt2_lds_bytes<KT>());
    run("nt2 NSET=2 no MFMA", ntx2_kernel<KT, NT_STORE, 2, 2, 1>, nt2_lds_bytes<KT>());
    run("nt2 NSET=2 no cut", ntx2_kernel<KT, NT_STORE, 2, 2, 2>, nt2_lds_bytes<KT>());
    run("nt2 NSET=2 no MFMA, no cut", ntx2_kernel<KT, NT_STORE, 2, 2, 3>, nt2_lds_bytes<KT>());
    run("nt2 NSET=4 no MFMA", ntx2_kernel<KT, NT_STORE, 2, 4, 1>, nt2_lds_bytes<KT>());
    run("nt2 NSET=4 no MFMA, no cut", ntx2_kernel<KT, NT_STORE, 2, 4, 3>, nt2_lds_bytes<KT>());
    return 0;
}
