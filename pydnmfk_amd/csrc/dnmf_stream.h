// dnmf_stream.h -- streaming kernels: clamp, scale, KL element-wise update, norms, row / column sums.
// Part of libdnmf_hip.so (single translation unit: csrc/dnmf.hip includes every header once).
#pragma once
#include "dnmf_common.h"

namespace {

// =============================================================================================== small kernels
__global__ __launch_bounds__(256) void clamp_kernel(float* X, long rows, long cols, long ldx, float eps) {
    const long total = rows * cols;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / cols, c = idx % cols;
        float* p = X + r * ldx + c;
        *p = fmaxf(*p, eps);
    }
}

// W[i][j] = W[i][j] / (s[j] + eps)   |   H[j][c] = H[j][c] * s[j]
template <int OP>
__global__ __launch_bounds__(256) void scale_kernel(float* X, long rows, long cols, long ldx, const float* s, float eps) {
    const long total = rows * cols;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / cols, c = idx % cols;
        float* p = X + r * ldx + c;
        if (OP == 0) *p = *p / (s[c] + eps);
        else *p = *p * s[r];
    }
}

// KL eltwise: X[r][c] *= S[r][c] / (x[BYROW ? r : c] + eps)   (dist_nmf.py:828-830, 847-849)
template <bool BYROW>
__global__ __launch_bounds__(256) void kl_update_kernel(float* X, long rows, long cols, long ldx, const float* S,
                                                        long lds_, const float* x, float eps, int clamp) {
    const long total = rows * cols;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / cols, c = idx % cols;
        const float q = S[r * lds_ + c] / (x[BYROW ? r : c] + eps);
        float v = X[r * ldx + c] * q;
        if (clamp) v = fmaxf(v, eps);
        X[r * ldx + c] = v;
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ void block_atomic_sum(double v, double* out) {
    __shared__ double red[16];
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
        atomicAdd(out, s);
    }
}

// sum of squares of an m x n matrix; fp32 products, fp64 accumulation
template <bool FAST, typename TA = float>
__global__ __launch_bounds__(256) void sqnorm_kernel(const TA* __restrict__ A, long m, long n, long lda, double* out) {
    double acc = 0.0;
    if constexpr (FAST) {
        // four independent nontemporal 4-element loads in flight per lane and trip; a matrix without row padding is
        // walked as one flat array (no 64-bit division per element).  Measured: 5.5 -> see DESIGN.md (stream ceiling 6.8 TB/s)
        const long n4 = n / 4, total = m * n4, stride = (long)gridDim.x * blockDim.x;
        const bool flat = lda == n;
        auto off = [&](long i) { return flat ? i * 4 : (i / n4) * lda + (i % n4) * 4; };
        long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
        for (; idx + 3 * stride < total; idx += 4 * stride) {
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) load_vec_raw_nt<4>(v[u], A + off(idx + u * stride));
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += (double)(v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3]);
        }
        for (; idx < total; idx += stride) {
            float v[4];
            load_vec_raw<4>(v, A + off(idx));
            acc += (double)(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
        }
    } else {
        const long total = m * n;
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
            float v[1];
            load_vec_raw<1>(v, A + (idx / n) * lda + idx % n);
            acc += (double)(v[0] * v[0]);
        }
    }
    block_atomic_sum(acc, out);
}

// x[j] = sum_c H[j][c]  -- one workgroup per row
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ H, long n, long ldh, float* x) {
    const float* row = H + (long)blockIdx.x * ldh;
    double acc = 0.0;
    for (long c = threadIdx.x; c < n; c += blockDim.x) acc += (double)row[c];
    __shared__ double red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) x[blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}

// stage 1 of x[j] = sum_i W[i][j]: partial[blk][j] over a slab of rows (coalesced along j)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ W, long m, int k, long ldw,
                                                             long rows_per_blk, float* partial, int kp) {
    __shared__ float red[256];
    const int j = threadIdx.x % kp, g = threadIdx.x / kp, ng = 256 / kp;
    const long r0 = (long)blockIdx.x * rows_per_blk;
    long r1 = r0 + rows_per_blk;
    if (r1 > m) r1 = m;
    float acc = 0.f;
    if (j < k)
        for (long r = r0 + g; r < r1; r += ng) acc += W[r * ldw + j];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0) {
        for (int q = 1; q < ng; ++q) acc += red[q * kp + j];
        partial[(long)blockIdx.x * kp + j] = acc;
    }
}

__global__ void colsum_final_kernel(const float* partial, int nblk, int kp, int k, float* x) {
    const int j = threadIdx.x;
    if (j >= k) return;
    double acc = 0.0;
    for (int b = 0; b < nblk; ++b) acc += (double)partial[(long)b * kp + j];
    x[j] = (float)acc;
}


}  // namespace
