// dnmf_hals.h -- HALS / Frobenius column and row sweeps.
// Part of libdnmf_hip.so (single translation unit: csrc/dnmf.hip includes every header once).
#pragma once
#include "dnmf_common.h"

namespace {

// =============================================================================================== HALS sweeps
// Frobenius HALS (dist_nmf.py:873-934, :411-470) reuses the MU contractions (A H^T, H H^T, W^T A, W^T W) and replaces
// the multiply-divide by a column-sequential sweep.
//
// W sweep, one launch per column kk (the global 2-norm of column kk must be known before column kk+1 is touched;
// with p_r > 1 the host allreduces the 8-byte sum of squares between launches, exactly where the reference calls
// utils.norm, dist_nmf.py:889):
//   first the pending normalisation of column kk-1 is applied (W[i][kk-1] /= ss, ss = sqrt(*prev_ss2), skipped when 0),
//   t = W[i][kk] * G[kk][kk] + AH[i][kk] - sum_j W[i][j] G[j][kk];  W[i][kk] = max(t, eps);  *ss2_out += W[i][kk]^2
// One lane per row; a row of W is k contiguous floats.
__global__ __launch_bounds__(256) void hals_w_col_kernel(float* __restrict__ W, long m, int k, long ldw,
                                                        const float* __restrict__ AH, long ldah,
                                                        const float* __restrict__ G, int kp, int kk,
                                                        const double* __restrict__ prev_ss2, float eps,
                                                        double* __restrict__ ss2_out) {
    __shared__ float gcol[DNMF_MAX_K];
    for (int j = threadIdx.x; j < k; j += blockDim.x) gcol[j] = G[(long)j * kp + kk];
    __syncthreads();
    float inv_den = 0.f;   // ss of the previous column (0 = no pending normalisation)
    if (kk > 0 && prev_ss2) inv_den = (float)sqrt(*prev_ss2);
    double sq = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long)gridDim.x * blockDim.x) {
        float* row = W + i * ldw;
        if (kk > 0 && inv_den > 0.f) row[kk - 1] = row[kk - 1] / inv_den;
        float dot = 0.f;
        for (int j = 0; j < k; ++j) dot = fmaf(row[j], gcol[j], dot);
        const float t = row[kk] * gcol[kk] + AH[i * ldah + kk] - dot;
        const float w = fmaxf(t, eps);
        row[kk] = w;
        sq += (double)w * (double)w;
    }
    block_atomic_sum(sq, ss2_out);
}

// final normalisation of one column: W[i][col] /= sqrt(*ss2) (skipped when 0)
__global__ __launch_bounds__(256) void hals_w_scale_kernel(float* __restrict__ W, long m, long ldw, int col,
                                                          const double* __restrict__ ss2) {
    const float ss = (float)sqrt(*ss2);
    if (!(ss > 0.f)) return;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long)gridDim.x * blockDim.x)
        W[i * ldw + col] = W[i * ldw + col] / ss;
}

// H sweep: columns are independent, rows are sequential (row kk uses the already updated rows < kk):
//   H[kk][c] = max(H[kk][c] + AtW[kk][c] - sum_j G[kk][j] H[j][c], eps)            (dist_nmf.py:905-909)
// One lane per column with the whole column of H in registers; G (= W^T W, zero padded) is broadcast from LDS.
template <int KP>
__global__ __launch_bounds__(256) void hals_h_kernel(float* __restrict__ H, int k, long n, long ldh,
                                                    const float* __restrict__ AtW, long ldatw,
                                                    const float* __restrict__ G, float eps) {
    extern __shared__ __attribute__((aligned(16))) float gs[];
    for (int idx = threadIdx.x; idx < KP * KP / 4; idx += 256)
        *reinterpret_cast<f32x4*>(&gs[idx * 4]) = *reinterpret_cast<const f32x4*>(G + idx * 4);
    __syncthreads();
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    float hc[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) hc[j] = j < k ? H[(long)j * ldh + c] : 0.f;
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
        if (kk < k) {
            float dot = 0.f;
#pragma unroll
            for (int j4 = 0; j4 < KP; j4 += 4) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(&gs[kk * KP + j4]);
                dot = fmaf(g[0], hc[j4], dot);
                dot = fmaf(g[1], hc[j4 + 1], dot);
                dot = fmaf(g[2], hc[j4 + 2], dot);
                dot = fmaf(g[3], hc[j4 + 3], dot);
            }
            const float t = hc[kk] + AtW[(long)kk * ldatw + c] - dot;
            hc[kk] = fmaxf(t, eps);
        }
    }
#pragma unroll
    for (int j = 0; j < KP; ++j)
        if (j < k) H[(long)j * ldh + c] = hc[j];
}

// Same sweep for KP = 128 with the column state in LDS instead of 128 registers per lane (runtime loops, 64 lanes per
// workgroup: hs[j][lane], G rows broadcast from LDS).
__global__ __launch_bounds__(64) void hals_h_kernel_lds(float* __restrict__ H, int k, long n, long ldh,
                                                       const float* __restrict__ AtW, long ldatw,
                                                       const float* __restrict__ G, int kp, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* gs = sm;                 // kp * kp
    float* hs = sm + kp * kp;       // kp * 64
    for (int idx = threadIdx.x; idx < kp * kp; idx += 64) gs[idx] = G[idx];
    const long c = (long)blockIdx.x * 64 + threadIdx.x;
    const bool live = c < n;
    for (int j = 0; j < k; ++j) hs[j * 64 + threadIdx.x] = live ? H[(long)j * ldh + c] : 0.f;
    __syncthreads();
    if (!live) return;
    for (int kk = 0; kk < k; ++kk) {
        float dot = 0.f;
        for (int j = 0; j < k; ++j) dot = fmaf(gs[kk * kp + j], hs[j * 64 + threadIdx.x], dot);
        const float t = hs[kk * 64 + threadIdx.x] + AtW[(long)kk * ldatw + c] - dot;
        hs[kk * 64 + threadIdx.x] = fmaxf(t, eps);
    }
    for (int j = 0; j < k; ++j) H[(long)j * ldh + c] = hs[j * 64 + threadIdx.x];
}


}  // namespace
