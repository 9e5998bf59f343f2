// dnmf_wide.hip -- ranks 128 < k <= 256 (the reference has no bound on k, dist_nmf.py:618-632).
//
// The tuned kernels keep their rank limit: their tiles are cut for KP <= 128 (an accumulator strip of KP columns per wave, G
// staged in LDS as KP x KP).  A wider rank is served by composition, one level above them:
//   * the two big contractions split EXACTLY along k -- A H^T by rows of H, W^T A by columns of W -- into two passes of the
//     tuned kernels over A (csrc/dnmf.hip: aht_impl / wta_impl), the Gram matrices into four block products;
//   * what contracts OVER k -- W G, G H in the multiplicative updates, S = W H in the KL quotient and the residual -- runs on the
//     plain kernels of this file: v_mfma_f32_16x16x4_f32, the wave's strip of the left operand in registers (up to 64
//     contraction steps), the right operand streamed from L2, no LDS.  The KL products go through the materialised quotient
//     U = A / (W H + eps) (as the reference, dist_nmf.py:806, and the float64 path do), followed by the tuned contractions on U.
// Every launch goes through DNMF_LAUNCH: batched fits (csrc/dnmf_fit.hip) cover these kernels too.
//
// v_mfma_f32_16x16x4_f32 operand maps (lane l, i = l & 15, q = l >> 4):
//   A-operand: A[row i][kk = q]   B-operand: B[kk = q][col i]   C/D: col = i, row = 4 q + reg (reg in [0, 4))
#include "dnmf_common.h"
#include "dnmf_host.h"
#include "dnmf_stream.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {

constexpr int WKS = 64;                       // contraction steps of 4: k <= 256

enum { W_QUOT = 0, W_SQSUM = 1, W_COLERR = 2, W_UPD_W = 3 };

// S[r][c] = sum_j X[r][j] Y[j][c] for the wave's 16 rows and a range of 16-column tiles (X = the wave's strip, in registers):
//   W_QUOT    O[r][c] = A[r][c] / (S + eps)                                   (dist_nmf.py:806)
//   W_SQSUM   *dsum += sum (A - S)^2                                          (pyDNMF.py:207)
//   W_COLERR  dsum[c] += sum_r (A - S)^2,  den[c] += sum_r A^2                (pyDNMF.py:229-230)
//   W_UPD_W   O[r][c] *= A[r][c] / (S + eps) with O = X = W, Y = G, A = A H^T (dist_nmf.py:731-732; the wave owns its rows of W
//             and holds them in registers before it writes)
template <int MODE>
__global__ __launch_bounds__(256) void wide_nn_rows_kernel(const float* X, long ldx, long m, int kc, const float* __restrict__ Y, long ldy,
                                                           long n, const float* __restrict__ A, long lda, float* O, long ldo, float eps,
                                                           long cols_per_wave, double* __restrict__ dsum, double* __restrict__ den, BatchTab bt) {
    REBASE(X); REBASE(Y); REBASE(A); REBASE(O); REBASE(dsum); REBASE(den);
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    const int ksn = (kc + 3) / 4;
    double acc_sq = 0.0;
    if (r0 < m) {
        float xa[WKS];
#pragma unroll
        for (int s = 0; s < WKS; ++s) xa[s] = (r0 + i < m && 4 * s + q < kc) ? X[(r0 + i) * ldx + 4 * s + q] : 0.f;
        const long cb = (long)blockIdx.y * cols_per_wave;
        const long ce = cb + cols_per_wave < n ? cb + cols_per_wave : n;
        for (long c0 = cb; c0 < ce; c0 += 16) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const bool cok = c0 + i < n;
#pragma unroll
            for (int s = 0; s < WKS; ++s) {
                if (s < ksn) {                                             // (wave-uniform: k = 130 runs 33 of the 64 steps)
                    const float b = (cok && 4 * s + q < kc) ? Y[(long)(4 * s + q) * ldy + c0 + i] : 0.f;
                    acc = MFMA16(xa[s], b, acc);
                }
            }
            double tn = 0.0, td = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = r0 + 4 * q + r;
                if (row < m && cok) {
                    const float a = A[row * lda + c0 + i];
                    if constexpr (MODE == W_QUOT) O[row * ldo + c0 + i] = a / (acc[r] + eps);
                    else if constexpr (MODE == W_UPD_W) O[row * ldo + c0 + i] = O[row * ldo + c0 + i] * (a / (acc[r] + eps));
                    else {
                        const float d = a - acc[r];
                        tn += (double)d * (double)d;
                        td += (double)a * (double)a;
                    }
                }
            }
            if constexpr (MODE == W_SQSUM) acc_sq += tn;
            if constexpr (MODE == W_COLERR) {                              // the four row groups of a column live in lanes i, i + 16, ...
                tn += __shfl_xor(tn, 16); tn += __shfl_xor(tn, 32);
                td += __shfl_xor(td, 16); td += __shfl_xor(td, 32);
                if (q == 0 && cok) { atomicAdd(dsum + c0 + i, tn); atomicAdd(den + c0 + i, td); }
            }
        }
    }
    if constexpr (MODE == W_SQSUM) block_atomic_sum(acc_sq, dsum);
}

// H[j][c] *= S[j][c] / ((G H)[j][c] + eps) for the wave's 16 columns and all rows j (dist_nmf.py:750-751); the wave holds its
// columns of H in registers before it writes them.  clamp: H = max(H, eps) afterwards (pyDNMF.py:156)
__global__ __launch_bounds__(256) void wide_upd_h_kernel(float* H, int k, long n, long ldh, const float* __restrict__ Sm, long lds_,
                                                         const float* __restrict__ G, long ldg, float eps, int clamp, BatchTab bt) {
    REBASE(H); REBASE(Sm); REBASE(G);
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const long c0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (c0 >= n) return;
    const bool cok = c0 + i < n;
    const int ksn = (k + 3) / 4;
    float hb[WKS];
#pragma unroll
    for (int s = 0; s < WKS; ++s) hb[s] = (cok && 4 * s + q < k) ? H[(long)(4 * s + q) * ldh + c0 + i] : 0.f;
    for (int jt = 0; 16 * jt < k; ++jt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < WKS; ++s) {
            if (s < ksn) {
                const float a = (16 * jt + i < k && 4 * s + q < k) ? G[(long)(16 * jt + i) * ldg + 4 * s + q] : 0.f;
                acc = MFMA16(a, hb[s], acc);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + 4 * q + r;
            if (j < k && cok) {
                float v = H[(long)j * ldh + c0 + i] * (Sm[(long)j * lds_ + c0 + i] / (acc[r] + eps));
                if (clamp) v = fmaxf(v, eps);
                H[(long)j * ldh + c0 + i] = v;
            }
        }
    }
}

// HALS H sweep for wide ranks: rows in sequence, a thread per column, H and G read through the caches (dist_nmf.py:905-909)
__global__ __launch_bounds__(256) void wide_hals_h_kernel(float* H, int k, long n, long ldh, const float* __restrict__ AtW, long ldatw,
                                                          const float* __restrict__ G, long ldg, float eps, BatchTab bt) {
    REBASE(H); REBASE(AtW); REBASE(G);
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    for (int kk = 0; kk < k; ++kk) {
        float dot = 0.f;
        for (int l = 0; l < k; ++l) dot = fmaf(G[(long)kk * ldg + l], H[(long)l * ldh + c], dot);
        const float v = H[(long)kk * ldh + c] + AtW[(long)kk * ldatw + c] - dot;
        H[(long)kk * ldh + c] = fmaxf(v, eps);
    }
}

template <int MODE>
int launch_rows(const float* X, long ldx, long m, int kc, const float* Y, long ldy, long n, const float* A, long lda, float* O, long ldo,
                float eps, double* dsum, double* den, hipStream_t st) {
    long cpw = n;                                                   // W_UPD_W: the wave owns its rows across the whole width
    if (MODE != W_UPD_W) {
        const long want = std::max<long>(1, 8192 / cdiv(m, 16));
        cpw = std::max<long>(64, round_up(cdiv(n, want), 16));
    }
    const dim3 grid((unsigned)cdiv(cdiv(m, 16), 4), (unsigned)cdiv(n, cpw));
    DNMF_LAUNCH((wide_nn_rows_kernel<MODE>), grid, dim3(256), 0, st, X, ldx, m, kc, Y, ldy, n, A, lda, O, ldo, eps, cpw, dsum, den);
    return check_launch("wide nn");
}

}  // namespace

// library-internal entry points (declared where they are called: csrc/dnmf.hip, dnmf_kl.hip, dnmf_hals.hip)
#define HID __attribute__((visibility("hidden")))
HID int dnmf_wide_mu_update_w_(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, long ldg, float eps, void* stream);
HID int dnmf_wide_mu_update_h_(float* H, int k, long n, long ldh, const float* AtW, long ldatw, const float* G, long ldg, float eps, int clamp,
                               void* stream);
HID int dnmf_wide_quot_(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k, float eps, float* U,
                        long ldu, void* stream);
HID int dnmf_wide_resid_(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k, double* out,
                         void* stream);
HID int dnmf_wide_column_err_(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k, double* num,
                              double* den, void* stream);
HID int dnmf_wide_hals_update_h_(float* H, int k, long n, long ldh, const float* AtW, long ldatw, const float* G, long ldg, float eps, void* stream);

int dnmf_wide_mu_update_w_(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, long ldg, float eps, void* stream) {
    return launch_rows<W_UPD_W>(W, ldw, m, k, G, ldg, k, AH, ldah, W, ldw, eps, nullptr, nullptr, S(stream));
}
int dnmf_wide_mu_update_h_(float* H, int k, long n, long ldh, const float* AtW, long ldatw, const float* G, long ldg, float eps, int clamp,
                           void* stream) {
    DNMF_LAUNCH(wide_upd_h_kernel, dim3((unsigned)cdiv(cdiv(n, 16), 4)), dim3(256), 0, S(stream), H, k, n, ldh, AtW, ldatw, G, ldg, eps, clamp);
    return check_launch("wide mu_update_h");
}
int dnmf_wide_quot_(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k, float eps, float* U,
                    long ldu, void* stream) {
    return launch_rows<W_QUOT>(W, ldw, m, k, H, ldh, n, A, lda, U, ldu, eps, nullptr, nullptr, S(stream));
}
int dnmf_wide_resid_(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k, double* out,
                     void* stream) {
    hipStream_t st = S(stream);
    if (batch_memset(out, 0, sizeof(double), st) != hipSuccess) return fail(DNMF_EHIP, "resid_sqnorm: memset failed");
    return launch_rows<W_SQSUM>(W, ldw, m, k, H, ldh, n, A, lda, nullptr, 0, 0.f, out, nullptr, st);
}
int dnmf_wide_column_err_(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k, double* num,
                          double* den, void* stream) {
    return launch_rows<W_COLERR>(W, ldw, m, k, H, ldh, n, A, lda, nullptr, 0, 0.f, num, den, S(stream));
}
int dnmf_wide_hals_update_h_(float* H, int k, long n, long ldh, const float* AtW, long ldatw, const float* G, long ldg, float eps, void* stream) {
    DNMF_LAUNCH(wide_hals_h_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, S(stream), H, k, n, ldh, AtW, ldatw, G, ldg, eps);
    return check_launch("wide hals_update_h");
}
