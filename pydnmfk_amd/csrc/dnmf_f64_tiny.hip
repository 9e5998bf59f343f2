// dnmf_f64_tiny.hip -- whole float64 fits of TINY problems as one kernel launch (round 6); part of libdnmf_hip.so.
//
// The reference's own tests factorise 24 x 12 float64 matrices with k = 2 over 2000+ iterations (tests/test_dist_nmf_1d.py:14-46,
// tests/test_dist_nmf_2d.py).  dnmf_f64_fit runs such a fit as the float64 primitives in the choreography's order: ~10 launches per
// step, each far below a microsecond of work -- 20 000 launches per fit.  Here ONE workgroup keeps A, W, H (and, for KL, the quotient
// image) in LDS for the whole fit and runs every step itself: pyDNMF.py:151-182 with dist_nmf.py:716-751 (MU/FRO), :806-849 (MU/KL),
// :873-934 with utils.py:367-391 (HALS), the clamp of pyDNMF.py:155-157, normalize_features (:185-194) and the two squared norms of
// relative_err (:205-218).  No workgroup waits for another one: nothing has to be co-resident, nothing can time out; a batch is
// gridDim.x independent workgroups.
// Arithmetic: plain float64 FMA chains, one thread per output element, every sum in index order (reductions over rows / columns by a fixed
// tree): deterministic, and within a few ulp of the primitives' MFMA sums (tests/test_gpu_f64.py holds both to the goldens at 1e-10 per
// step / 1e-8 per fit).  Shapes: everything must fit 160 KiB of LDS -- f64_tiny_lds_bytes; dnmf_f64_fit takes this kernel when it does
// and k <= 16 (the loops are O(m n k) per thread team: fine for the sizes of tests, slow beyond).
#include "dnmf_common.h"
#include "dnmf_host.h"

namespace {

constexpr int TINY_T = 256;

struct TinyArgs {
    const double* A; long lda, a_stride;
    double* W; long ldw, w_stride;
    double* H; long ldh, h_stride;
    int m, n, k, method, w_update, itr;
    double eps;
    double* sq; long sq_stride;                  // per problem {sum (A - W H)^2, sum A^2}
};

__host__ __device__ inline size_t f64_tiny_lds_doubles(long m, long n, int k, int method) {
    // A | (KL: U) | W | H | S = max(m k, k n) | G = k k | x = k | red = 256
    return (size_t)m * n * (method == 1 ? 2 : 1) + (size_t)(m + n) * k + (size_t)std::max<long>(m * k, (long)k * n) + (size_t)k * k + k + TINY_T;
}

// sum of v over the workgroup's threads, the same value in all of them (fixed tree)
__device__ __forceinline__ double tiny_sum(double v, double* red) {
    __syncthreads();
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = TINY_T / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(TINY_T) void f64_tiny_fit_kernel(TinyArgs a) {
    extern __shared__ __attribute__((aligned(16))) double tiny_smem[];
    const int m = a.m, n = a.n, k = a.k, tid = threadIdx.x;
    const double eps = a.eps;
    const long z = blockIdx.x;
    const double* Ag = a.A + z * a.a_stride;
    double* Wg = a.W + z * a.w_stride;
    double* Hg = a.H + z * a.h_stride;
    double* As = tiny_smem;                                  // [m][n]
    double* Us = As + (long)m * n;                           // [m][n] (KL only)
    double* Ws = Us + (a.method == 1 ? (long)m * n : 0);     // [m][k]
    double* Hs = Ws + (long)m * k;                           // [k][n]
    double* Ss = Hs + (long)k * n;                           // A H^T [m][k] / W^T A [k][n]
    double* Gs = Ss + (m * k > k * n ? m * k : k * n);       // [k][k]
    double* xs = Gs + k * k;                                 // [k]
    double* red = xs + k;                                    // [256]
    for (int e = tid; e < m * n; e += TINY_T) As[e] = Ag[(long)(e / n) * a.lda + e % n];
    for (int e = tid; e < m * k; e += TINY_T) Ws[e] = Wg[(long)(e / k) * a.ldw + e % k];
    for (int e = tid; e < k * n; e += TINY_T) Hs[e] = Hg[(long)(e / n) * a.ldh + e % n];
    __syncthreads();

    auto gram_h = [&]() {                                    // G = H H^T
        for (int e = tid; e < k * k; e += TINY_T) {
            const int j = e / k, l = e % k;
            double s = 0.0;
            for (int c = 0; c < n; ++c) s = fma(Hs[j * n + c], Hs[l * n + c], s);
            Gs[e] = s;
        }
    };
    auto gram_w = [&]() {                                    // G = W^T W
        for (int e = tid; e < k * k; e += TINY_T) {
            const int j = e / k, l = e % k;
            double s = 0.0;
            for (int r = 0; r < m; ++r) s = fma(Ws[r * k + j], Ws[r * k + l], s);
            Gs[e] = s;
        }
    };
    auto x_ht = [&](const double* X) {                       // S[m][k] = X H^T
        for (int e = tid; e < m * k; e += TINY_T) {
            const int r = e / k, j = e % k;
            double s = 0.0;
            for (int c = 0; c < n; ++c) s = fma(X[r * n + c], Hs[j * n + c], s);
            Ss[e] = s;
        }
    };
    auto wt_x = [&](const double* X) {                       // S[k][n] = W^T X
        for (int e = tid; e < k * n; e += TINY_T) {
            const int j = e / n, c = e % n;
            double s = 0.0;
            for (int r = 0; r < m; ++r) s = fma(Ws[r * k + j], X[r * n + c], s);
            Ss[e] = s;
        }
    };
    auto quotient = [&]() {                                  // U = A / (W H + eps)
        for (int e = tid; e < m * n; e += TINY_T) {
            const int r = e / n, c = e % n;
            double s = 0.0;
            for (int l = 0; l < k; ++l) s = fma(Ws[r * k + l], Hs[l * n + c], s);
            Us[e] = As[e] / (s + eps);
        }
    };
    auto clamp_all = [&]() {
        for (int e = tid; e < m * k; e += TINY_T) Ws[e] = Ws[e] > eps ? Ws[e] : eps;
        for (int e = tid; e < k * n; e += TINY_T) Hs[e] = Hs[e] > eps ? Hs[e] : eps;
    };

    for (int it = 0; it < a.itr; ++it) {
        if (a.method == 0) {                                                          // dist_nmf.py:716-751
            if (a.w_update) {
                gram_h(); x_ht(As);
                __syncthreads();
                for (int e = tid; e < m * k; e += TINY_T) {
                    const int r = e / k, j = e % k;
                    double d = 0.0;
                    for (int l = 0; l < k; ++l) d = fma(Ws[r * k + l], Gs[l * k + j], d);
                    Ss[e] = Ws[e] * (Ss[e] / (d + eps));                              // the new element (W *= AH / (...), :731-732), stored after every row has been read
                }
                __syncthreads();
                for (int e = tid; e < m * k; e += TINY_T) Ws[e] = Ss[e];
                __syncthreads();
            }
            gram_w(); wt_x(As);
            __syncthreads();
            for (int c = tid; c < n; c += TINY_T) {                                   // H *= AtW / (G H + eps): a thread per column, rows from a copy
                double hc[16];
                for (int l = 0; l < k; ++l) hc[l] = Hs[l * n + c];
                for (int j = 0; j < k; ++j) {
                    double d = 0.0;
                    for (int l = 0; l < k; ++l) d = fma(Gs[j * k + l], hc[l], d);
                    Hs[j * n + c] = hc[j] * (Ss[j * n + c] / (d + eps));
                }
            }
            __syncthreads();
        } else if (a.method == 1) {                                                   // dist_nmf.py:806-849
            if (a.w_update) {
                for (int j = tid; j < k; j += TINY_T) {
                    double s = 0.0;
                    for (int c = 0; c < n; ++c) s += Hs[j * n + c];
                    xs[j] = s;
                }
                quotient();
                __syncthreads();
                x_ht(Us);
                __syncthreads();
                for (int e = tid; e < m * k; e += TINY_T) Ws[e] = Ws[e] * (Ss[e] / (xs[e % k] + eps));
                __syncthreads();
            }
            for (int j = tid; j < k; j += TINY_T) {
                double s = 0.0;
                for (int r = 0; r < m; ++r) s += Ws[r * k + j];
                xs[j] = s;
            }
            quotient();
            __syncthreads();
            wt_x(Us);
            __syncthreads();
            for (int e = tid; e < k * n; e += TINY_T) Hs[e] = Hs[e] * (Ss[e] / (xs[e / n] + eps));
            __syncthreads();
        } else {                                                                      // dist_nmf.py:873-934, utils.py:367-391
            if (a.w_update) {
                gram_h(); x_ht(As);
                __syncthreads();
                for (int kk = 0; kk < k; ++kk) {
                    double acc = 0.0;
                    for (int r = tid; r < m; r += TINY_T) {
                        double dot = 0.0;
                        for (int l = 0; l < k; ++l) dot = fma(Ws[r * k + l], Gs[l * k + kk], dot);
                        double v = Ws[r * k + kk] * Gs[kk * k + kk] + Ss[r * k + kk] - dot;
                        v = v > eps ? v : eps;
                        Ws[r * k + kk] = v;
                        acc = fma(v, v, acc);
                    }
                    const double nrm = sqrt(tiny_sum(acc, red));
                    if (nrm > 0.0)
                        for (int r = tid; r < m; r += TINY_T) Ws[r * k + kk] = Ws[r * k + kk] / nrm;
                    __syncthreads();
                }
            }
            gram_w(); wt_x(As);
            __syncthreads();
            for (int c = tid; c < n; c += TINY_T)
                for (int kk = 0; kk < k; ++kk) {
                    double dot = 0.0;
                    for (int l = 0; l < k; ++l) dot = fma(Gs[kk * k + l], Hs[l * n + c], dot);
                    const double v = Hs[kk * n + c] + Ss[kk * n + c] - dot;
                    Hs[kk * n + c] = v > eps ? v : eps;
                }
            __syncthreads();
        }
        if (it % 10 == 0) {                                                           // pyDNMF.py:155-157 / :170-172
            clamp_all();
            __syncthreads();
        }
    }
    // normalize_features (pyDNMF.py:185-194)
    for (int j = tid; j < k; j += TINY_T) {
        double s = 0.0;
        for (int r = 0; r < m; ++r) s += Ws[r * k + j];
        xs[j] = s;
    }
    __syncthreads();
    for (int e = tid; e < m * k; e += TINY_T) Ws[e] = Ws[e] / (xs[e % k] + eps);
    for (int e = tid; e < k * n; e += TINY_T) Hs[e] = Hs[e] * xs[e / n];
    __syncthreads();
    // relative_err's squared norms (:205-218)
    double num = 0.0, den = 0.0;
    for (int e = tid; e < m * n; e += TINY_T) {
        const int r = e / n, c = e % n;
        double s = 0.0;
        for (int l = 0; l < k; ++l) s = fma(Ws[r * k + l], Hs[l * n + c], s);
        const double d = As[e] - s;
        num = fma(d, d, num);
        den = fma(As[e], As[e], den);
    }
    num = tiny_sum(num, red);
    den = tiny_sum(den, red);
    if (tid == 0) { a.sq[z * a.sq_stride] = num; a.sq[z * a.sq_stride + 1] = den; }
    for (int e = tid; e < m * k; e += TINY_T) Wg[(long)(e / k) * a.ldw + e % k] = Ws[e];
    for (int e = tid; e < k * n; e += TINY_T) Hg[(long)(e / n) * a.ldh + e % n] = Hs[e];
}

}  // namespace

// 1: the shape does not take the kernel (the caller keeps the chain of primitives); DNMF_OK: launched
__attribute__((visibility("hidden"))) int dnmf_f64_tiny_fit_(int method, const double* A, long m, long n, long lda, long a_stride, double* W, long ldw,
                                                              long w_stride, double* H, long ldh, long h_stride, int k, double eps, int w_update, int itr,
                                                              int batch, double* sq_out, long sq_stride, void* stream) {
    if (k > 16 || m > 4096 || n > 4096 || itr < 1 || batch < 1) return 1;
    const size_t lds = f64_tiny_lds_doubles(m, n, k, method) * sizeof(double);
    if (lds > 160 * 1024 - 64) return 1;
    static bool once = false;
    if (!once) { allow_lds(f64_tiny_fit_kernel, 160 * 1024); once = true; }
    TinyArgs a{};
    a.A = A; a.lda = lda; a.a_stride = a_stride; a.W = W; a.ldw = ldw; a.w_stride = w_stride; a.H = H; a.ldh = ldh; a.h_stride = h_stride;
    a.m = (int)m; a.n = (int)n; a.k = k; a.method = method; a.w_update = w_update; a.itr = itr; a.eps = eps;
    a.sq = sq_out; a.sq_stride = sq_stride;
    hipLaunchKernelGGL(f64_tiny_fit_kernel, dim3((unsigned)batch), dim3(TINY_T), lds, S(stream), a);
    return check_launch("f64_tiny_fit_kernel");
}

extern "C" {

int dnmf_f64_fit_tiny(long m, long n, int k, int method) {
    return (m >= 1 && n >= 1 && k >= 1 && k <= 16 && m <= 4096 && n <= 4096 && method >= 0 && method <= 2 &&
            f64_tiny_lds_doubles(m, n, k, method) * sizeof(double) <= 160 * 1024 - 64) ? 1 : 0;
}

}  // extern "C"
