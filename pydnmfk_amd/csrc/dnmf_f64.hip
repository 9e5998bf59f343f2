// dnmf_f64.hip -- the update path in float64 (the reference computes in the dtype of A_ij, pyDNMF.py:68, and its own tests feed
// np.random.rand float64 arrays, tests/test_dist_nmf_1d.py:14-20).  One tile shape per kernel family on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64), operands straight from memory into registers (no LDS); round 6 made the operand streams branch-free
// (MUBUF descriptor loads, "MUBUF operand loads" below) and fused the KL quotient into its products (dnmf_f64_kl.h):
// 77-88 % MFMA busy at 65536 x 4096, k = 64 (profiles/r06_f64_*).
//
// v_mfma_f64_16x16x4_f64 operand maps (lane l, i = l & 15, q = l >> 4):
//   A-operand: A[row i][kk = q]   B-operand: B[kk = q][col i]   C/D: col = i, row = q + 4 * reg (reg in [0, 4))
// Three GEMM forms (cf. csrc/dnmf.hip):
//   NT  C[r][j] = sum_c X[r][c] Y[j][c]      A H^T, H H^T      (contraction index contiguous in both operands)
//   TN  C[j][c] = sum_r X[r][j] Y[r][c]      W^T A, W^T W      (contraction over the rows; partial slabs + ordered reduction)
//   NN  S[r][c] = sum_j X[r][j] Y[j][c]      W H, W G, G H     (fused epilogues: quotient, squared residual, the MU updates)
// The contraction order inside a tile is permuted freely (sums are order-agnostic up to rounding).
#include "dnmf_common.h"

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

namespace {

#define REQ(cond, ...) do { if (!(cond)) return fail(DNMF_EINVAL, __VA_ARGS__); } while (0)
inline hipStream_t ST(void* s) { clear_hip_error(); return reinterpret_cast<hipStream_t>(s); }
inline size_t al256(size_t x) { return (x + 255) & ~size_t(255); }
inline int tiles16(int k) { return (k + 15) / 16; }

// four consecutive doubles row[c .. c + 3]; zero outside [.., cend) or when !ok.  vec: the row pointer is 16-byte aligned and c is even
__device__ __forceinline__ void ld4(double (&d)[4], const double* __restrict__ row, long c, long cend, bool ok, bool vec) {
    if (ok && vec && c + 4 <= cend) {
        const f64x2 a = *reinterpret_cast<const f64x2*>(row + c), b = *reinterpret_cast<const f64x2*>(row + c + 2);
        d[0] = a[0]; d[1] = a[1]; d[2] = b[0]; d[3] = b[1];
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = (ok && c + e < cend) ? row[c + e] : 0.0;
    }
}

// N consecutive doubles (N = 1, 2, 4) as ld4 above
template <int N>
__device__ __forceinline__ void ldn(double (&d)[N], const double* __restrict__ row, long c, long cend, bool ok, bool vec) {
    if constexpr (N == 4) ld4(d, row, c, cend, ok, vec);
    else if constexpr (N == 2) {
        if (ok && vec && c + 2 <= cend) { const f64x2 a = *reinterpret_cast<const f64x2*>(row + c); d[0] = a[0]; d[1] = a[1]; }
        else { d[0] = (ok && c < cend) ? row[c] : 0.0; d[1] = (ok && c + 1 < cend) ? row[c + 1] : 0.0; }
    } else d[0] = (ok && c < cend) ? row[c] : 0.0;
}

// ---- MUBUF operand loads (dnmf_common.h "buffer addressing").  Round 6: the flat loads above sit behind runtime branches (vector or
// element-wise, in or out of range), and hipcc answers a load under a branch with s_waitcnt vmcnt(0) at the next use -- the "two
// register sets" of the NT / TN kernels never overlapped anything (3 waits in f64_nt_kernel<4,4>, all vmcnt(0), behind 335 branches).
// A buffer load needs no branch: a lane is switched off through its offset, rows beyond the operand fall outside the descriptor's
// byte count (the hardware returns 0), a 16-byte access only needs 4-byte alignment.
// descriptor over `bytes` bytes from `base` (both wave-uniform; at most 2 GiB: offsets are 31-bit, BUF_OOB is the off switch)
__device__ __forceinline__ i32x4 rsrc64(const double* base, long bytes) {
    i32x4 r = buf_rsrc(base);
    const long b = bytes < 0 ? 0 : (bytes > 0x7fffffffL ? 0x7fffffffL : bytes);
    r[2] = __builtin_amdgcn_readfirstlane((int)b);
    return r;
}
// N consecutive doubles (N = 1, 2, 4) at byte offset vo + soff; `nvalid` of them are in range (the rest: zero).  VEC: nvalid is 0 or N
// by construction of the caller (column counts divisible by N), so the lane moves as a whole
template <int N, bool VEC>
__device__ __forceinline__ void ldq(double (&d)[N], i32x4 rs, int vo, int soff, int nvalid) {
    if constexpr (VEC) {
        const int o = nvalid > 0 ? vo : BUF_OOB;
        if constexpr (N == 4) {
            const f32x4 a = buf_ld_f32x4(rs, o, soff, 0), b = buf_ld_f32x4(rs, o + 16, soff, 0);
            const f64x2 x = __builtin_bit_cast(f64x2, a), y = __builtin_bit_cast(f64x2, b);
            d[0] = x[0]; d[1] = x[1]; d[2] = y[0]; d[3] = y[1];
        } else if constexpr (N == 2) {
            const f64x2 x = __builtin_bit_cast(f64x2, buf_ld_f32x4(rs, o, soff, 0));
            d[0] = x[0]; d[1] = x[1];
        } else d[0] = __builtin_bit_cast(double, buf_ld_f32x2(rs, o, soff, 0));
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) d[e] = __builtin_bit_cast(double, buf_ld_f32x2(rs, e < nvalid ? vo + 8 * e : BUF_OOB, soff, 0));
    }
}
__device__ __forceinline__ int sgpr(long x) { return __builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ int clampi(long x, int hi) { return x < 0 ? 0 : (x > hi ? hi : (int)x); }

// ================================================================================================ NT form
// out[split][r][j] = sum_{c in split} X[r][c] Y[j][c];  one wave = RT 16-row tiles x NT 16-column tiles of the output.
// Round 5 (second pass): the first version gave a wave ONE row tile, so every wave read all of Y for 16 rows of X -- four times the
// bytes of X through the L2 at k = 64 -- and nothing was in flight while its 16 MFMAs ran.  Now a Y fragment feeds RT row tiles
// (RT NT = 16 accumulators of 16 x 16), the loads of chunk c + 16 are issued before the MFMAs of chunk c (two register sets), and
// consecutive MFMAs are independent (e outermost).
// CH = 16-column chunks per step: lane group q owns the 4 CH CONSECUTIVE columns c + 4 CH q .. of a step in both operands (the order of
// the contraction is free), so a row moves in pieces of 128 CH bytes.
constexpr int nt_ch(int rt, int nt) { return rt * nt <= 8 ? 2 : 1; }      // (two register sets of loads must fit 256 VGPRs)
// VEC: n % 4 == 0 (a lane's four columns are in range together)
template <int RT, int NT, bool VEC, int D = 2>      // D = register sets of loads in flight (chunk c + D - 1 is issued before the MFMAs of chunk c)
__global__ __launch_bounds__(256) void f64_nt_kernel(const double* __restrict__ X, long ldx, long m, long n, const double* __restrict__ Y,
                                                     long ldy, int kc, double* __restrict__ out, long ldo, long split_stride,
                                                     long cols_per_split) {
    constexpr int CH = nt_ch(RT, NT);
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const long r0 = ((long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * (16 * RT);
    if (r0 >= m) return;
    const long cb = (long)blockIdx.y * cols_per_split;
    const long ce = cb + cols_per_split < n ? cb + cols_per_split : n;
    f64x4 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[rt][t] = f64x4{0.0, 0.0, 0.0, 0.0};
    // the wave's rows of X behind one descriptor (rows beyond m: outside its byte count), one descriptor per 16-row tile of Y
    const long rows = m - r0 < 16 * RT ? m - r0 : 16 * RT;
    const i32x4 xd = rsrc64(X + r0 * ldx, ((rows - 1) * ldx + n) * 8);
    i32x4 yd[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) yd[t] = rsrc64(Y + (long)16 * t * ldy, ((long)(kc - 16 * t - 1) * ldy + n) * 8);
    int xvo[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) xvo[rt] = (int)(((16 * rt + i) * ldx + 4 * CH * q) * 8);
    const int yvo = (int)((i * ldy + 4 * CH * q) * 8);
    auto load = [&](double (&a)[RT][CH][4], double (&b)[NT][CH][4], long c) __attribute__((always_inline)) {
        const int so = sgpr(c * 8);
#pragma unroll
        for (int v = 0; v < CH; ++v) {
            const int nv = clampi(ce - (c + 4 * CH * q + 4 * v), 4);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) ldq<4, VEC>(a[rt][v], xd, xvo[rt] + 32 * v, so, nv);
#pragma unroll
            for (int t = 0; t < NT; ++t) ldq<4, VEC>(b[t][v], yd[t], yvo + 32 * v, so, nv);
        }
    };
    auto mma = [&](const double (&a)[RT][CH][4], const double (&b)[NT][CH][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < CH; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[rt][t] = MFMA64(a[rt][v][e], b[t][v][e], acc[rt][t]);
    };
    constexpr long STEP = 16 * CH;
    double a[D][RT][CH][4], b[D][NT][CH][4];
    const long nch = cdiv(ce - cb, STEP);
    auto colof = [&](long it) __attribute__((always_inline)) { return cb + it * STEP; };
    // A load under a condition makes hipcc's wait counts conservative at the join (it then waits for the loads it has just issued:
    // the same lesson as csrc/dnmf_team.h), so nothing here is conditional: the trip count is rounded up to the ring depth, a chunk
    // beyond the range loads zeros (all its lanes are switched off) and its MFMAs add nothing.
    static_for<0, D - 1>([&](auto d) __attribute__((always_inline)) { load(a[d], b[d], colof(d)); });
    for (long it = 0; it < nch; it += D)
        static_for<0, D>([&](auto d) __attribute__((always_inline)) {
            load(a[(d + D - 1) % D], b[(d + D - 1) % D], colof(it + d + (D - 1)));
            __builtin_amdgcn_sched_barrier(0);                    // (the loads of a step go out before its MFMAs: hipcc otherwise spreads them over the block)
            mma(a[d], b[d]);
        });
    double* o = out + (long)blockIdx.y * split_stride;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = r0 + 16 * rt + q + 4 * r;
                const int col = 16 * t + i;
                if (row < m && col < kc) o[row * ldo + col] = acc[rt][t][r];
            }
}

// ================================================================================================ TN form
// P[chunk][j][c] = sum_{r in chunk} X[r][j] Y[r][c];  one wave = 16 CT columns of Y x NT 16-row tiles of the output (CT NT = 16
// accumulators: CT = 4 up to k = 64, 2 beyond).
// The MFMA's 16 columns / 16 rows are a free permutation of the output's: column tile cb holds the columns c0 + CT i + cb, so lane i
// reads CT consecutive doubles of a row of Y with one access (and writes its results the same way); row tile t holds the factor
// columns (t / VJ) 16 VJ + VJ i + t % VJ (VJ = min(NT, 4) consecutive doubles of a row of X per lane).
// Operands through per-tile descriptors (16 rows from the tile's first: rows beyond the chunk lie outside the byte count).
// VEC: n % CT == 0 and kc % VJ == 0.
constexpr int tn_ct(int nt) { return nt <= 4 ? 4 : 2; }
template <int NT, bool VEC, int D = 2>
__global__ __launch_bounds__(256) void f64_tn_kernel(const double* __restrict__ X, long ldx, int kc, const double* __restrict__ Y, long ldy,
                                                     long n, long m, long rows_per_chunk, int ncolblk, long nwaves,
                                                     double* __restrict__ P, long chunk_stride, long ldp) {
    constexpr int VJ = NT < 4 ? NT : 4, NV = NT / VJ, CT = tn_ct(NT);
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const long gw = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (gw >= nwaves) return;
    const long chunk = gw / ncolblk, c0 = (gw % ncolblk) * (16 * CT);
    const long rb = chunk * rows_per_chunk;
    const long re = rb + rows_per_chunk < m ? rb + rows_per_chunk : m;
    f64x4 acc[NT][CT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int cb = 0; cb < CT; ++cb) acc[t][cb] = f64x4{0.0, 0.0, 0.0, 0.0};
    int yvo[4], xvo[4], nvx[NV];
    const int nvy = clampi(n - (c0 + CT * i), CT);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        yvo[u] = (int)(((q + 4 * u) * ldy + c0 + CT * i) * 8);
        xvo[u] = (int)(((q + 4 * u) * ldx + VJ * i) * 8);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) nvx[v] = clampi(kc - (16 * VJ * v + VJ * i), VJ);
    auto load = [&](double (&a)[4][NV][VJ], double (&b)[4][CT], long r) __attribute__((always_inline)) {
        const i32x4 yd = rsrc64(Y + r * ldy, ((re - r - 1) * ldy + n) * 8), xd = rsrc64(X + r * ldx, ((re - r - 1) * ldx + kc) * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u) {                              // four steps of 4 rows
            ldq<CT, VEC>(b[u], yd, yvo[u], 0, nvy);
#pragma unroll
            for (int v = 0; v < NV; ++v) ldq<VJ, VEC>(a[u][v], xd, xvo[u] + 16 * VJ * 8 * v, 0, nvx[v]);
        }
    };
    auto mma = [&](const double (&a)[4][NV][VJ], const double (&b)[4][CT]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int cb = 0; cb < CT; ++cb) acc[t][cb] = MFMA64(a[u][t / VJ][t % VJ], b[u][cb], acc[t][cb]);
    };
    double a[D][4][NV][VJ], b[D][4][CT];
    // (nothing conditional around a load, see f64_nt_kernel: a tile beyond the range has descriptors of zero bytes and reads zeros)
    static_for<0, D - 1>([&](auto d) __attribute__((always_inline)) { load(a[d], b[d], rb + 16 * d); });
    for (long r = rb; r < re; r += 16 * D)
        static_for<0, D>([&](auto d) __attribute__((always_inline)) {
            load(a[(d + D - 1) % D], b[(d + D - 1) % D], r + 16 * (d + D - 1));
            __builtin_amdgcn_sched_barrier(0);
            mma(a[d], b[d]);
        });
    double* o = P + chunk * chunk_stride;
    const bool vst = (ldp % 2 == 0) && (((uintptr_t)P & 15) == 0) && (chunk_stride % 2 == 0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ip = q + 4 * r;
            const int j = (t / VJ) * 16 * VJ + VJ * ip + t % VJ;
            if (j >= kc) continue;
            double* dst = o + (long)j * ldp + c0 + CT * i;
            if (vst && c0 + CT * i + CT <= n) {
#pragma unroll
                for (int cb = 0; cb < CT; cb += 2) *reinterpret_cast<f64x2*>(dst + cb) = f64x2{acc[t][cb][r], acc[t][cb + 1][r]};
            } else {
#pragma unroll
                for (int cb = 0; cb < CT; ++cb)
                    if (c0 + CT * i + cb < n) dst[cb] = acc[t][cb][r];
            }
        }
}

// out[r][c] = sum_s P[s][r][c] in slab order (bitwise deterministic); a thread owns two consecutive columns (cols even, 16-byte rows:
// the host checks) or one
template <int V>
__global__ __launch_bounds__(256) void f64_reduce_kernel(const double* __restrict__ P, long stride, long ldp, int nsplit,
                                                         double* __restrict__ out, long ldo, long rows, long cols) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long cv = cols / V;
    if (idx >= rows * cv) return;
    const long r = idx / cv, c = (idx % cv) * V;
    const double* p = P + r * ldp + c;
    double s[V];
#pragma unroll
    for (int e = 0; e < V; ++e) s[e] = 0.0;
    int k = 0;
    for (; k + 4 <= nsplit; k += 4) {                              // four slabs in flight, added in slab order
        double v[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (V == 2) { const f64x2 t = *reinterpret_cast<const f64x2*>(p + (long)(k + u) * stride); v[u][0] = t[0]; v[u][1] = t[1]; }
            else v[u][0] = p[(long)(k + u) * stride];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < V; ++e) s[e] += v[u][e];
    }
    for (; k < nsplit; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) s[e] += p[(long)k * stride + e];
#pragma unroll
    for (int e = 0; e < V; ++e) out[r * ldo + c + e] = s[e];
}
inline void launch_reduce64(const double* P, long stride, long ldp, int nsplit, double* out, long ldo, long rows, long cols, hipStream_t st) {
    const bool v2 = cols % 2 == 0 && ldp % 2 == 0 && stride % 2 == 0 && ((uintptr_t)P & 15) == 0;
    if (v2) hipLaunchKernelGGL(f64_reduce_kernel<2>, dim3((unsigned)cdiv(rows * (cols / 2), 256)), dim3(256), 0, st, P, stride, ldp, nsplit, out, ldo, rows, cols);
    else hipLaunchKernelGGL(f64_reduce_kernel<1>, dim3((unsigned)cdiv(rows * cols, 256)), dim3(256), 0, st, P, stride, ldp, nsplit, out, ldo, rows, cols);
}

// ================================================================================================ NN form, row strips
// S[r][c] = sum_j X[r][j] Y[j][c] for the wave's 16 rows (its X fragment stays in registers) and a range of 16-column tiles:
//   NN_QUOT    O[r][c] = A[r][c] / (S + eps)              (the KL quotient U, dist_nmf.py:806)
//   NN_SQDIFF  O[r][c] = (A[r][c] - S)^2                  (relative_err / column_err, pyDNMF.py:207, :229)
//   NN_UPD_W   X[r][c] *= A[r][c] / (S + eps), X = W, Y = G, A = A H^T   (dist_nmf.py:731-732; the wave owns its rows of W and has
//              read them into registers before it writes)
enum { NN_QUOT = 0, NN_SQDIFF = 1, NN_UPD_W = 2 };
template <int KS, int MODE>       // KS = contraction steps of 4 (kc <= 4 KS)
__global__ __launch_bounds__(256) void f64_nn_rows_kernel(const double* X, long ldx, long m, int kc, const double* __restrict__ Y,
                                                          long ldy, long n, const double* __restrict__ A, long lda, double* O,
                                                          long ldo, double eps, long cols_per_wave) {
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (r0 >= m) return;
    double xa[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) xa[s] = (r0 + i < m && 4 * s + q < kc) ? X[(r0 + i) * ldx + 4 * s + q] : 0.0;
    const long cb = (long)blockIdx.y * cols_per_wave;
    const long ce = cb + cols_per_wave < n ? cb + cols_per_wave : n;
    for (long c0 = cb; c0 < ce; c0 += 16) {
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        const bool cok = c0 + i < n;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const double b = (cok && 4 * s + q < kc) ? Y[(long)(4 * s + q) * ldy + c0 + i] : 0.0;
            acc = MFMA64(xa[s], b, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = r0 + q + 4 * r;
            if (row < m && cok) {
                const double a = A[row * lda + c0 + i];
                if constexpr (MODE == NN_QUOT) O[row * ldo + c0 + i] = a / (acc[r] + eps);
                else if constexpr (MODE == NN_SQDIFF) { const double d = a - acc[r]; O[row * ldo + c0 + i] = d * d; }
                else O[row * ldo + c0 + i] = O[row * ldo + c0 + i] * (a / (acc[r] + eps));
            }
        }
    }
}

// The two m x n images (quotient, squared residual), column-resident: one wave owns 16 CT columns -- its k x 16 CT fragment of H stays
// in registers -- and walks down a range of rows 16 at a time, the loads of the next row tile (its rows of W and its 16 x 16 CT piece of
// A) issued before the MFMAs of this one.  The 16 CT columns are permuted as in the TN form (tile cb = columns c0 + CT i + cb: rows of
// H, A and O move as CT consecutive doubles per lane); the contraction index of step s in lane group q is q KS + s, so a lane reads KS
// CONSECUTIVE doubles of its row of W.  (The first version kept 16 rows of W resident and re-read all of H per row tile -- four
// times A's bytes through the L2 at k = 64 -- with nothing in flight under its MFMAs: 2.67 ms at 65536 x 4096, k = 64.)
constexpr int nn_ct(int ks) { return ks <= 16 ? 4 : 2; }
// N consecutive doubles to byte offset vo of a descriptor, `nvalid` of them in range (VEC: 0 or N) -- the store twin of ldq: no branch
// (a store under a branch makes the wait counts of the loads after it conservative, exactly as a load does); never with an SGPR offset
// (dnmf_common.h RULE)
template <int N, bool VEC>
__device__ __forceinline__ void stq(const double (&d)[N], i32x4 rs, int vo, int nvalid) {
    if constexpr (VEC && N == 4) {
        const int o = nvalid > 0 ? vo : BUF_OOB;
        buf_st_f32x4(__builtin_bit_cast(f32x4, f64x2{d[0], d[1]}), rs, o, 0, 0);
        buf_st_f32x4(__builtin_bit_cast(f32x4, f64x2{d[2], d[3]}), rs, o + 16, 0, 0);
    } else if constexpr (VEC && N == 2) {
        buf_st_f32x4(__builtin_bit_cast(f32x4, f64x2{d[0], d[1]}), rs, nvalid > 0 ? vo : BUF_OOB, 0, 0);
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) buf_st_f32x2(__builtin_bit_cast(f32x2, d[e]), rs, e < nvalid ? vo + 8 * e : BUF_OOB, 0, 0);
    }
}
// Round 6: descriptor loads and stores, nothing conditional around a memory instruction (f64_nt_kernel).  VEC: n % CT == 0, kc % 4 == 0.
template <int KS, int MODE, bool VEC, int D = 2>
__global__ __launch_bounds__(256) void f64_nn_cols_kernel(const double* __restrict__ X, long ldx, long m, int kc, const double* __restrict__ Y,
                                                          long ldy, long n, const double* __restrict__ A, long lda, double* __restrict__ O,
                                                          long ldo, double eps, long rows_per_wave, int ncolblk, long nwaves) {
    constexpr int CT = nn_ct(KS);
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const long gw = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (gw >= nwaves) return;
    const long chunk = gw / ncolblk, c0 = (gw % ncolblk) * (16 * CT);
    const long rb = chunk * rows_per_wave;
    const long re = rb + rows_per_wave < m ? rb + rows_per_wave : m;
    const int nva = clampi(n - (c0 + CT * i), CT);
    double hb[KS][CT];
    {
        const i32x4 hd = rsrc64(Y, ((long)(kc - 1) * ldy + n) * 8);
#pragma unroll
        for (int s = 0; s < KS; ++s) ldq<CT, VEC>(hb[s], hd, q * KS + s < kc ? (int)(((q * KS + s) * ldy + c0 + CT * i) * 8) : BUF_OOB, 0, nva);
    }
    const int xvo = (int)((i * ldx + q * KS) * 8);
    int avo[4], ovo[4], nvx[KS / 4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        avo[r] = (int)(((q + 4 * r) * lda + c0 + CT * i) * 8);
        ovo[r] = (int)(((q + 4 * r) * ldo + c0 + CT * i) * 8);
    }
#pragma unroll
    for (int v = 0; v < KS / 4; ++v) nvx[v] = clampi(kc - (q * KS + 4 * v), 4);
    auto load = [&](double (&xa)[KS], double (&av)[4][CT], long r0) __attribute__((always_inline)) {
        const i32x4 xd = rsrc64(X + r0 * ldx, ((re - r0 - 1) * ldx + kc) * 8), ad = rsrc64(A + r0 * lda, ((re - r0 - 1) * lda + n) * 8);
#pragma unroll
        for (int v = 0; v < KS / 4; ++v) {
            double t4[4];
            ldq<4, VEC>(t4, xd, xvo + 32 * v, 0, nvx[v]);
#pragma unroll
            for (int e = 0; e < 4; ++e) xa[4 * v + e] = t4[e];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ldq<CT, VEC>(av[r], ad, avo[r], 0, nva);
    };
    auto tile = [&](const double (&xa)[KS], const double (&av)[4][CT], long r0) __attribute__((always_inline)) {
        const i32x4 od = rsrc64(O + r0 * ldo, ((re - r0 - 1) * ldo + n) * 8);       // (rows beyond the range: outside the byte count, dropped)
        f64x4 acc[CT];
#pragma unroll
        for (int cb = 0; cb < CT; ++cb) acc[cb] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int cb = 0; cb < CT; ++cb) acc[cb] = MFMA64(xa[s], hb[s][cb], acc[cb]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double o[CT];
#pragma unroll
            for (int cb = 0; cb < CT; ++cb) {
                const double a = av[r][cb], sum = acc[cb][r];
                if constexpr (MODE == NN_QUOT) o[cb] = a / (sum + eps);
                else { const double d = a - sum; o[cb] = d * d; }
            }
            stq<CT, VEC>(o, od, ovo[r], nva);
        }
    };
    double xa[D][KS], av[D][4][CT];
    static_for<0, D - 1>([&](auto d) __attribute__((always_inline)) { load(xa[d], av[d], rb + 16 * d); });
    for (long r = rb; r < re; r += 16 * D)
        static_for<0, D>([&](auto d) __attribute__((always_inline)) {
            load(xa[(d + D - 1) % D], av[(d + D - 1) % D], r + 16 * (d + D - 1));
            __builtin_amdgcn_sched_barrier(0);
            tile(xa[d], av[d], r + 16 * d);
        });
}

// NN form, column strips: H[j][c] *= S[j][c] / ((G H)[j][c] + eps) for the wave's 16 columns and ALL rows j (dist_nmf.py:750-751);
// the wave reads its columns of H into registers before it writes them.  clamp: H = max(H, eps) afterwards (pyDNMF.py:156)
template <int KS>
__global__ __launch_bounds__(256) void f64_upd_h_kernel(double* __restrict__ H, int k, long n, long ldh, const double* __restrict__ Sm, long lds_,
                                                        const double* __restrict__ G, long ldg, double eps, int clamp) {
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const long c0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
    if (c0 >= n) return;
    const bool cok = c0 + i < n;
    double hb[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) hb[s] = (cok && 4 * s + q < k) ? H[(long)(4 * s + q) * ldh + c0 + i] : 0.0;
    for (int jt = 0; 16 * jt < k; ++jt) {
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const double a = (16 * jt + i < k && 4 * s + q < k) ? G[(long)(16 * jt + i) * ldg + 4 * s + q] : 0.0;
            acc = MFMA64(a, hb[s], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * jt + q + 4 * r;
            if (j < k && cok) {
                double v = H[(long)j * ldh + c0 + i] * (Sm[(long)j * lds_ + c0 + i] / (acc[r] + eps));
                if (clamp) v = v > eps ? v : eps;
                H[(long)j * ldh + c0 + i] = v;
            }
        }
    }
}

// ================================================================================================ streaming kernels
enum { E_CLAMP = 0, E_COLS_DIV = 1, E_ROWS_MUL = 2, E_KL_BYROW = 3, E_KL_BYCOL = 4 };
__global__ __launch_bounds__(256) void f64_ew_kernel(int op, double* __restrict__ X, long rows, long cols, long ldx, const double* __restrict__ Sm,
                                                     long lds_, const double* __restrict__ x, double eps, int clamp) {
    const long total = rows * cols;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / cols, c = idx % cols;
        double v = X[r * ldx + c];
        if (op == E_CLAMP) v = v > eps ? v : eps;                                  // np.maximum(X, eps), pyDNMF.py:156
        else if (op == E_COLS_DIV) v = v / (x[c] + eps);                          // pyDNMF.py:192
        else if (op == E_ROWS_MUL) v = v * x[r];                                  // pyDNMF.py:193
        else if (op == E_KL_BYROW) v = v * (Sm[r * lds_ + c] / (x[r] + eps));      // dist_nmf.py:847-849
        else v = v * (Sm[r * lds_ + c] / (x[c] + eps));                            // dist_nmf.py:828-830
        if (clamp && op != E_CLAMP) v = v > eps ? v : eps;
        X[r * ldx + c] = v;
    }
}

__device__ __forceinline__ double block_sum(double v, double* red) {       // fixed-order tree: deterministic
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double out = red[0];
    __syncthreads();
    return out;
}

// part[blockIdx.x] = sum over the block's grid-stride share of f(X); SQ: squares
template <bool SQ>
__global__ __launch_bounds__(256) void f64_sum_partial_kernel(const double* __restrict__ X, long rows, long cols, long ldx, double* __restrict__ part) {
    __shared__ double red[256];
    const long total = rows * cols;
    double acc = 0.0;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const double v = X[(idx / cols) * ldx + idx % cols];
        acc += SQ ? v * v : v;
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void f64_sum_final_kernel(const double* __restrict__ part, int nparts, double* __restrict__ out) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 256) acc += part[p];
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) *out = s;
}

// x[r] = sum_c H[r][c]: one workgroup per row
__global__ __launch_bounds__(256) void f64_rowsum_kernel(const double* __restrict__ H, long n, long ldh, double* __restrict__ x) {
    __shared__ double red[256];
    const double* row = H + (long)blockIdx.x * ldh;
    double acc = 0.0;
    for (long c = threadIdx.x; c < n; c += 256) acc += row[c];
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) x[blockIdx.x] = s;
}

// P[chunk][c] = sum_{r in chunk} f(X[r][c]) (SQ: squares): a thread per column, coalesced across the row
template <bool SQ>
__global__ __launch_bounds__(256) void f64_colsum_partial_kernel(const double* __restrict__ X, long m, long n, long ldx, long rows_per_chunk,
                                                                 double* __restrict__ P) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const long rb = (long)blockIdx.y * rows_per_chunk;
    const long re = rb + rows_per_chunk < m ? rb + rows_per_chunk : m;
    double acc = 0.0;
    for (long r = rb; r < re; ++r) { const double v = X[r * ldx + c]; acc += SQ ? v * v : v; }
    P[(long)blockIdx.y * n + c] = acc;
}

// ================================================================================================ HALS (dist_nmf.py:873-934)
// one column of the W sweep: pending normalisation of column kk - 1, then W[:, kk] = max(W[:, kk] G[kk][kk] + AH[:, kk] - W G[:, kk],
// eps) and the partial sums of its squares (utils.py:367-391)
__global__ __launch_bounds__(256) void f64_hals_w_col_kernel(double* __restrict__ W, long m, int k, long ldw, const double* __restrict__ AH,
                                                             long ldah, const double* __restrict__ G, long ldg, int kk,
                                                             const double* __restrict__ prev_ss2, double eps, double* __restrict__ part) {
    __shared__ double red[256];
    const double pn = (prev_ss2 && kk > 0) ? sqrt(*prev_ss2) : 0.0;
    double acc = 0.0;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < m; r += (long)gridDim.x * 256) {
        double* w = W + r * ldw;
        if (pn > 0.0) w[kk - 1] = w[kk - 1] / pn;
        double dot = 0.0;
        for (int l = 0; l < k; ++l) dot += w[l] * G[(long)l * ldg + kk];
        double v = w[kk] * G[(long)kk * ldg + kk] + AH[r * ldah + kk] - dot;
        v = v > eps ? v : eps;
        w[kk] = v;
        acc += v * v;
    }
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void f64_scale_col_kernel(double* __restrict__ W, long m, long ldw, int col, const double* __restrict__ ss2) {
    const double nrm = sqrt(*ss2);
    if (!(nrm > 0.0)) return;
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < m; r += (long)gridDim.x * 256) W[r * ldw + col] = W[r * ldw + col] / nrm;
}
// H sweep: rows in sequence, a thread per column (dist_nmf.py:905-909)
__global__ __launch_bounds__(256) void f64_hals_h_kernel(double* __restrict__ H, int k, long n, long ldh, const double* __restrict__ AtW,
                                                         long ldatw, const double* __restrict__ G, long ldg, double eps) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    for (int kk = 0; kk < k; ++kk) {
        double dot = 0.0;
        for (int l = 0; l < k; ++l) dot += G[(long)kk * ldg + l] * H[(long)l * ldh + c];
        const double v = H[(long)kk * ldh + c] + AtW[(long)kk * ldatw + c] - dot;
        H[(long)kk * ldh + c] = v > eps ? v : eps;
    }
}

// ------------------------------------------------------------------------------------------------ host side
bool vec_ok(const double* p, long ld) { return ((uintptr_t)p & 15) == 0 && ld % 2 == 0; }
// `rows` rows of pitch ld (plus one more row of slack for the lane offsets inside a row) stay inside a descriptor's 31-bit offsets
bool buf_ok(long ld, long rows) { return (rows + 1) * ld * 8 < 0x7fffffffL; }

struct TnPlan64 { int ncolblk; long nchunks, rows_per_chunk; };
TnPlan64 plan_tn64(long m, long n, int k) {
    TnPlan64 p;
    p.ncolblk = (int)cdiv(n, 16 * tn_ct(tiles16(k)));
    long nch = std::max<long>(1, 2048 / p.ncolblk);
    nch = std::min<long>(nch, std::max<long>(1, cdiv(m, 64)));
    p.rows_per_chunk = round_up(cdiv(m, nch), 16);
    p.nchunks = cdiv(m, p.rows_per_chunk);
    return p;
}
long nt_splits(long m, long n) {
    if (m <= 1024) return std::max<long>(1, std::min<long>(256, n / 512));
    const long forced = tune("DNMF_F64_NT_SPLIT", 0);
    return forced > 0 ? std::min<long>(forced, std::max<long>(1, n / 512)) : 1;
}

int sum_all(bool sq, const double* X, long rows, long cols, long ldx, double* out, void* ws, size_t ws_bytes, hipStream_t st) {
    const int nparts = (int)std::min<long>(1024, cdiv(rows * cols, 256 * 8));
    if (ws_bytes < (size_t)nparts * sizeof(double)) return fail(DNMF_EWS, "f64 sum: workspace too small");
    if (sq) hipLaunchKernelGGL(f64_sum_partial_kernel<true>, dim3(nparts), dim3(256), 0, st, X, rows, cols, ldx, (double*)ws);
    else hipLaunchKernelGGL(f64_sum_partial_kernel<false>, dim3(nparts), dim3(256), 0, st, X, rows, cols, ldx, (double*)ws);
    hipLaunchKernelGGL(f64_sum_final_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nparts, out);
    return check_launch("f64 sum");
}

int colsum_any(bool sq, const double* X, long m, long n, long ldx, double* x, void* ws, size_t ws_bytes, hipStream_t st) {
    const long nch = std::max<long>(1, std::min<long>(cdiv(m, 64), 2048 / std::max<long>(1, cdiv(n, 256))));
    const long rpc = cdiv(m, nch);
    const long chunks = cdiv(m, rpc);
    if (ws_bytes < (size_t)chunks * n * sizeof(double)) return fail(DNMF_EWS, "f64 colsum: workspace too small");
    const dim3 grid((unsigned)cdiv(n, 256), (unsigned)chunks);
    if (sq) hipLaunchKernelGGL(f64_colsum_partial_kernel<true>, grid, dim3(256), 0, st, X, m, n, ldx, rpc, (double*)ws);
    else hipLaunchKernelGGL(f64_colsum_partial_kernel<false>, grid, dim3(256), 0, st, X, m, n, ldx, rpc, (double*)ws);
    launch_reduce64((const double*)ws, n, n, (int)chunks, x, n, 1L, n, st);
    return check_launch("f64 colsum");
}

template <int MODE>
int launch_nn_rows(const double* X, long ldx, long m, int kc, const double* Y, long ldy, long n, const double* A, long lda, double* O, long ldo,
                   double eps, hipStream_t st) {
    if constexpr (MODE != NN_UPD_W) {
        const int ks = kc <= 16 ? 4 : kc <= 32 ? 8 : kc <= 64 ? 16 : 32;
        const int ncolblk = (int)cdiv(n, 16 * nn_ct(ks));
        long nch = std::max<long>(1, 2048 / ncolblk);
        nch = std::min<long>(nch, std::max<long>(1, cdiv(m, 64)));
        const long rpw = round_up(cdiv(m, nch), 16);
        const long nwaves = cdiv(m, rpw) * ncolblk;
        if (!(buf_ok(ldx, 16) && buf_ok(ldy, kc) && buf_ok(lda, 16) && buf_ok(ldo, 16))) return fail(DNMF_EINVAL, "f64 nn(cols): a row pitch beyond the 2 GiB window of the float64 operand loads");
        const bool vec = n % nn_ct(ks) == 0 && kc % 4 == 0;
#define NNC_CASE(KS_) do { if (vec) hipLaunchKernelGGL((f64_nn_cols_kernel<KS_, MODE, true>), dim3((unsigned)cdiv(nwaves, 4)), dim3(256), 0, st, X, ldx, m, kc, Y, ldy, n, \
                                                       A, lda, O, ldo, eps, rpw, ncolblk, nwaves); \
                           else hipLaunchKernelGGL((f64_nn_cols_kernel<KS_, MODE, false>), dim3((unsigned)cdiv(nwaves, 4)), dim3(256), 0, st, X, ldx, m, kc, Y, ldy, n, \
                                                   A, lda, O, ldo, eps, rpw, ncolblk, nwaves); } while (0)
        if (ks == 4) NNC_CASE(4); else if (ks == 8) NNC_CASE(8); else if (ks == 16) NNC_CASE(16); else NNC_CASE(32);
#undef NNC_CASE
        return check_launch("f64 nn(cols)");
    }
    // column range per wave: the whole width for the in-place W update (the wave owns its rows), else about 8192 waves in all
    long cpw = n;
    if (MODE != NN_UPD_W) {
        const long rowtiles = cdiv(m, 16);
        const long want = std::max<long>(1, 8192 / rowtiles);
        cpw = std::max<long>(64, round_up(cdiv(n, want), 16));
    }
    const dim3 grid((unsigned)cdiv(cdiv(m, 16), 4), (unsigned)cdiv(n, cpw));
#define NN_CASE(KS_) hipLaunchKernelGGL((f64_nn_rows_kernel<KS_, MODE>), grid, dim3(256), 0, st, X, ldx, m, kc, Y, ldy, n, A, lda, O, ldo, eps, cpw)
    if (kc <= 16) NN_CASE(4); else if (kc <= 32) NN_CASE(8); else if (kc <= 64) NN_CASE(16); else NN_CASE(32);
#undef NN_CASE
    return check_launch("f64 nn");
}

#include "dnmf_f64_kl.h"
inline bool kl64_buf_ok(long lda, long ldw, long ldh, int k) { return buf_ok(lda, 16) && buf_ok(ldw, 16) && buf_ok(ldh, k); }

}  // namespace

extern "C" {

// scratch for any f64 entry point on an m x n block: the partial slabs of the TN / NT forms and of the reductions; the m x n
// quotient / residual image the KL products and the error evaluation go through is a separate buffer of the caller (U)
size_t dnmf_f64_ws_bytes(long m, long n, int k) {
    if (k < 1 || k > DNMF_TUNED_MAX_K || m < 1 || n < 1) return 0;
    const size_t kp = 16 * tiles16(k), D = sizeof(double);
    auto colsum_slabs = [](long rows, long cols) {
        return (size_t)std::max<long>(1, std::min<long>(cdiv(rows, 64), 2048 / std::max<long>(1, cdiv(cols, 256)))) * cols;
    };
    size_t b = 2048 * D;                                                               // block partials of the sums
    b = std::max(b, (size_t)plan_tn64(m, n, k).nchunks * kp * round_up(n, 16) * D);       // W^T A
    b = std::max(b, (size_t)plan_tn64(m, k, k).nchunks * kp * round_up(k, 16) * D);       // W^T W
    b = std::max(b, (size_t)nt_splits(k, n) * k * kp * D);                             // H H^T (column splits)
    if (nt_splits(m, n) > 1) b = std::max(b, (size_t)nt_splits(m, n) * m * kp * D);    // A H^T of a short A
    b = std::max(b, colsum_slabs(m, n) * D);                                           // column sums of an m x n image
    b = std::max(b, colsum_slabs(m, k) * D);                                           // column sums of W
    b = std::max(b, kl64_ws_bytes(m, n, k));                                           // the fused KL products (partial slabs)
    return al256(b) + 256;
}

// C[m x kc] = X[m x n] Y[kc x n]^T  (A H^T: global_mm(A, H.T), dist_nmf.py:730; H H^T: global_gram(H.T), :729 -- X = Y = H)
int dnmf_f64_aht(const double* X, long m, long n, long ldx, const double* Y, int kc, long ldy, double* C, long ldc, void* ws, size_t ws_bytes,
                 void* stream) {
    REQ(X && Y && C && m >= 1 && n >= 1 && kc >= 1 && kc <= DNMF_TUNED_MAX_K && ldx >= n && ldy >= n && ldc >= kc, "f64 aht: bad arguments");
    hipStream_t st = ST(stream);
    const long ns = nt_splits(m, n);
    const long cps = ns > 1 ? round_up(cdiv(n, ns), 16) : round_up(n, 16);
    const long nsplit = cdiv(n, cps);
    const int kp = 16 * tiles16(kc);
    double* out = C; long ldo = ldc, sstride = 0;
    if (nsplit > 1) {
        if (!ws || ws_bytes < (size_t)nsplit * m * kp * sizeof(double)) return fail(DNMF_EWS, "f64 aht: workspace too small");
        out = (double*)ws; ldo = kp; sstride = m * kp;
    }
    REQ(buf_ok(ldx, 16) && buf_ok(ldy, kc), "f64 aht: a row pitch beyond the 2 GiB window of the float64 operand loads");
    const bool vec = n % 4 == 0;
    const int nt = tiles16(kc);
    int rt = (m >= 16 * 4 * 256 && nt <= 4) ? 4 : (m >= 16 * 2 * 256 ? 2 : 1);     // row tiles per wave: 16 accumulators at most
    if (tune("DNMF_F64_NT_RT", 0)) rt = (int)tune("DNMF_F64_NT_RT", 0);
    while (rt > 1 && !buf_ok(ldx, 16 * rt)) rt >>= 1;
// (three register sets in flight for k <= 16, where the kernel is a stream of A: 0.565 -> 0.509 ms at 65536 x 4096; two beyond)
#define NT_LAUNCH(RT_, NT_, VEC_) hipLaunchKernelGGL((f64_nt_kernel<RT_, NT_, VEC_, (NT_ == 1 ? 3 : 2)>), dim3((unsigned)cdiv(cdiv(m, 16 * RT_), 4), (unsigned)nsplit), dim3(256), 0, st, \
                                                     X, ldx, m, n, Y, ldy, kc, out, ldo, sstride, cps)
#define NT_CASE(RT_, NT_) do { if (vec) NT_LAUNCH(RT_, NT_, true); else NT_LAUNCH(RT_, NT_, false); } while (0)
#define NT_ROWS(NT_) do { if (rt == 4) NT_CASE(4, NT_); else if (rt == 2) NT_CASE(2, NT_); else NT_CASE(1, NT_); } while (0)
    if (nt <= 1) NT_ROWS(1); else if (nt <= 2) NT_ROWS(2); else if (nt <= 4) NT_ROWS(4); else if (rt >= 2) NT_CASE(2, 8); else NT_CASE(1, 8);
#undef NT_ROWS
#undef NT_CASE
#undef NT_LAUNCH
    int rc = check_launch("f64 aht");
    if (rc || nsplit == 1) return rc;
    launch_reduce64((const double*)ws, sstride, ldo, (int)nsplit, C, ldc, m, (long)kc, st);
    return check_launch("f64 aht(reduce)");
}

// C[kc x n] = X[m x kc]^T Y[m x n]  (W^T A: global_mm(W.T, A), dist_nmf.py:749; W^T W: global_gram(W), :748 -- Y = X = W)
int dnmf_f64_wta(const double* Y, long m, long n, long ldy, const double* X, int kc, long ldx, double* C, long ldc, void* ws, size_t ws_bytes,
                 void* stream) {
    REQ(X && Y && C && ws && m >= 1 && n >= 1 && kc >= 1 && kc <= DNMF_TUNED_MAX_K && ldy >= n && ldx >= kc && ldc >= n, "f64 wta: bad arguments");
    hipStream_t st = ST(stream);
    const TnPlan64 p = plan_tn64(m, n, kc);
    const int kp = 16 * tiles16(kc);
    const long ldp = round_up(n, 16);
    if (ws_bytes < (size_t)p.nchunks * kp * ldp * sizeof(double)) return fail(DNMF_EWS, "f64 wta: workspace too small");
    const long nwaves = p.nchunks * p.ncolblk;
    const dim3 grid((unsigned)cdiv(nwaves, 4));
    REQ(buf_ok(ldx, 16) && buf_ok(ldy, 16), "f64 wta: a row pitch beyond the 2 GiB window of the float64 operand loads");
    const int nt = tiles16(kc);
    const int ntk = nt <= 1 ? 1 : nt <= 2 ? 2 : nt <= 4 ? 4 : 8;                    // the instantiation below
    const bool vec = n % tn_ct(ntk) == 0 && kc % std::min(ntk, 4) == 0;
#define TN_CASE(NT_) do { if (vec) hipLaunchKernelGGL((f64_tn_kernel<NT_, true>), grid, dim3(256), 0, st, X, ldx, kc, Y, ldy, n, m, p.rows_per_chunk, p.ncolblk, nwaves, \
                                                      (double*)ws, (long)kp * ldp, ldp); \
                          else hipLaunchKernelGGL((f64_tn_kernel<NT_, false>), grid, dim3(256), 0, st, X, ldx, kc, Y, ldy, n, m, p.rows_per_chunk, p.ncolblk, nwaves, \
                                                  (double*)ws, (long)kp * ldp, ldp); } while (0)
    if (nt <= 1) TN_CASE(1); else if (nt <= 2) TN_CASE(2); else if (nt <= 4) TN_CASE(4); else TN_CASE(8);
#undef TN_CASE
    int rc = check_launch("f64 wta");
    if (rc) return rc;
    launch_reduce64((const double*)ws, (long)kp * ldp, ldp, (int)p.nchunks, C, ldc, (long)kc, n, st);
    return check_launch("f64 wta(reduce)");
}

// W *= AH / (W G + eps)  (dist_nmf.py:731-732, :244-245)
int dnmf_f64_mu_update_w(double* W, long m, int k, long ldw, const double* AH, long ldah, const double* G, long ldg, double eps, void* stream) {
    REQ(W && AH && G && m >= 1 && k >= 1 && k <= DNMF_TUNED_MAX_K && ldw >= k && ldah >= k && ldg >= k, "f64 mu_update_w: bad arguments");
    return launch_nn_rows<NN_UPD_W>(W, ldw, m, k, G, ldg, k, AH, ldah, W, ldw, eps, ST(stream));
}

// H *= AtW / (G H + eps), clamp: H = max(H, eps) afterwards  (dist_nmf.py:750-751, :224-225; pyDNMF.py:156)
int dnmf_f64_mu_update_h(double* H, int k, long n, long ldh, const double* AtW, long ldatw, const double* G, long ldg, double eps, int clamp,
                         void* stream) {
    REQ(H && AtW && G && n >= 1 && k >= 1 && k <= DNMF_TUNED_MAX_K && ldh >= n && ldatw >= n && ldg >= k, "f64 mu_update_h: bad arguments");
    hipStream_t st = ST(stream);
    const dim3 grid((unsigned)cdiv(cdiv(n, 16), 4));
#define UH_CASE(KS_) hipLaunchKernelGGL((f64_upd_h_kernel<KS_>), grid, dim3(256), 0, st, H, k, n, ldh, AtW, ldatw, G, ldg, eps, clamp)
    if (k <= 16) UH_CASE(4); else if (k <= 32) UH_CASE(8); else if (k <= 64) UH_CASE(16); else UH_CASE(32);
#undef UH_CASE
    return check_launch("f64 mu_update_h");
}

// U[m x n] = A / (W H + eps)  (the KL quotient, dist_nmf.py:806; the reference materialises it too)
int dnmf_f64_kl_quot(const double* A, long m, long n, long lda, const double* W, long ldw, const double* H, long ldh, int k, double eps,
                     double* U, long ldu, void* stream) {
    REQ(A && W && H && U && m >= 1 && n >= 1 && k >= 1 && k <= DNMF_TUNED_MAX_K && lda >= n && ldw >= k && ldh >= n && ldu >= n, "f64 kl_quot: bad arguments");
    return launch_nn_rows<NN_QUOT>(W, ldw, m, k, H, ldh, n, A, lda, U, ldu, eps, ST(stream));
}

// S[m x k] = (A / (W H + eps)) H^T  (dist_nmf.py:806, :810) -- k <= 64: one pass over A, the quotient never leaves the registers
// (csrc/dnmf_f64_kl.h); beyond: through the image U (m x n, the caller's; DNMF_EINVAL without one)
int dnmf_f64_kl_uht(const double* A, long m, long n, long lda, const double* W, long ldw, const double* H, long ldh, int k, double eps,
                    double* S, long lds_, double* U, void* ws, size_t ws_bytes, void* stream) {
    REQ(A && W && H && S && m >= 1 && n >= 1 && k >= 1 && k <= DNMF_TUNED_MAX_K && lda >= n && ldw >= k && ldh >= n && lds_ >= k, "f64 kl_uht: bad arguments");
    if (k > KL64_MAX_K || tune("DNMF_F64_KL_IMAGE", 0) || !kl64_buf_ok(lda, ldw, ldh, k)) {
        REQ(U, "f64 kl_uht: k = %d > %d needs the m x n image U", k, KL64_MAX_K);
        int rc = dnmf_f64_kl_quot(A, m, n, lda, W, ldw, H, ldh, k, eps, U, n, stream);
        return rc ? rc : dnmf_f64_aht(U, m, n, n, H, k, ldh, S, lds_, ws, ws_bytes, stream);
    }
    hipStream_t st = ST(stream);
    KlUhtPlan p = plan_kl_uht(m, n, k);
    if (tune("DNMF_F64_KL_RT", 0)) { p.rt = (int)tune("DNMF_F64_KL_RT", 0); }
    const int kp = 16 * tiles16(k);
    double* out = S; long ldo = lds_, sstride = 0;
    if (p.nsplit > 1) {
        if (!ws || ws_bytes < (size_t)p.nsplit * m * kp * sizeof(double)) return fail(DNMF_EWS, "f64 kl_uht: workspace too small");
        out = (double*)ws; ldo = kp; sstride = m * kp;
    }
    const bool vec = n % 4 == 0;
    while (p.rt > 1 && !(buf_ok(lda, 16 * p.rt) && buf_ok(ldw, 16 * p.rt))) p.rt >>= 1;
    const bool full = n % p.cps == 0 && p.cps % 64 == 0;          // every split a whole number of groups of four tiles: the mask-free kernel
#define UHT_LAUNCH(NT_, RT_, VEC_, FULL_) hipLaunchKernelGGL((f64_kl_uht_kernel<NT_, RT_, VEC_, FULL_>), dim3((unsigned)cdiv(cdiv(m, 16 * RT_), 4), (unsigned)p.nsplit), dim3(256), 0, st, \
                                                             A, lda, m, n, W, ldw, H, ldh, k, eps, out, ldo, sstride, p.cps)
#define UHT_CASE(NT_, RT_) do { if (full) UHT_LAUNCH(NT_, RT_, true, true); else if (vec) UHT_LAUNCH(NT_, RT_, true, false); else UHT_LAUNCH(NT_, RT_, false, false); } while (0)
    const int nt = tiles16(k);
    if (nt <= 2 && p.rt >= 4) { if (nt <= 1) UHT_CASE(1, 4); else UHT_CASE(2, 4); }
    else if (p.rt >= 2) { if (nt <= 1) UHT_CASE(1, 2); else if (nt <= 2) UHT_CASE(2, 2); else UHT_CASE(4, 2); }
    else { if (nt <= 1) UHT_CASE(1, 1); else if (nt <= 2) UHT_CASE(2, 1); else UHT_CASE(4, 1); }
#undef UHT_LAUNCH
#undef UHT_CASE
    int rc = check_launch("f64 kl_uht");
    if (rc || p.nsplit == 1) return rc;
    launch_reduce64((const double*)ws, sstride, ldo, (int)p.nsplit, S, lds_, m, (long)k, st);
    return check_launch("f64 kl_uht(reduce)");
}

// S[k x n] = W^T (A / (W H + eps))  (dist_nmf.py:806, :808) -- as dnmf_f64_kl_uht
int dnmf_f64_kl_wtu(const double* A, long m, long n, long lda, const double* W, long ldw, const double* H, long ldh, int k, double eps,
                    double* S, long lds_, double* U, void* ws, size_t ws_bytes, void* stream) {
    REQ(A && W && H && S && ws && m >= 1 && n >= 1 && k >= 1 && k <= DNMF_TUNED_MAX_K && lda >= n && ldw >= k && ldh >= n && lds_ >= n, "f64 kl_wtu: bad arguments");
    if (k > KL64_MAX_K || tune("DNMF_F64_KL_IMAGE", 0) || !kl64_buf_ok(lda, ldw, ldh, k)) {
        REQ(U, "f64 kl_wtu: k = %d > %d needs the m x n image U", k, KL64_MAX_K);
        int rc = dnmf_f64_kl_quot(A, m, n, lda, W, ldw, H, ldh, k, eps, U, n, stream);
        return rc ? rc : dnmf_f64_wta(U, m, n, n, W, k, ldw, S, lds_, ws, ws_bytes, stream);
    }
    hipStream_t st = ST(stream);
    const KlWtuPlan p = plan_kl_wtu(m, n, k);
    const int kp = 16 * tiles16(k);
    const long ldp = round_up(n, 16);
    if (ws_bytes < (size_t)p.nchunks * kp * ldp * sizeof(double)) return fail(DNMF_EWS, "f64 kl_wtu: workspace too small");
    const long nwaves = p.nchunks * p.ncolblk;
    const dim3 grid((unsigned)cdiv(nwaves, 4));
    const int nt = tiles16(k);
    const bool vec = n % p.ct == 0 && k % 4 == 0;
#define WTU_CASE(NT_) do { if (vec) hipLaunchKernelGGL((f64_kl_wtu_kernel<NT_, kl_ct(NT_), true>), grid, dim3(256), 0, st, A, lda, m, n, W, ldw, H, ldh, k, eps, \
                                                       p.rows_per_chunk, p.ncolblk, nwaves, (double*)ws, (long)kp * ldp, ldp); \
                           else hipLaunchKernelGGL((f64_kl_wtu_kernel<NT_, kl_ct(NT_), false>), grid, dim3(256), 0, st, A, lda, m, n, W, ldw, H, ldh, k, eps, \
                                                   p.rows_per_chunk, p.ncolblk, nwaves, (double*)ws, (long)kp * ldp, ldp); } while (0)
    if (nt <= 1) WTU_CASE(1); else if (nt <= 2) WTU_CASE(2); else WTU_CASE(4);
#undef WTU_CASE
    int rc = check_launch("f64 kl_wtu");
    if (rc) return rc;
    launch_reduce64((const double*)ws, (long)kp * ldp, ldp, (int)p.nchunks, S, lds_, (long)k, n, st);
    return check_launch("f64 kl_wtu(reduce)");
}

// R[m x n] = (A - W H)^2 element-wise  (pyDNMF.py:207, :229: the caller sums it -- dnmf_f64_sum / dnmf_f64_colsum)
int dnmf_f64_sqdiff(const double* A, long m, long n, long lda, const double* W, long ldw, const double* H, long ldh, int k, double* R, long ldr,
                    void* stream) {
    REQ(A && W && H && R && m >= 1 && n >= 1 && k >= 1 && k <= DNMF_TUNED_MAX_K && lda >= n && ldw >= k && ldh >= n && ldr >= n, "f64 sqdiff: bad arguments");
    return launch_nn_rows<NN_SQDIFF>(W, ldw, m, k, H, ldh, n, A, lda, R, ldr, 0.0, ST(stream));
}

// *out = sum X (sq == 0) or sum X^2 (sq != 0) over a rows x cols matrix, fixed summation order
int dnmf_f64_sum(const double* X, long rows, long cols, long ldx, int sq, double* out, void* ws, size_t ws_bytes, void* stream) {
    REQ(X && out && ws && rows >= 1 && cols >= 1 && ldx >= cols, "f64 sum: bad arguments");
    return sum_all(sq != 0, X, rows, cols, ldx, out, ws, ws_bytes, ST(stream));
}

// x[c] = sum_r X[r][c] (or of squares): sum_along_axis(W, axis=0), dist_nmf.py:793; the per-column sums of column_err, pyDNMF.py:229
int dnmf_f64_colsum(const double* X, long m, long n, long ldx, int sq, double* x, void* ws, size_t ws_bytes, void* stream) {
    REQ(X && x && ws && m >= 1 && n >= 1 && ldx >= n, "f64 colsum: bad arguments");
    return colsum_any(sq != 0, X, m, n, ldx, x, ws, ws_bytes, ST(stream));
}

// x[r] = sum_c H[r][c]  (sum_along_axis(H, axis=1), dist_nmf.py:793-795)
int dnmf_f64_rowsum(const double* H, int k, long n, long ldh, double* x, void* stream) {
    REQ(H && x && k >= 1 && n >= 1 && ldh >= n, "f64 rowsum: bad arguments");
    hipLaunchKernelGGL(f64_rowsum_kernel, dim3(k), dim3(256), 0, ST(stream), H, n, ldh, x);
    return check_launch("f64 rowsum");
}

// op 0: X = max(X, eps); 1: X[r][c] /= x[c] + eps; 2: X[r][c] *= x[r]; 3: X[r][c] *= S[r][c] / (x[r] + eps); 4: ... / (x[c] + eps);
// clamp (ops 3, 4): max(., eps) afterwards
int dnmf_f64_ew(int op, double* X, long rows, long cols, long ldx, const double* Sm, long lds_, const double* x, double eps, int clamp,
                void* stream) {
    REQ(X && rows >= 1 && cols >= 1 && ldx >= cols && op >= 0 && op <= 4 && (op == E_CLAMP || x) && (op < E_KL_BYROW || (Sm && lds_ >= cols)),
        "f64 ew: bad arguments");
    const unsigned grid = (unsigned)std::min<long>(cdiv(rows * cols, 256), 8192);
    hipLaunchKernelGGL(f64_ew_kernel, dim3(grid), dim3(256), 0, ST(stream), op, X, rows, cols, ldx, Sm, lds_, x, eps, clamp);
    return check_launch("f64 ew");
}

// one column of the HALS W sweep (dist_nmf.py:886-887): see dnmf_hals_w_col; *ss2_out = sum of squares of the new column
int dnmf_f64_hals_w_col(double* W, long m, int k, long ldw, const double* AH, long ldah, const double* G, long ldg, int kk,
                        const double* prev_ss2, double eps, double* ss2_out, void* ws, size_t ws_bytes, void* stream) {
    REQ(W && AH && G && ss2_out && ws && m >= 1 && k >= 1 && k <= DNMF_TUNED_MAX_K && kk >= 0 && kk < k && ldw >= k && ldah >= k && ldg >= k,
        "f64 hals_w_col: bad arguments");
    hipStream_t st = ST(stream);
    const int nb = (int)std::min<long>(cdiv(m, 256), 1024);
    if (ws_bytes < (size_t)nb * sizeof(double)) return fail(DNMF_EWS, "f64 hals_w_col: workspace too small");
    hipLaunchKernelGGL(f64_hals_w_col_kernel, dim3(nb), dim3(256), 0, st, W, m, k, ldw, AH, ldah, G, ldg, kk, prev_ss2, eps, (double*)ws);
    hipLaunchKernelGGL(f64_sum_final_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, nb, ss2_out);
    return check_launch("f64 hals_w_col");
}
// W[:, col] /= sqrt(*ss2) if > 0 (dist_nmf.py:890-891)
int dnmf_f64_hals_w_scale(double* W, long m, long ldw, int col, const double* ss2, void* stream) {
    REQ(W && ss2 && m >= 1 && col >= 0 && ldw > col, "f64 hals_w_scale: bad arguments");
    hipLaunchKernelGGL(f64_scale_col_kernel, dim3((unsigned)std::min<long>(cdiv(m, 256), 2048)), dim3(256), 0, ST(stream), W, m, ldw, col, ss2);
    return check_launch("f64 hals_w_scale");
}
// H sweep (dist_nmf.py:905-909)
int dnmf_f64_hals_update_h(double* H, int k, long n, long ldh, const double* AtW, long ldatw, const double* G, long ldg, double eps, void* stream) {
    REQ(H && AtW && G && n >= 1 && k >= 1 && k <= DNMF_TUNED_MAX_K && ldh >= n && ldatw >= n && ldg >= k, "f64 hals_update_h: bad arguments");
    hipLaunchKernelGGL(f64_hals_h_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, ST(stream), H, k, n, ldh, AtW, ldatw, G, ldg, eps);
    return check_launch("f64 hals_update_h");
}


// ---- whole fits in float64 on one rank (PyNMF.fit, pyDNMF.py:138-182 with p_r = p_c = 1): the primitives above in the order the
// choreography (pydnmfk_amd/dist_nmf.py over engine.HipOpsF64) issues them -- same kernels, same operands, so the same bits -- without a
// Python frame and a ctypes call per launch (about 15 per step: 150-200 us of host time against 60-80 us of kernels on the reference's own
// test sizes).  method: 0 = MU/FRO, 1 = MU/KL, 2 = HALS/FRO.  sq_out (device): {sum (A - W H)^2, sum A^2}.
}  // extern "C"
__attribute__((visibility("hidden"))) int dnmf_f64_tiny_fit_(int method, const double* A, long m, long n, long lda, long a_stride, double* W, long ldw,
                                                              long w_stride, double* H, long ldh, long h_stride, int k, double eps, int w_update, int itr,
                                                              int batch, double* sq_out, long sq_stride, void* stream);
extern "C" {
size_t dnmf_f64_ws_bytes_fit(long m, long n, int k) {
    const size_t prim = dnmf_f64_ws_bytes(m, n, k);
    if (!prim) return 0;
    const size_t D = sizeof(double);
    return al256(prim) + al256((size_t)k * k * D) + al256((size_t)std::max(m * k, (long)k * n) * D) + 2 * al256((size_t)k * D) + al256((size_t)m * n * D) + 256;
}

int dnmf_f64_fit(int method, const double* A, long m, long n, long lda, double* W, long ldw, double* H, long ldh, int k, double eps, int w_update,
                 int itr, double* sq_out, void* ws, size_t ws_bytes, void* stream) {
    REQ(method >= 0 && method <= 2 && A && W && H && sq_out && ws && m >= 1 && n >= 1 && k >= 1 && k <= DNMF_TUNED_MAX_K && itr >= 0 && lda >= n &&
            ldw >= k && ldh >= n, "f64 fit: bad arguments");
    const size_t need = dnmf_f64_ws_bytes_fit(m, n, k);
    if (ws_bytes < need) return fail(DNMF_EWS, "f64 fit: workspace %zu < %zu", ws_bytes, need);
    static const bool tiny_on = tune("DNMF_F64_TINY", 1) != 0;     // (tuning build: 0 = the chain of primitives, for the before / after timing)
    if (itr >= 1 && tiny_on) {      // tiny problems (the reference's own test sizes): the whole fit as ONE single-workgroup launch (csrc/dnmf_f64_tiny.hip)
        const int rc_tiny = dnmf_f64_tiny_fit_(method, A, m, n, lda, 0, W, ldw, 0, H, ldh, 0, k, eps, w_update, itr, 1, sq_out, 2, stream);
        if (rc_tiny != 1) return rc_tiny;
    }
    const size_t D = sizeof(double), prim_bytes = dnmf_f64_ws_bytes(m, n, k);
    char* b = (char*)ws;
    void* prim = b;                                   b += al256(prim_bytes);
    double* G = (double*)b;                           b += al256((size_t)k * k * D);
    double* S = (double*)b;                           b += al256((size_t)std::max(m * k, (long)k * n) * D);
    double* x = (double*)b;                           b += al256((size_t)k * D);
    double* ss2 = (double*)b;                         b += al256((size_t)k * D);
    double* U = (double*)b;                           b += al256((size_t)m * n * D);
    double* sq = (double*)b;
    int rc = DNMF_OK;
#define F64(call) do { if ((rc = (call))) return rc; } while (0)
    for (int i = 0; i < itr; ++i) {
        const int clamp = (i % 10 == 0);                                              // pyDNMF.py:155 / :170
        if (method == 0) {                                                            // dist_nmf.py:716-751
            if (w_update) {
                F64(dnmf_f64_aht(H, k, n, ldh, H, k, ldh, G, k, prim, prim_bytes, stream));
                F64(dnmf_f64_aht(A, m, n, lda, H, k, ldh, S, k, prim, prim_bytes, stream));
                F64(dnmf_f64_mu_update_w(W, m, k, ldw, S, k, G, k, eps, stream));
            }
            F64(dnmf_f64_wta(W, m, k, ldw, W, k, ldw, G, k, prim, prim_bytes, stream));
            F64(dnmf_f64_wta(A, m, n, lda, W, k, ldw, S, n, prim, prim_bytes, stream));
            F64(dnmf_f64_mu_update_h(H, k, n, ldh, S, n, G, k, eps, clamp, stream));
            if (clamp) F64(dnmf_f64_ew(0, W, m, k, ldw, nullptr, 0, nullptr, eps, 0, stream));
        } else if (method == 1) {                                                     // dist_nmf.py:806-849
            if (w_update) {
                F64(dnmf_f64_rowsum(H, k, n, ldh, x, stream));
                F64(dnmf_f64_kl_uht(A, m, n, lda, W, ldw, H, ldh, k, eps, S, k, U, prim, prim_bytes, stream));
                F64(dnmf_f64_ew(4, W, m, k, ldw, S, k, x, eps, 0, stream));
            }
            F64(dnmf_f64_colsum(W, m, k, ldw, 0, x, prim, prim_bytes, stream));
            F64(dnmf_f64_kl_wtu(A, m, n, lda, W, ldw, H, ldh, k, eps, S, n, U, prim, prim_bytes, stream));
            F64(dnmf_f64_ew(3, H, k, n, ldh, S, n, x, eps, clamp, stream));
            if (clamp) F64(dnmf_f64_ew(0, W, m, k, ldw, nullptr, 0, nullptr, eps, 0, stream));
        } else {                                                                      // dist_nmf.py:873-934
            if (w_update) {
                F64(dnmf_f64_aht(H, k, n, ldh, H, k, ldh, G, k, prim, prim_bytes, stream));
                F64(dnmf_f64_aht(A, m, n, lda, H, k, ldh, S, k, prim, prim_bytes, stream));
                if (hipMemsetAsync(ss2, 0, (size_t)k * D, ST(stream)) != hipSuccess) return fail(DNMF_EHIP, "f64 fit: memset failed");
                for (int kk = 0; kk < k; ++kk)
                    F64(dnmf_f64_hals_w_col(W, m, k, ldw, S, k, G, k, kk, kk > 0 ? ss2 + kk - 1 : nullptr, eps, ss2 + kk, prim, prim_bytes, stream));
                F64(dnmf_f64_hals_w_scale(W, m, ldw, k - 1, ss2 + k - 1, stream));
            }
            F64(dnmf_f64_wta(W, m, k, ldw, W, k, ldw, G, k, prim, prim_bytes, stream));
            F64(dnmf_f64_wta(A, m, n, lda, W, k, ldw, S, n, prim, prim_bytes, stream));
            F64(dnmf_f64_hals_update_h(H, k, n, ldh, S, n, G, k, eps, stream));
            if (clamp) {
                F64(dnmf_f64_ew(0, H, k, n, ldh, nullptr, 0, nullptr, eps, 0, stream));
                F64(dnmf_f64_ew(0, W, m, k, ldw, nullptr, 0, nullptr, eps, 0, stream));
            }
        }
    }
    // normalize_features (pyDNMF.py:185-194), then the two squared norms of relative_err (:205-218)
    F64(dnmf_f64_colsum(W, m, k, ldw, 0, x, prim, prim_bytes, stream));
    F64(dnmf_f64_ew(1, W, m, k, ldw, nullptr, 0, x, eps, 0, stream));
    F64(dnmf_f64_ew(2, H, k, n, ldh, nullptr, 0, x, 0.0, 0, stream));
    F64(dnmf_f64_sqdiff(A, m, n, lda, W, ldw, H, ldh, k, U, n, stream));
    F64(dnmf_f64_sum(U, m, n, n, 0, sq, prim, prim_bytes, stream));
    F64(dnmf_f64_sum(A, m, n, lda, 1, sq + 1, prim, prim_bytes, stream));
#undef F64
    if (hipMemcpyAsync(sq_out, sq, 2 * D, hipMemcpyDeviceToDevice, ST(stream)) != hipSuccess) return fail(DNMF_EHIP, "f64 fit: copy of the squared norms failed");
    return DNMF_OK;
}

}  // extern "C"
