// dnmf_stream.h -- streaming kernels: clamp, scale, KL element-wise update, norms, row / column sums.
// Part of libdnmf_hip.so (kernels live in anonymous namespaces of the headers; the translation units csrc/*.hip include what they launch).
#pragma once
#include "dnmf_common.h"

namespace {

// =============================================================================================== NMFk perturbation
// X_per = X * (1 + nv + 2 nv U), U ~ U[0,1) per element (reference pyDNMFk.py:42-44, `sample.randM`: the perturbed copy every
// one of the 20 fits per k of an NMFk sweep starts from).  ONE pass: 8 elements per thread are read in their storage type (fp32
// or bf16), scaled in fp32 and written back rounded once; the uniforms come from a counter-based generator -- the splitmix64
// finaliser of (seed, element-pair index), 24 bits per element -- so there is no state, no second buffer, and the result
// depends only on (seed, position).  Round 4: the torch expression this replaces made five passes over fp32 temporaries of
// the data's shape (about 9 GB of traffic per fit at 65536 x 4096 against 1 GB here).
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

template <typename TA>
__global__ __launch_bounds__(256) void perturb_uniform_kernel(const TA* __restrict__ X, TA* __restrict__ out, long rows, long cols,
                                                              long ldx, long ldo, float nv, unsigned long long seed) {
    const long cvecs = cols / 8;                            // 8 elements per thread (cols % 8 == 0: checked by the host)
    const long total = rows * cvecs;
    const unsigned long long key = mix64(seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / cvecs, c = (idx % cvecs) * 8;
        float v[8];
        load_vec_raw<4>(reinterpret_cast<float (&)[4]>(v[0]), X + r * ldx + c);
        load_vec_raw<4>(reinterpret_cast<float (&)[4]>(v[4]), X + r * ldx + c + 4);
        const unsigned long long base = ((unsigned long long)r * (unsigned long long)cols + (unsigned long long)c) >> 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long h = mix64(key + (base + q) * 0x9E3779B97F4A7C15ull);
            const float u0 = (float)((unsigned)(h >> 40)) * (1.0f / 16777216.0f);          // 24 bits each: [0, 1)
            const float u1 = (float)((unsigned)(h >> 8) & 0xffffffu) * (1.0f / 16777216.0f);
            v[2 * q] *= fmaf(2.0f * nv, u0, 1.0f + nv);
            v[2 * q + 1] *= fmaf(2.0f * nv, u1, 1.0f + nv);
        }
        if constexpr (std::is_same<TA, float>::value) {
            *reinterpret_cast<f32x4*>(out + r * ldo + c) = f32x4{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(out + r * ldo + c + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {                                            // round to nearest even, as torch's .to(bfloat16) does
            unsigned int w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned int lo = __float_as_uint(v[2 * q]), hi = __float_as_uint(v[2 * q + 1]);
                lo = (lo + 0x7fffu + ((lo >> 16) & 1u)) >> 16;
                hi = (hi + 0x7fffu + ((hi >> 16) & 1u)) & 0xffff0000u;
                w[q] = lo | hi;
            }
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<u32x4*>(out + r * ldo + c) = u32x4{w[0], w[1], w[2], w[3]};
        }
    }
}

// The same perturbation for ANY shape and alignment, one element per thread: element (r, c) has linear index L = r cols + c and
// takes the high (L even) or low (L odd) 24 bits of the hash of pair L >> 1 -- exactly the value the vector kernel above gives
// it, so which kernel runs never changes a result (one random stream per seed whatever the block's shape: ADVICE r04).
template <typename TA>
__global__ __launch_bounds__(256) void perturb_uniform_any_kernel(const TA* __restrict__ X, TA* __restrict__ out, long rows, long cols,
                                                                  long ldx, long ldo, float nv, unsigned long long seed) {
    const long total = rows * cols;
    const unsigned long long key = mix64(seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / cols, c = idx % cols;
        const unsigned long long h = mix64(key + ((unsigned long long)idx >> 1) * 0x9E3779B97F4A7C15ull);
        const float u = (idx & 1) ? (float)((unsigned)(h >> 8) & 0xffffffu) * (1.0f / 16777216.0f)
                                  : (float)((unsigned)(h >> 40)) * (1.0f / 16777216.0f);
        if constexpr (std::is_same<TA, float>::value) {
            out[r * ldo + c] = X[r * ldx + c] * fmaf(2.0f * nv, u, 1.0f + nv);
        } else {
            const float v = bf16_lo((unsigned int)X[r * ldx + c]) * fmaf(2.0f * nv, u, 1.0f + nv);
            const unsigned int w = __float_as_uint(v);
            out[r * ldo + c] = (bf16_t)((w + 0x7fffu + ((w >> 16) & 1u)) >> 16);
        }
    }
}

// =============================================================================================== element-wise passes
// clamp / column- and row-scaling / the KL multiply-divide: one read-modify-write pass over a rows x cols matrix, at most
// one streamed operand S beside it and a k-vector x indexed by row or by column.  HBM-bound, so what matters is 16-byte
// accesses, several of them in flight per lane, no per-element index arithmetic, and that a workgroup sweeps CONTIGUOUS
// memory (a first version walked each thread down the rows of a long H, 16 MB between consecutive accesses: 4.1-4.3 TB/s
// where a plain contiguous sweep reaches 5.8).  Two shapes:
//   LONG  (a long H: cols >= 1024): blockIdx.y = row, blockIdx.x = chunk of U x 256 vectors of that row (16 KiB);
//   short (a tall W: cols = k):     the 256 threads form a TX x TY patch (TX = 2^txs column vectors, 16 for k = 64), a
//          workgroup handles U consecutive patches = U x TY consecutive rows (16 KiB when ld == cols); the column-indexed
//          operand stays in registers.
enum { EW_CLAMP = 0, EW_COLS_DIV = 1, EW_ROWS_MUL = 2, EW_KL_BYROW = 3, EW_KL_BYCOL = 4 };

template <int OP, int V, bool LONG, bool NTP = false>
__global__ __launch_bounds__(256) void ew_kernel(float* __restrict__ X, long rows, long cols, long ldx,
                                                 const float* __restrict__ Sm, long lds_, const float* __restrict__ x,
                                                 float eps, int clamp, int txs, BatchTab bt) {
    REBASE(X); REBASE(Sm); REBASE(x);
    constexpr bool HAS_S = OP == EW_KL_BYROW || OP == EW_KL_BYCOL;
    constexpr bool BYCOL = OP == EW_COLS_DIV || OP == EW_KL_BYCOL;
    constexpr bool BYROW = OP == EW_ROWS_MUL || OP == EW_KL_BYROW;
    constexpr int U = 4;
    const long cvecs = cols / V;
    long r[U], cv[U];
    bool ok[U];
    if constexpr (LONG) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u] = blockIdx.y;
            cv[u] = ((long)blockIdx.x * U + u) * 256 + threadIdx.x;
            ok[u] = cv[u] < cvecs;
        }
    } else {
        const int tx = threadIdx.x & ((1 << txs) - 1), ty = threadIdx.x >> txs, TY = 256 >> txs;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            r[u] = ((long)blockIdx.x * U + u) * TY + ty;
            cv[u] = tx;
            ok[u] = r[u] < rows && tx < cvecs;
        }
    }
    // interior chunks (one block-uniform test) run without per-lane predicates: hipcc drains vmcnt(0) at the join of
    // every exec-masked load, which would serialise the U accesses
    bool full = true;
#pragma unroll
    for (int u = 0; u < U; ++u) full = full && ok[u];
    full = __syncthreads_and(full);
    auto body = [&](auto interior) {
        constexpr bool IN = decltype(interior)::value;
        float v[U][V], sv[U][V], colv[U][V], rowv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (IN || ok[u]) {
                if constexpr (NTP) {
                    load_vec_raw_nt<V>(v[u], X + r[u] * ldx + cv[u] * V);
                    if constexpr (HAS_S) load_vec_raw_nt<V>(sv[u], Sm + r[u] * lds_ + cv[u] * V);
                } else {
                    load_vec_raw<V>(v[u], X + r[u] * ldx + cv[u] * V);
                    if constexpr (HAS_S) load_vec_raw<V>(sv[u], Sm + r[u] * lds_ + cv[u] * V);
                }
                if constexpr (BYCOL) {
                    if (LONG || u == 0) load_vec_raw<V>(colv[u], x + cv[u] * V);
                }
                if constexpr (BYROW) rowv[u] = x[r[u]];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!(IN || ok[u])) continue;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float cden = BYCOL ? colv[LONG ? u : 0][e] + eps : 0.f;
                if constexpr (OP == EW_CLAMP) v[u][e] = fmaxf(v[u][e], eps);
                if constexpr (OP == EW_COLS_DIV) v[u][e] = v[u][e] / cden;
                if constexpr (OP == EW_ROWS_MUL) v[u][e] = v[u][e] * rowv[u];
                if constexpr (HAS_S) {
                    const float q = sv[u][e] / (BYROW ? rowv[u] + eps : cden);
                    v[u][e] = v[u][e] * q;
                    if (clamp) v[u][e] = fmaxf(v[u][e], eps);
                }
            }
            if constexpr (NTP && V == 4) __builtin_nontemporal_store(f32x4{v[u][0], v[u][1], v[u][2], v[u][3]}, reinterpret_cast<f32x4*>(X + r[u] * ldx + cv[u] * V));
            else store_tile_vec<V, true, true>(v[u], X + r[u] * ldx, cv[u] * V, cols, true);
        }
    };
    if (full) body(std::true_type{});
    else body(std::false_type{});
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
__global__ __launch_bounds__(256) void sqnorm_kernel(const TA* __restrict__ A, long m, long n, long lda, double* out, BatchTab bt) {
    REBASE(A); REBASE(out);
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

// x[j] = sum_c H[j][c]  -- one 16-wave workgroup per row, 16-byte loads where the row allows, fp64 accumulation in a fixed
// order (per-thread stride, wave butterfly, waves in order)
__global__ __launch_bounds__(1024) void rowsum_kernel(const float* __restrict__ H, long n, long ldh, float* x, BatchTab bt) {
    REBASE(H); REBASE(x);
    const float* row = H + (long)blockIdx.x * ldh;
    double acc = 0.0;
    const bool vec = ((uintptr_t)row & 15) == 0;
    const long n4 = vec ? n / 4 : 0;
    for (long c = threadIdx.x; c < n4; c += blockDim.x) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * c);
        acc += (double)((v[0] + v[1]) + (v[2] + v[3]));
    }
    for (long c = 4 * n4 + threadIdx.x; c < n; c += blockDim.x) acc += (double)row[c];
    __shared__ double red[16];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
        x[blockIdx.x] = (float)s;
    }
}

// stage 1 of x[j] = sum_i W[i][j]: partial[blk][j] over a slab of rows (coalesced along j)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ W, long m, int k, long ldw,
                                                             long rows_per_blk, float* partial, int kp, BatchTab bt) {
    REBASE(W); REBASE(partial);
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

// x[j] = sum over the slab partials, fp64.  One workgroup of 1024 threads: thread (j, g) adds slabs g, g + G, ... (G = 1024 / kp)
// in order, the G sums of a column are combined in fixed order through LDS -> deterministic.  (One thread per column walking
// all slabs was a 60 us latency chain at 256 slabs: 4 % of a KL step.)
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ partial, int nblk, int kp, int k,
                                                            float* __restrict__ x, BatchTab bt) {
    REBASE(partial); REBASE(x);
    __shared__ double red[1024];
    const int j = threadIdx.x % kp, g = threadIdx.x / kp, G = 1024 / kp;
    double acc = 0.0;
    for (int b = g; b < nblk; b += G) acc += (double)partial[(long)b * kp + j];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0 && j < k) {
        for (int q = 1; q < G; ++q) acc += red[q * kp + j];
        x[j] = (float)acc;
    }
}


}  // namespace
