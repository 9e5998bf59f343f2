// dnmf_k16.h -- the two big contractions for rank k <= 16 on v_mfma_f32_16x16x4_f32.
// Part of libdnmf_hip.so (kernels live in anonymous namespaces of the headers; the translation units csrc/*.hip include what they launch).
//
// The 32x32x2 kernels pad the rank to 32, so an NMFk sweep over k = 2..16 pays twice the matrix work it needs; with
// bf16-stored X (half the HBM bytes) that padded work, not the memory, sets the pace.  The 16x16x4 instruction has
// the same flop rate and a 16-wide rank dimension.  Operand maps (lane l, i16 = l & 15, kq = l >> 4):
//   A-operand: A[i = i16][kk = kq]     B-operand: B[kk = kq][j = i16]     C/D (4 registers): C[4 kq + r][i16]
// Same staging, swizzles and pipelines as the 32-wide kernels (dnmf_nt.h, dnmf_tn.h); every tile is made interior by
// clamping row indices (rows >= nrows / factor rows >= k read a valid row whose outputs are never stored), the host
// falls back to the 32-wide kernels when the column count is not a multiple of the tile.
#pragma once
#include "dnmf_common.h"
#include "dnmf_nt.h"
#include "dnmf_tn.h"

namespace {

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ================================================================================================ NT form, k <= 16
// Workgroup = 4 waves x 32 rows (two 16-row sub-tiles per wave); k-tile = 128 bytes of every X row (32 fp32 or 64 bf16
// contraction indices); Y tile = the 16 factor rows x the same contraction range in fp32.  Two tiles in flight, branch
// free (nt_mainloop_p2).  Fragment of sub-tile rs, group s in {0, 1}: lane (i16, kq) reads the 16-byte chunk 4s + kq of
// X row rs*16 + i16 and the matching contraction range of Y row i16.
template <typename TX, int MODE>
__global__ __launch_bounds__(256) void nt16_kernel(NtArgs p, BatchTab bt) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    rebase_args(p, bt);
    constexpr bool B16 = std::is_same<TX, bf16_t>::value;
    constexpr int BKE = B16 ? BKH : BK;          // contraction indices per k-tile
    constexpr int EPC = B16 ? 8 : 4;             // X elements per 16-byte chunk
    constexpr int BM = 128, T = 256, XT = BM * BK, YT = 16 * BKE, STAGE = XT + YT;
    constexpr int YCH = BKE / 4;                 // 16-byte chunks per Y row (8 or 16)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, kq = lane >> 4;
    const long row0 = (long)blockIdx.x * BM;
    const TX* X = static_cast<const TX*>(p.X);
    const long nk = p.ncols / BKE;

    // staging addresses: X thread t -> chunk t & 7 of rows (t >> 3) + 32 it; Y thread t -> chunk t % YCH of row t / YCH
    const TX* xp[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        long r = row0 + (tid >> 3) + 32 * it;
        r = r < p.nrows ? r : p.nrows - 1;
        xp[it] = X + r * p.ldx + (tid & 7) * EPC;
    }
    // (fp32: the Y tile has 128 chunks, so threads 128..255 duplicate the loads and the LDS writes of threads 0..127 --
    // identical values to identical addresses -- instead of branching: an exec-masked load would make hipcc drain vmcnt)
    const int ty = tid % (16 * YCH);
    int yr = ty / YCH;
    yr = yr < p.yrows ? yr : p.yrows - 1;
    const float* yp = p.Y + (long)yr * p.ldy + (ty % YCH) * 4;
    const long kshift = (long)((blockIdx.x * 37u) % (unsigned long)nk);
    f32x4 x0[4], x1[4], y0, y1;
    auto load = [&](f32x4 (&xr)[4], f32x4& yv, long kt) {
        kt = kt < nk ? kt : nk - 1;
        kt += kshift;
        kt = kt >= nk ? kt - nk : kt;
        const long c0 = kt * BKE;
#pragma unroll
        for (int it = 0; it < 4; ++it) xr[it] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xp[it] + c0));
        yv = *reinterpret_cast<const f32x4*>(yp + c0 + yblk_off(p.yb, c0));     // (Y as column blocks: NtArgs::yb)
    };
    auto store = [&](float* st, const f32x4 (&xr)[4], const f32x4& yv) {
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(&st[lds_idx((tid >> 3) + 32 * it, tid & 7)]) = xr[it];
        if constexpr (B16) *reinterpret_cast<f32x4*>(&st[XT + ydx(ty / YCH, ty % YCH)]) = yv;
        else *reinterpret_cast<f32x4*>(&st[XT + lds_idx(ty / YCH, ty % YCH)]) = yv;
    };
    f32x4 acc[2];
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) acc[rs] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto group = [&](const float* st, int s) {
        f32x4 a[2];
#pragma unroll
        for (int rs = 0; rs < 2; ++rs) a[rs] = *reinterpret_cast<const f32x4*>(&st[lds_idx(wave * 32 + rs * 16 + i16, 4 * s + kq)]);
        if constexpr (B16) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(&st[XT + ydx(i16, 2 * (4 * s + kq))]);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(&st[XT + ydx(i16, 2 * (4 * s + kq) + 1)]);
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int rs = 0; rs < 2; ++rs) {
                    const unsigned int w = __float_as_uint(a[rs][e >> 1]);
                    acc[rs] = MFMA16((e & 1) ? bf16_hi(w) : bf16_lo(w), e < 4 ? b0[e & 3] : b1[e & 3], acc[rs]);
                }
        } else {
            const f32x4 b = *reinterpret_cast<const f32x4*>(&st[XT + lds_idx(i16, 4 * s + kq)]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int rs = 0; rs < 2; ++rs) acc[rs] = MFMA16(a[rs][e], b[e], acc[rs]);
        }
    };
    float* st0 = smem;
    float* st1 = smem + STAGE;
    load(x0, y0, 0);
    store(st0, x0, y0);
    __syncthreads();
    load(x1, y1, 1);
    long kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        load(x0, y0, kt + 2);
        group(st0, 0);
        __builtin_amdgcn_sched_barrier(0);
        store(st1, x1, y1);
        __builtin_amdgcn_sched_barrier(0);
        group(st0, 1);
        __syncthreads();
        load(x1, y1, kt + 3);
        group(st1, 0);
        __builtin_amdgcn_sched_barrier(0);
        store(st0, x0, y0);
        __builtin_amdgcn_sched_barrier(0);
        group(st1, 1);
        __syncthreads();
    }
    if (kt < nk) {
        group(st0, 0);
        group(st0, 1);
    }

    // acc[rs] (register r, lane (i16, kq)) = C[row = row0 + wave*32 + rs*16 + 4 kq + r][j = i16]
    if constexpr (MODE == NT_STORE) {
#pragma unroll
        for (int rs = 0; rs < 2; ++rs)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = row0 + wave * 32 + rs * 16 + 4 * kq + r;
                if (row < p.nrows && i16 < p.yrows) p.out[row * p.ldo + i16] = acc[rs][r];
            }
    } else {
        // second product acc2 = W[rows] . G (16 contraction indices: MFMA t takes kk = 4 kq + t); G is KP x KP = 32 x 32,
        // zero padded and symmetric, W columns >= k are predicated to zero
        f32x4 acc2[2];
#pragma unroll
        for (int rs = 0; rs < 2; ++rs) {
            acc2[rs] = f32x4{0.f, 0.f, 0.f, 0.f};
            const long wr = row0 + wave * 32 + rs * 16 + i16;
            float w[4];
            if (p.wfast) load_vec<4, true>(w, p.W + wr * p.ldw, 4 * kq, p.k, wr < p.nrows);
            else load_vec<4, false>(w, p.W + wr * p.ldw, 4 * kq, p.k, wr < p.nrows);
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.G + i16 * 32 + 4 * kq);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc2[rs] = MFMA16(w[t], g[t], acc2[rs]);
        }
#pragma unroll
        for (int rs = 0; rs < 2; ++rs)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = row0 + wave * 32 + rs * 16 + 4 * kq + r;
                if (row < p.nrows && i16 < p.k) {
                    const float wv = p.W[row * p.ldw + i16];
                    p.W[row * p.ldw + i16] = wv * (acc[rs][r] / (acc2[rs][r] + p.eps));   // dist_nmf.py:731-732
                }
            }
    }
}

constexpr size_t nt16_lds_bytes(bool b16) { return 2ul * (128 * BK + 16 * (b16 ? BKH : BK)) * sizeof(float); }

// ================================================================================================ TN form, k <= 16
// One wave owns 16 V columns (V = elements per 16-byte load: 4 fp32 / 8 bf16) and a chunk of rows; a step contracts 4
// rows: lane (c16, kq) loads Y[r + kq][col0 + V c16 .. + V - 1] (one row = 256 contiguous bytes across 16 lanes) and
// X[r + kq][j = c16]; V MFMAs per step.  Branch-free pipeline of U steps per batch, one batch ahead (tn_mainloop).
// GRAM: the wave also sums X^T X over its rows -- the lane's X value is the A AND the B operand of that product (A[i = c16][k =
// kq] = X[r + kq][c16] = B[k = kq][n = c16]): one more MFMA per step, no more loads.  Run in ONE wave per row chunk (tn16_kernel).
template <typename TY, bool GRAM>
__device__ __forceinline__ void tn16_body(const TnArgs& p, long chunk, long colblk, f32x4* accg) {
    constexpr int V = std::is_same<TY, bf16_t>::value ? 8 : 4, U = 4;
    const int lane = threadIdx.x & 63, c16 = lane & 15, kq = lane >> 4;
    const long col0 = colblk * 16 * V;
    const long rbeg = chunk * p.rows_per_chunk;
    long rend = rbeg + p.rows_per_chunk;
    if (rend > p.nrows) rend = p.nrows;
    const TY* Y = static_cast<const TY*>(p.Y);
    f32x4 acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int xj = c16 < p.xcols ? c16 : p.xcols - 1;
    const float* xb = p.X + (long)kq * p.ldx + xj;
    const TY* yb = Y + (long)kq * p.ldy + col0 + V * c16;
    Raw<TY, V> q0[U], q1[U];
    float w0[U], w1[U];
    auto ld = [&](Raw<TY, V> (&q)[U], float (&w)[U], long r, int u) {
        w[u] = xb[(r + 4 * u) * p.ldx];
        q[u].load_nt(yb + (r + 4 * u) * p.ldy);
    };
    auto mm = [&](const Raw<TY, V> (&q)[U], const float (&w)[U], int u) {
        float bb[V];
        q[u].get(bb);
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = MFMA16(w[u], bb[v], acc[v]);
        if constexpr (GRAM) *accg = MFMA16(w[u], w[u], *accg);
    };
    long r = rbeg;
    const long nb = (rend - rbeg) / (4 * U);
    if (nb > 0) {
        const long rlast = rbeg + (nb - 1) * 4 * U;
#pragma unroll
        for (int u = 0; u < U; ++u) ld(q0, w0, r, u);
        long b = 0;
        for (; b + 2 <= nb; b += 2) {
            const long r1 = r + 4 * U;
            long r2 = r + 8 * U;
            r2 = r2 < rlast ? r2 : rlast;          // prefetch past the end re-reads the last batch (unused)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ld(q1, w1, r1, u);
                __builtin_amdgcn_sched_barrier(0);
                mm(q0, w0, u);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ld(q0, w0, r2, u);
                __builtin_amdgcn_sched_barrier(0);
                mm(q1, w1, u);
                __builtin_amdgcn_sched_barrier(0);
            }
            r += 8 * U;
        }
        if (b < nb) {
#pragma unroll
            for (int u = 0; u < U; ++u) mm(q0, w0, u);
            r += 4 * U;
        }
    }
    for (; r < rend; r += 4) {                     // ragged tail: predicated, zero fill
        const bool ok = r + kq < rend;
        const float w = ok ? xb[r * p.ldx] : 0.f;
        float bb[V];
        load_vec<V, true>(bb, Y + (r + kq) * p.ldy, col0 + V * c16, p.ycols, ok);
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = MFMA16(w, bb[v], acc[v]);
        if constexpr (GRAM) *accg = MFMA16(w, w, *accg);
    }
    // acc[v] (register r, lane (c16, kq)) = C[j = 4 kq + r][c = col0 + V c16 + v]
    float* Pc = p.P + chunk * p.chunk_stride;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        float d[V];
#pragma unroll
        for (int v = 0; v < V; ++v) d[v] = acc[v][rr];
        float* dst = Pc + (long)(4 * kq + rr) * p.ldp + col0 + V * c16;
#pragma unroll
        for (int v = 0; v < V; v += 4) *reinterpret_cast<f32x4*>(dst + v) = f32x4{d[v], d[v + 1], d[v + 2], d[v + 3]};
    }
}

template <typename TY, bool GRAM = false>
__global__ __launch_bounds__(256) void tn16_kernel(TnArgs p, BatchTab bt) {
    rebase_args(p, bt);
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long gw = (long)blockIdx.x * 4 + wid;
    const long chunk = gw / p.ncolblk, colblk = gw % p.ncolblk;
    if (chunk >= p.nchunks) return;
    if constexpr (GRAM) {
        if (colblk == 0) {                       // wave-uniform: the whole loop is instantiated twice, no conditional MFMA in it
            const int lane = threadIdx.x & 63, c16 = lane & 15, kq = lane >> 4;
            f32x4 accg = {0.f, 0.f, 0.f, 0.f};
            tn16_body<TY, true>(p, chunk, colblk, &accg);
            float* g = p.Pg + chunk * 256;       // accg (register r, lane (c16, kq)) = G[j = 4 kq + r][j' = c16]; j, j' >= xcols never read
#pragma unroll
            for (int r = 0; r < 4; ++r) g[(4 * kq + r) * 16 + c16] = accg[r];
            return;
        }
    }
    tn16_body<TY, false>(p, chunk, colblk, nullptr);
}

}  // namespace
