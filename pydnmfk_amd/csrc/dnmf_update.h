// dnmf_update.h -- Frobenius multiplicative-update kernels (element-wise pass with the k x k product inside).
// Part of libdnmf_hip.so (single translation unit: csrc/dnmf.hip includes every header once).
#pragma once
#include "dnmf_common.h"

namespace {

// =============================================================================================== MU update kernels
// The Frobenius multiplicative updates are HBM-bound element-wise passes with a k x k product inside:
//   H[j][c] *= S[j][c] / ((G H)[j][c] + eps)      (dist_nmf.py:750-751, G = W^T W, S = W^T A)
//   W[i][j] *= S[i][j] / ((W G)[i][j] + eps)      (dist_nmf.py:731-732, G = H H^T, S = A H^T)
// Both kernels load their whole tile of the factor AND of S up front (maximum memory-level parallelism), use the
// factor registers directly as the MFMA B operand, and -- by choosing which two contraction indices each MFMA pairs --
// make the register that fed step t the very value the epilogue needs at accumulator position t, so the factor is read
// from memory exactly once and nothing goes through LDS.  Algorithmic traffic: 12 bytes per factor element.

// H update: wave tile = KP rows x 32*NT columns, KT*NT == 4.  acc[ke][ne] (reg r, lane (li,h)) = (G H)[j][c] with
// j = KT*crow(r,h) + ke, c = col0 + NT*li + ne.  Step (r, ke) contracts the row pair jj(h) = KT*crow(r,h) + ke:
// B operand = hreg[r][ke][ne] = H[jj(h)][c] (exactly the epilogue's H value), A operand lane (li,h) = G[jj(h)][KT*li + ke'].
template <int KT, int NT, bool FAST, bool INTERIOR>
__device__ __forceinline__ void update_h_tile(float* __restrict__ H, int k, long n, long ldh, const float* __restrict__ Sm,
                                              long lds_, const float* gs, float eps, int clamp, long c, int li, int h) {
    constexpr int KP = 32 * KT;
    float hreg[16][KT][NT], sreg[16][KT][NT];
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int ke = 0; ke < KT; ++ke) {
            const int j = KT * crow(r, h) + ke;
            load_tile_vec<NT, FAST, INTERIOR>(hreg[r][ke], H + (long)j * ldh, c, n, j < k);
            load_tile_vec<NT, FAST, INTERIOR>(sreg[r][ke], Sm + (long)j * lds_, c, n, j < k);
        }
    f32x16 acc[KT][NT];
#pragma unroll
    for (int ke = 0; ke < KT; ++ke)
#pragma unroll
        for (int ne = 0; ne < NT; ++ne)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ke][ne][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int ke = 0; ke < KT; ++ke) {
            const int jj = KT * crow(r, h) + ke;           // this lane half's contraction row (jj < KP; G is zero padded)
            float a[KT];
            load_vec_raw<KT>(a, &gs[jj * KP + KT * li]);       // 32 lanes x KT floats contiguous: conflict free
#pragma unroll
            for (int k2 = 0; k2 < KT; ++k2)
#pragma unroll
                for (int ne = 0; ne < NT; ++ne) acc[k2][ne] = MFMA32(a[k2], hreg[r][ke][ne], acc[k2][ne]);
        }
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int ke = 0; ke < KT; ++ke) {
            const int j = KT * crow(r, h) + ke;
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) {
                const float q = sreg[r][ke][ne] / (acc[ke][ne][r] + eps);
                float v = hreg[r][ke][ne] * q;
                if (clamp) v = fmaxf(v, eps);
                hreg[r][ke][ne] = v;
            }
            store_tile_vec<NT, FAST, INTERIOR>(hreg[r][ke], H + (long)j * ldh, c, n, j < k);
        }
}

template <int KT, int NT, bool FAST>
__global__ __launch_bounds__(256, 2) void update_h_kernel(float* __restrict__ H, int k, long n, long ldh,
                                                          const float* __restrict__ Sm, long lds_,
                                                          const float* __restrict__ G, float eps, int clamp) {
    constexpr int KP = 32 * KT;
    extern __shared__ __attribute__((aligned(16))) float gs[];   // G staged once per workgroup: rows jj, KP floats each
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long col0 = ((long)blockIdx.x * (blockDim.x >> 6) + wid) * 32 * NT;   // 1 or 4 waves per workgroup
    const long c = col0 + (long)NT * li;
    for (int idx = threadIdx.x; idx < KP * KP / 4; idx += blockDim.x)
        *reinterpret_cast<f32x4*>(&gs[idx * 4]) = *reinterpret_cast<const f32x4*>(G + idx * 4);
    __syncthreads();
    if (col0 >= n) return;
    if (FAST && k == KP && col0 + 32 * NT <= n) update_h_tile<KT, NT, FAST, true>(H, k, n, ldh, Sm, lds_, gs, eps, clamp, c, li, h);
    else update_h_tile<KT, NT, FAST, false>(H, k, n, ldh, Sm, lds_, gs, eps, clamp, c, li, h);
}

// W update: wave tile = 32 rows x KP columns.  Lane (li,h) owns row i = row0 + li and keeps W[i][8s + 4h + e] in
// wreg[s][e] (a contiguous 32-row block of W is read with 16-B pieces).  out[jt] (reg r = 4g + e, lane (li,h)) =
// (W G)[i][j], j = 32 jt + 8g + 4h + e = exactly the index of wreg[4 jt + g][e]; B operand of step (s, e) = wreg[s][e],
// A operand lane (li,h) = G[32 jt + li][8s + 4h + e] (G symmetric).
template <int KT, bool FAST, bool INTERIOR>
__device__ __forceinline__ void update_w_tile(float* __restrict__ W, int k, long ldw, const float* __restrict__ Sm,
                                              long lds_, const float* gs, float eps, long row, bool rok, int li, int h) {
    constexpr int KP = 32 * KT, GP = KP + 4;
    float wreg[4 * KT][4], sreg[4 * KT][4];
#pragma unroll
    for (int s = 0; s < 4 * KT; ++s) {
        load_tile_vec<4, FAST, INTERIOR>(wreg[s], W + row * ldw, 8 * s + 4 * h, k, rok);
        load_tile_vec<4, FAST, INTERIOR>(sreg[s], Sm + row * lds_, 8 * s + 4 * h, k, rok);
    }
    f32x16 out[KT];
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[jt][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4 * KT; ++s)
#pragma unroll
        for (int jt = 0; jt < KT; ++jt) {
            float a[4];
            load_vec_raw<4>(a, &gs[(jt * 32 + li) * GP + 8 * s + 4 * h]);
#pragma unroll
            for (int e = 0; e < 4; ++e) out[jt] = MFMA32(a[e], wreg[s][e], out[jt]);
        }
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int s = 4 * jt + g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float q = sreg[s][e] / (out[jt][4 * g + e] + eps);
                wreg[s][e] = wreg[s][e] * q;
            }
            store_tile_vec<4, FAST, INTERIOR>(wreg[s], W + row * ldw, 8 * s + 4 * h, k, rok);
        }
}

template <int KT, bool FAST>
__global__ __launch_bounds__(256, 2) void update_w_kernel(float* __restrict__ W, long m, int k, long ldw,
                                                          const float* __restrict__ Sm, long lds_,
                                                          const float* __restrict__ G, float eps) {
    constexpr int KP = 32 * KT, GP = KP + 4;   // LDS row pitch: +16 B so that rows 0..15 land on distinct 16-B slots
    extern __shared__ __attribute__((aligned(16))) float gs[];
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long row0 = ((long)blockIdx.x * 4 + wid) * 32;
    const long row = row0 + li;
    for (int idx = threadIdx.x; idx < KP * KP / 4; idx += 256) {
        const int gr = idx / (KP / 4), gc = (idx % (KP / 4)) * 4;
        *reinterpret_cast<f32x4*>(&gs[gr * GP + gc]) = *reinterpret_cast<const f32x4*>(G + gr * KP + gc);
    }
    __syncthreads();
    if (row0 >= m) return;
    if (FAST && k == KP && row0 + 32 <= m) update_w_tile<KT, FAST, true>(W, k, ldw, Sm, lds_, gs, eps, row, true, li, h);
    else update_w_tile<KT, FAST, false>(W, k, ldw, Sm, lds_, gs, eps, row, row < m, li, h);
}


}  // namespace

namespace {

// =============================================================================================== block-sequential form
// Same arithmetic and operand pairing as above, but the KT output blocks of 32 factor indices are produced one after the
// other: only ONE 32-index block of the product (16 accumulator registers per 32 columns) and of the numerator S is live at
// a time, beside the factor tile that every block needs as its MFMA B operand.  Registers per lane: factor KP/2 (per 32
// columns / per 32 rows) + 16 + 16 instead of 3 x KP/2, i.e. 4-5 waves per SIMD at k = 64 and 4 at k = 128 where the
// all-at-once form above holds 2 (and spills at k = 128).  The numerator block is requested at the top of its block's
// MFMA loop and consumed in its epilogue.  Updated values go straight to memory from temporaries: the factor registers
// must keep the OLD values, later blocks still contract over them.

// V floats at ub[loff ..]: `ub` is a wave-uniform row pointer (an SGPR pair), `loff` the lane's 32-bit element offset, so
// that no 64-bit per-row address has to be kept in vector registers between a tile's loads and its stores (hipcc
// otherwise keeps all of them live: 2 VGPRs per row).  Element e is valid iff ok && e < nvalid; FAST: all or none.
template <int V, bool FAST, bool INTERIOR>
__device__ __forceinline__ void ldu(float (&d)[V], const float* __restrict__ ub, unsigned loff, bool ok, long nvalid) {
    if constexpr (FAST && INTERIOR) load_vec_raw<V>(d, ub + loff);
    else if constexpr (FAST) {
        if (ok && nvalid > 0) load_vec_raw<V>(d, ub + loff);
        else {
#pragma unroll
            for (int e = 0; e < V; ++e) d[e] = 0.f;
        }
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] = (ok && e < nvalid) ? ub[loff + e] : 0.f;
    }
}

template <int V, bool FAST, bool INTERIOR>
__device__ __forceinline__ void stu(const float (&d)[V], float* __restrict__ ub, unsigned loff, bool ok, long nvalid) {
    if constexpr (FAST) {
        if (INTERIOR || (ok && nvalid > 0)) {
            if constexpr (V == 4) *reinterpret_cast<f32x4*>(ub + loff) = f32x4{d[0], d[1], d[2], d[3]};
            else if constexpr (V == 2) *reinterpret_cast<f32x2*>(ub + loff) = f32x2{d[0], d[1]};
            else ub[loff] = d[0];
        }
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e)
            if (ok && e < nvalid) ub[loff + e] = d[e];
    }
}

// H: wave tile = KP rows x 32*NT columns starting at column col0.  hreg[jb][r] in lane (li, h) = H[32 jb + crow(r, h)][c ..
// c+NT), c = col0 + NT li.  Row crow(r, h) = ju(r) + 4h with ju(r) = (r & 3) + 8 (r >> 2) wave-uniform: the row pointer of
// ju is uniform and the lane offset 4h ldh + NT li is the same for every row.
// Step (jb, r) of output block ob: A operand lane (li, h) = G[32 ob + li][32 jb + crow(r, h)], read four at a time
// (r = 4g .. 4g+3 are four consecutive columns of G starting at 32 jb + 8g + 4h).
template <int KT, int NT, bool FAST, bool INTERIOR>
__device__ __forceinline__ void update_h_seq_tile(float* __restrict__ H, int k, long n, long ldh, const float* __restrict__ Sm,
                                                  long lds_, const float* gs, float eps, int clamp, long col0, int li, int h) {
    constexpr int KP = 32 * KT, GP = KP + 4;
    const unsigned hoff = (unsigned)(4 * h) * (unsigned)ldh + (unsigned)(NT * li);
    const unsigned soff = (unsigned)(4 * h) * (unsigned)lds_ + (unsigned)(NT * li);
    const long nvalid = n - (col0 + NT * li);
    float* __restrict__ Hb = H + col0;
    const float* __restrict__ Sb = Sm + col0;
    float hreg[KT][16][NT];
#pragma unroll
    for (int jb = 0; jb < KT; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ju = 32 * jb + (r & 3) + 8 * (r >> 2);
            ldu<NT, FAST, INTERIOR>(hreg[jb][r], Hb + (long)ju * ldh, hoff, ju + 4 * h < k, nvalid);
        }
#pragma unroll
    for (int ob = 0; ob < KT; ++ob) {
        float sreg[16][NT];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ju = 32 * ob + (r & 3) + 8 * (r >> 2);
            ldu<NT, FAST, INTERIOR>(sreg[r], Sb + (long)ju * lds_, soff, ju + 4 * h < k, nvalid);
        }
        f32x16 acc[NT];
#pragma unroll
        for (int ne = 0; ne < NT; ++ne)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ne][r] = 0.f;
#pragma unroll
        for (int jb = 0; jb < KT; ++jb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float a[4];
                load_vec_raw<4>(a, &gs[(32 * ob + li) * GP + 32 * jb + 8 * g + 4 * h]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ne = 0; ne < NT; ++ne) acc[ne] = MFMA32(a[e], hreg[jb][4 * g + e][ne], acc[ne]);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ju = 32 * ob + (r & 3) + 8 * (r >> 2);
            float o[NT];
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) {
                const float q = sreg[r][ne] / (acc[ne][r] + eps);
                float v = hreg[ob][r][ne] * q;
                if (clamp) v = fmaxf(v, eps);
                o[ne] = v;
            }
            stu<NT, FAST, INTERIOR>(o, Hb + (long)ju * ldh, hoff, ju + 4 * h < k, nvalid);
        }
    }
}

// Workgroups walk the column tiles grid-stride (G is staged once per workgroup, not once per tile).
template <int KT, int NT, bool FAST, int OCC>
__global__ __launch_bounds__(256, OCC) void update_h_seq_kernel(float* __restrict__ H, int k, long n, long ldh,
                                                                const float* __restrict__ Sm, long lds_,
                                                                const float* __restrict__ G, float eps, int clamp) {
    constexpr int KP = 32 * KT, GP = KP + 4;
    extern __shared__ __attribute__((aligned(16))) float gs[];
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int idx = threadIdx.x; idx < KP * KP / 4; idx += 256) {
        const int gr = idx / (KP / 4), gc = (idx % (KP / 4)) * 4;
        *reinterpret_cast<f32x4*>(&gs[gr * GP + gc]) = *reinterpret_cast<const f32x4*>(G + gr * KP + gc);
    }
    __syncthreads();
    const long ntiles = cdiv(n, 32 * NT);
    for (long t = (long)blockIdx.x * 4 + wid; t < ntiles; t += (long)gridDim.x * 4) {
        const long col0 = t * 32 * NT;
        if (FAST && k == KP && col0 + 32 * NT <= n) update_h_seq_tile<KT, NT, FAST, true>(H, k, n, ldh, Sm, lds_, gs, eps, clamp, col0, li, h);
        else update_h_seq_tile<KT, NT, FAST, false>(H, k, n, ldh, Sm, lds_, gs, eps, clamp, col0, li, h);
    }
}

// W: wave tile = 32 rows x KP columns, lane (li, h) owns row row0 + li; wreg[s][e] = W[i][8s + 4h + e] (as above).
template <int KT, bool FAST, bool INTERIOR>
__device__ __forceinline__ void update_w_seq_tile(float* __restrict__ W, int k, long ldw, const float* __restrict__ Sm,
                                                  long lds_, const float* gs, float eps, long row, bool rok, int li, int h) {
    constexpr int KP = 32 * KT, GP = KP + 4;
    float wreg[4 * KT][4];
#pragma unroll
    for (int s = 0; s < 4 * KT; ++s) load_tile_vec<4, FAST, INTERIOR>(wreg[s], W + row * ldw, 8 * s + 4 * h, k, rok);
#pragma unroll
    for (int jt = 0; jt < KT; ++jt) {
        float sreg[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) load_tile_vec<4, FAST, INTERIOR>(sreg[g], Sm + row * lds_, 32 * jt + 8 * g + 4 * h, k, rok);
        f32x16 out;
#pragma unroll
        for (int r = 0; r < 16; ++r) out[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4 * KT; ++s) {
            float a[4];
            load_vec_raw<4>(a, &gs[(jt * 32 + li) * GP + 8 * s + 4 * h]);
#pragma unroll
            for (int e = 0; e < 4; ++e) out = MFMA32(a[e], wreg[s][e], out);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = wreg[4 * jt + g][e] * (sreg[g][e] / (out[4 * g + e] + eps));
            store_tile_vec<4, FAST, INTERIOR>(o, W + row * ldw, 32 * jt + 8 * g + 4 * h, k, rok);
        }
    }
}

template <int KT, bool FAST, int OCC>
__global__ __launch_bounds__(256, OCC) void update_w_seq_kernel(float* __restrict__ W, long m, int k, long ldw,
                                                                const float* __restrict__ Sm, long lds_,
                                                                const float* __restrict__ G, float eps) {
    constexpr int KP = 32 * KT, GP = KP + 4;
    extern __shared__ __attribute__((aligned(16))) float gs[];
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int idx = threadIdx.x; idx < KP * KP / 4; idx += 256) {
        const int gr = idx / (KP / 4), gc = (idx % (KP / 4)) * 4;
        *reinterpret_cast<f32x4*>(&gs[gr * GP + gc]) = *reinterpret_cast<const f32x4*>(G + gr * KP + gc);
    }
    __syncthreads();
    const long ntiles = cdiv(m, 32);
    for (long t = (long)blockIdx.x * 4 + wid; t < ntiles; t += (long)gridDim.x * 4) {
        const long row0 = t * 32, row = row0 + li;
        if (FAST && k == KP && row0 + 32 <= m) update_w_seq_tile<KT, FAST, true>(W, k, ldw, Sm, lds_, gs, eps, row, true, li, h);
        else update_w_seq_tile<KT, FAST, false>(W, k, ldw, Sm, lds_, gs, eps, row, row < m, li, h);
    }
}

}  // namespace
