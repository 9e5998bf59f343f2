// dnmf_tn.h -- TN form: C[j][c] = sum_i X[i][j] Y[i][c] straight from global memory (W^T A, W^T W) + the partial-sum reduction.
// Part of libdnmf_hip.so (kernels live in anonymous namespaces of the headers; the translation units csrc/*.hip include what they launch).
#pragma once
#include "dnmf_common.h"

namespace {

// =============================================================================================== TN form
enum { TN_PARTIAL = 0 };

struct TnArgs {
    const float* X; long ldx; int xcols;     // [nrows x xcols]  -> output rows j
    const void* Y; long ldy; long ycols;     // [nrows x ycols]  -> output cols c (float, or bf16 bits: TY of tn_kernel)
    long nrows; long rows_per_chunk; int nchunks; int ncolblk;
    float* P; long chunk_stride; long ldp;   // P[chunk][KP][ldp]
    float* Pg;                               // tn16_kernel<., GRAM>: partial X^T X per chunk [chunk][16][16]
};
__device__ __forceinline__ void rebase_args(TnArgs& p, const BatchTab& bt) {
    rebase(p.X, bt); rebase(p.Y, bt); rebase(p.P, bt); rebase(p.Pg, bt);
}

template <int KT, int NT, bool FAST, int U, typename TY>
__device__ __forceinline__ void tn_load(float (&a)[U][KT], float (&b)[U][NT], const float* __restrict__ X, long ldx,
                                        int xcols, const TY* __restrict__ Y, long ldy, long ycols, long col0,
                                        long r, long rend, int li, int h) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long row = r + 2 * u + h;
        const bool ok = row < rend;
        load_vec<KT, FAST>(a[u], X + row * ldx, (long)KT * li, xcols, ok);
        load_vec<NT, FAST>(b[u], Y + row * ldy, col0 + (long)NT * li, ycols, ok);
    }
}

template <int KT, int NT, int U>
__device__ __forceinline__ void tn_comp(f32x16 (&acc)[KT][NT], const float (&a)[U][KT], const float (&b)[U][NT]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int ke = 0; ke < KT; ++ke)
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) acc[ke][ne] = MFMA32(a[u][ke], b[u][ne], acc[ke][ne]);
}

// acc[ke][ne] (reg, lane) = C[j = KT*crow(reg,h) + ke][c = col0 + NT*li + ne], contraction over rows [rbeg, rend)
//
// FAST path = software pipeline over full batches of U row pairs, one batch ahead, with the two loads of the NEXT
// batch's row pair u issued right before the KT*NT MFMAs of THIS batch's row pair u (issue order pinned with
// sched_barrier).  Measured on MI355X (262144 x 8192, k = 64; tools/kbench.py): this interleave 2.34 ms; the same loads
// as one block of 8 ahead of the 32 MFMAs 3.5 ms (waves stall issuing VMEM while the matrix pipe idles: MFMA busy 51 %
// vs 88 %); exec-masked predicated loads (hipcc then drains with vmcnt(0)) 2.58 ms.  Loads are branch-free: the
// batch base is a wave-uniform pointer, the per-lane part (2u + h) * ld + column a loop-invariant 32-bit offset.
// Lanes whose output row j >= xcols or output column c >= ycols read a clamped (valid) column instead: an MFMA output
// row / column depends only on the matching A- / B-operand lane, so they only pollute outputs that are never stored.
template <int KT, int NT, bool FAST, bool NTY = false, typename TY = float>
__device__ __forceinline__ void tn_mainloop(f32x16 (&acc)[KT][NT], const float* __restrict__ X, long ldx, int xcols,
                                            const TY* __restrict__ Y, long ldy, long ycols, long col0, long rbeg,
                                            long rend, int li, int h) {
    constexpr int U = 4;  // row pairs per register batch
    float a0[U][KT], b0[U][NT], a1[U][KT];
    Raw<TY, NT> q0[U], q1[U];                // the streamed operand as loaded (bf16: widened right before its MFMAs)
    long r = rbeg;
    if constexpr (FAST) {
        const long nb = (rend - rbeg) / (2 * U);
        if (nb > 0 && 8 * ldx < 0x7fffffffL && 8 * ldy < 0x7fffffffL) {
            long xc = (long)KT * li, yc = col0 + (long)NT * li;
            xc = xc < xcols ? xc : xcols - KT;
            yc = yc < ycols ? yc : ycols - NT;
            int xo[U], yo[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xo[u] = (int)((2 * u + h) * ldx + xc);
                yo[u] = (int)((2 * u + h) * ldy + yc);
            }
            const long rlastb = rbeg + (nb - 1) * 2 * U;   // first row of the last full batch
            // round 3: fp32 operands whose chunk fits one 2 GiB window are addressed through MUBUF descriptors at the chunk's
            // first row -- lane offset loop invariant, batch offset scalar: no vector address arithmetic in the loop (the
            // pointer form costs one v_lshl_add_u64 per load, and every vector instruction takes matrix-pipe time here)
            constexpr bool F32 = std::is_same<TY, float>::value;
            const bool bufok = F32 && ((rend - rbeg + 8) * ldx + xcols) * 4 < 0x7fffffffL && ((rend - rbeg + 8) * ldy + ycols) * 4 < 0x7fffffffL;
            if (F32 && bufok) {
                if constexpr (F32) {
                    const i32x4 rsx = buf_rsrc(X + rbeg * ldx), rsy = buf_rsrc(reinterpret_cast<const float*>(Y) + rbeg * ldy);
                    const int ldx4 = (int)(ldx * 4), ldy4 = (int)(ldy * 4);
                    int xb[U], yb[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { xb[u] = xo[u] * 4; yb[u] = yo[u] * 4; }
                    auto ldb = [&](float (&a)[U][KT], Raw<TY, NT> (&q)[U], int u, int rrel) {
                        buf_load<KT, 0>(a[u], rsx, xb[u], rrel * ldx4);
                        buf_load<NT, NTY ? 2 : 0>(q[u].v, rsy, yb[u], rrel * ldy4);
                    };
#pragma unroll
                    for (int u = 0; u < U; ++u) ldb(a0, q0, u, 0);
                    const int nbi = (int)nb, rlast_rel = (int)(rlastb - rbeg);
                    int rr = 0, b = 0;
                    for (; b + 2 <= nbi; b += 2) {
                        const int r1 = rr + 2 * U;
                        int r2 = rr + 4 * U;
                        r2 = r2 < rlast_rel ? r2 : rlast_rel;          // prefetch past the end re-reads the last batch (unused)
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            ldb(a1, q1, u, r1);
                            __builtin_amdgcn_sched_barrier(0);
                            float bb[NT];
                            q0[u].get(bb);
#pragma unroll
                            for (int ke = 0; ke < KT; ++ke)
#pragma unroll
                                for (int ne = 0; ne < NT; ++ne) acc[ke][ne] = MFMA32(a0[u][ke], bb[ne], acc[ke][ne]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            ldb(a0, q0, u, r2);
                            __builtin_amdgcn_sched_barrier(0);
                            float bb[NT];
                            q1[u].get(bb);
#pragma unroll
                            for (int ke = 0; ke < KT; ++ke)
#pragma unroll
                                for (int ne = 0; ne < NT; ++ne) acc[ke][ne] = MFMA32(a1[u][ke], bb[ne], acc[ke][ne]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        rr += 4 * U;
                    }
                    r += (long)rr;
                    if (b < nbi) {
#pragma unroll
                        for (int u = 0; u < U; ++u) q0[u].get(b0[u]);
                        tn_comp<KT, NT, U>(acc, a0, b0);
                        r += 2 * U;
                    }
                }
            } else {
            {
                const float* X0 = X + r * ldx; const TY* Y0 = Y + r * ldy;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    load_vec_raw<KT>(a0[u], X0 + xo[u]);
                    if constexpr (NTY) q0[u].load_nt(Y0 + yo[u]); else q0[u].load(Y0 + yo[u]);
                }
            }
            long b = 0;
            for (; b + 2 <= nb; b += 2) {
                const long r1 = r + 2 * U;
                long r2 = r + 4 * U;
                r2 = r2 < rlastb ? r2 : rlastb;              // prefetch past the end re-reads the last batch (unused)
                const float* X1 = X + r1 * ldx; const TY* Y1 = Y + r1 * ldy;
                const float* X2 = X + r2 * ldx; const TY* Y2 = Y + r2 * ldy;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    load_vec_raw<KT>(a1[u], X1 + xo[u]);
                    if constexpr (NTY) q1[u].load_nt(Y1 + yo[u]); else q1[u].load(Y1 + yo[u]);
                    __builtin_amdgcn_sched_barrier(0);
                    float bb[NT];
                    q0[u].get(bb);
#pragma unroll
                    for (int ke = 0; ke < KT; ++ke)
#pragma unroll
                        for (int ne = 0; ne < NT; ++ne) acc[ke][ne] = MFMA32(a0[u][ke], bb[ne], acc[ke][ne]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    load_vec_raw<KT>(a0[u], X2 + xo[u]);
                    if constexpr (NTY) q0[u].load_nt(Y2 + yo[u]); else q0[u].load(Y2 + yo[u]);
                    __builtin_amdgcn_sched_barrier(0);
                    float bb[NT];
                    q1[u].get(bb);
#pragma unroll
                    for (int ke = 0; ke < KT; ++ke)
#pragma unroll
                        for (int ne = 0; ne < NT; ++ne) acc[ke][ne] = MFMA32(a1[u][ke], bb[ne], acc[ke][ne]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                r += 4 * U;
            }
            if (b < nb) {
#pragma unroll
                for (int u = 0; u < U; ++u) q0[u].get(b0[u]);
                tn_comp<KT, NT, U>(acc, a0, b0);
                r += 2 * U;
            }
            }
        }
    }
    // ragged tail of the FAST path and the whole generic path: predicated loads, zero fill
    for (; r < rend; r += 2 * U) {
        tn_load<KT, NT, FAST, U, TY>(a0, b0, X, ldx, xcols, Y, ldy, ycols, col0, r, rend, li, h);
        tn_comp<KT, NT, U>(acc, a0, b0);
    }
}

template <int KT, int NT, bool FAST, int MODE, bool NTY = false, typename TY = float>
__global__ __launch_bounds__(256, 2) void tn_kernel(TnArgs p, BatchTab bt) {
    rebase_args(p, bt);
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    // wave-uniform quantities kept provably scalar (readfirstlane) so row bases live in SGPRs
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Workgroup b covers 4 adjacent column blocks of one row chunk; consecutive workgroups (which the dispatcher
    // spreads round-robin over the 8 XCDs) continue along the same rows.  An XCD-aware order (a chunk's workgroups
    // consecutive on ONE XCD, so that W is fetched into one L2 instead of eight) was measured: HBM reads 9.14 -> 8.66 GB
    // per launch as predicted, but the kernel got 7-10 % SLOWER (2.19 -> 2.35 ms at 262144 rows, 0.354 -> 0.392 ms at
    // 32768): with the natural order the eight XCDs stream neighbouring 2 KiB pieces of the same rows of A at the same
    // time, which is worth more at the DRAM than the 5 % of traffic.
    const long gw = (long)blockIdx.x * (blockDim.x >> 6) + wid;
    const long chunk = gw / p.ncolblk;
    const long colblk = gw % p.ncolblk;
    if (chunk >= p.nchunks) return;
    const long col0 = colblk * 32 * NT;
    const long rbeg = chunk * p.rows_per_chunk;
    long rend = rbeg + p.rows_per_chunk;
    if (rend > p.nrows) rend = p.nrows;

    f32x16 acc[KT][NT];
#pragma unroll
    for (int ke = 0; ke < KT; ++ke)
#pragma unroll
        for (int ne = 0; ne < NT; ++ne)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ke][ne][r] = 0.f;

    tn_mainloop<KT, NT, FAST, NTY, TY>(acc, p.X, p.ldx, p.xcols, static_cast<const TY*>(p.Y), p.ldy, p.ycols, col0, rbeg, rend, li, h);

    if constexpr (MODE == TN_PARTIAL) {
        float* Pc = p.P + chunk * p.chunk_stride;
#pragma unroll
        for (int ke = 0; ke < KT; ++ke)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = KT * crow(r, h) + ke;
                float d[NT];
#pragma unroll
                for (int ne = 0; ne < NT; ++ne) d[ne] = acc[ke][ne][r];
                store_vec<NT, true>(d, Pc + (long)j * p.ldp, col0 + (long)NT * li, p.ldp, true);
            }
    }
}

// Optional tail of a reduction launch: one extra workgroup sums the partial Gram tiles a GRAM kernel wrote ([nsplit][dim][dim],
// dim = 16 / 32) into the zero-padded kp x kp Gram buffer G -- entries beyond k x k are written as 0, the garbage the clamped
// lanes left there is never read.  Sums in slab order: bitwise deterministic.
struct GramTail { const float* Pg; float* G; int dim, k, kp, nsplit; };
inline int gram_tail_blocks(int k) { return (k * k + 255) / 256; }      // host: extra workgroups of the reduction launch
// Tail workgroup b: the live entries b * 256 + thread of the k x k tile, one per thread, eight slab loads in flight (this is a
// latency chain next to a reduction that takes ~5 us: a first version walked the padded tile in ONE workgroup, 2 entries x 32
// dependent loads per thread at k = 16, and cost +9 us per iteration); workgroup 0 also writes the zero padding.
__device__ __forceinline__ void gram_tail(const GramTail& t, int b) {
    const int e = b * 256 + threadIdx.x;
    if (e < t.k * t.k) {
        const int j = e / t.k, jj = e % t.k;
        const float* __restrict__ src = t.Pg + j * t.dim + jj;
        float s = 0.f;
#pragma unroll 8
        for (int q = 0; q < t.nsplit; ++q) s += src[(long)q * t.dim * t.dim];
        t.G[j * t.kp + jj] = s;
    }
    if (b == 0)
        for (int idx = threadIdx.x; idx < t.kp * t.kp; idx += 256)
            if (idx / t.kp >= t.k || idx % t.kp >= t.k) t.G[idx] = 0.f;
}

// out[y][j][c] = sum_{s in slice y} P[s][j][c], j < rows, c < cols.  256 threads = 64 consecutive float4 outputs x 4
// split lanes; lane g sums splits g, g+4, ... of its slice in order, the four lane sums are combined in fixed order
// through LDS -> bitwise deterministic.  Everything else inside [rows_out x cols_out] is written as 0 (zero padding
// of the gram buffers).  gridDim.y > 1 = first stage of a two-stage reduction (out = scratch, y_stride apart).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ P, long stride, long ldp,
                                                              int nsplit, int splits_per_y, float* __restrict__ out,
                                                              long ldo, long y_stride, int rows, long cols,
                                                              int rows_out, long cols_out, GramTail gt, BatchTab bt) {
    __shared__ f32x4 red[256];
    REBASE(P); REBASE(out); rebase(gt.Pg, bt); rebase(gt.G, bt);
    if (gt.Pg && (int)blockIdx.x >= (int)gridDim.x - (gt.k * gt.k + 255) / 256) {     // (single-stage launches only: gridDim.y == 1)
        gram_tail(gt, (int)blockIdx.x - ((int)gridDim.x - (gt.k * gt.k + 255) / 256));
        return;
    }
    const long c4 = cdiv(cols_out, 4);
    const long total = (long)rows_out * c4;
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long idx = (long)blockIdx.x * 64 + o;
    const int s0 = blockIdx.y * splits_per_y;
    const int s1 = min(nsplit, s0 + splits_per_y);
    const int j = idx / c4;
    const long c = (idx % c4) * 4;
    const bool live = idx < total && j < rows && c < cols;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float* src = P + (long)j * ldp + c;
#pragma unroll 4
        for (int k = s0 + g; k < s1; k += 4) s += *reinterpret_cast<const f32x4*>(src + k * stride);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && idx < total) {
        s = ((red[o] + red[64 + o]) + red[128 + o]) + red[192 + o];
        float* dst = out + (long)blockIdx.y * y_stride + (long)j * ldo;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < cols_out) dst[c + e] = (live && c + e < cols) ? s[e] : 0.f;
    }
}

// Wide outputs (W^T A: k x n with n >= 4096): one lane per float4 output, 256 consecutive outputs (4 KiB of every slab)
// per workgroup, the lane adds the slabs in order with eight loads in flight.  Same sums as above in a different
// association; single stage only (nsplit <= 64).
__global__ __launch_bounds__(256) void reduce_partials_wide_kernel(const float* __restrict__ P, long stride, long ldp,
                                                                   int nsplit, float* __restrict__ out, long ldo, int rows,
                                                                   long cols, GramTail gt, BatchTab bt) {
    REBASE(P); REBASE(out); rebase(gt.Pg, bt); rebase(gt.G, bt);
    if (gt.Pg && (int)blockIdx.x >= (int)gridDim.x - (gt.k * gt.k + 255) / 256) {
        gram_tail(gt, (int)blockIdx.x - ((int)gridDim.x - (gt.k * gt.k + 255) / 256));
        return;
    }
    const long c4 = cols / 4;                       // cols % 4 == 0 (host checked)
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)rows * c4) return;
    const int j = idx / c4;
    const long c = (idx % c4) * 4;
    const float* src = P + (long)j * ldp + c;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int k = 0; k < nsplit; ++k) s += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + k * stride));
    *reinterpret_cast<f32x4*>(out + (long)j * ldo + c) = s;
}

}  // namespace
