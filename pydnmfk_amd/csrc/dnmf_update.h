// dnmf_update.h -- Frobenius multiplicative-update kernels (element-wise pass with the k x k product inside).
// Part of libdnmf_hip.so (kernels live in anonymous namespaces of the headers; the translation units csrc/*.hip include what they launch).
#pragma once
#include "dnmf_common.h"

namespace {

// =============================================================================================== MU update kernels
// The Frobenius multiplicative updates are HBM-bound element-wise passes with a k x k product inside:
//   H[j][c] *= S[j][c] / ((G H)[j][c] + eps)      (dist_nmf.py:750-751, G = W^T W, S = W^T A)
//   W[i][j] *= S[i][j] / ((W G)[i][j] + eps)      (dist_nmf.py:731-732, G = H H^T, S = A H^T)
// The factor tile is loaded once and its registers are the MFMA B operand directly; the pairing of contraction indices
// per MFMA is chosen so that the register that fed step t is the very value the epilogue needs at accumulator position
// t -- the factor is read from memory exactly once and nothing but G (staged once per workgroup) goes through LDS.
// Algorithmic traffic: 12 bytes per factor element.
//
// Block-sequential form: the KT output blocks of 32 factor indices are produced one after the other, so only ONE
// 32-index block of the product (16 accumulator registers per 32 columns) and of the numerator S is live at a time beside
// the factor tile that every block needs as its MFMA B operand: registers per lane = factor KP/2 (per 32 columns / 32
// rows) + 16 + 16.  (The first version formed all KT blocks at once -- three KP/2-register tiles, 2 waves per SIMD,
// spilling at k = 128: 3.07 TB/s at k = 64 and 1.35 TB/s at k = 128 on a 3.2 / 6.4 GB pass where this form reaches
// 4.97 / 3.44.)  The numerator block is requested at the top of its block's MFMA loop and consumed in its epilogue.
// Updated values go straight to memory from temporaries: the factor registers must keep the OLD values, later blocks
// still contract over them.  All global accesses are buffer loads / stores (dnmf_common.h): one offset register per
// lane for the whole tile, edge lanes switched off through the offset.

// s / d for d = (product) + eps > 0 (round 4): numerator times v_rcp_f32 -- 2 vector instructions (+ the add of eps) where
// hipcc's IEEE division sequence (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup) is 10.  On gfx950 every fp32 vector
// instruction costs matrix-pipe time (tools/coissue.hip), and these kernels carry one division per factor element beside
// k / 32 MFMAs per element-row: at k = 64 the epilogue was ~30 % of a tile's cycles.  <= 1.5 ulp from the correctly rounded
// quotient -- the same form as the KL products (dnmf_nn.h kl_quot); a step's parity budget is 1e-5.
// eps is ADDED here, not used as the accumulators' initial value as in the KL kernels: these loops carry no scheduling
// barriers, and hipcc then reused the dead 16-register block that held eps (the C operand of the first MFMA) as the
// destination of the next ds_read_b128 -- whose data lands while that MFMA is still reading C (HAZARD 2 in dnmf_common.h;
// found as 5 % errors in the last 16 factor columns of test_mu_updates[300-260-32]).
__device__ __forceinline__ float mu_quot(float s, float d) { return s * __builtin_amdgcn_rcpf(d); }

// waves per SIMD asked of the compiler for the kernels that contain the edge tile code (per-row offset selects: ~40 more
// registers) or the dword form of the W tile: what they hold without spilling
constexpr int edge_occ(int kt, int occ) { return kt == 4 ? 2 : (occ < 4 ? occ : 4); }

// H: wave tile = KP rows x 32*NT columns starting at column col0.  hreg[jb][r] in lane (li, h) = H[32 jb + crow(r, h)][c ..
// c+NT), c = col0 + NT li.  Row crow(r, h) = ju(r) + 4h with ju(r) = (r & 3) + 8 (r >> 2) wave-uniform -> SGPR offset
// ju ldh; the lane offset 4h ldh + NT li is the same for every row.  The descriptor is re-based per 32-row block.
// Step (jb, r) of output block ob: A operand lane (li, h) = G[32 ob + li][32 jb + crow(r, h)], read four at a time
// (r = 4g .. 4g+3 are four consecutive columns of G starting at 32 jb + 8g + 4h).
// NT = 1 needs no alignment at all (dword accesses); NT = 2 needs 8-byte aligned rows and an even n.
template <int KT, int NT, bool INTERIOR, bool MMA = true, int AUXL = 0, int AUXS = 0>
__device__ __forceinline__ void update_h_seq_tile(float* __restrict__ H, int k, long n, long ldh, const float* __restrict__ Sm,
                                                  long lds_, const float* gs, float eps, int clamp, long col0, int li, int h) {
    constexpr int KP = 32 * KT, GP = KP + 4;
    int hoff = (int)(((long)(4 * h) * ldh + NT * li) * 4);
    int soff = (int)(((long)(4 * h) * lds_ + NT * li) * 4);
    if constexpr (!INTERIOR) {
        if (col0 + NT * li >= n) { hoff = BUF_OOB; soff = BUF_OOB; }   // FAST semantics: a vector is wholly in or out
    }
    // row predicate of the edge variant: rows >= k read as zero / are not written
    auto vo = [&](int off, int ju) -> int {
        if constexpr (INTERIOR) return off;
        else return (ju + 4 * h < k) ? off : BUF_OOB;
    };
    i32x4 hrs[KT], srs[KT];
#pragma unroll
    for (int jb = 0; jb < KT; ++jb) {
        hrs[jb] = buf_rsrc(H + col0 + (long)(32 * jb) * ldh);
        srs[jb] = buf_rsrc(Sm + col0 + (long)(32 * jb) * lds_);
    }
    const int ldh4 = (int)(ldh * 4), lds4 = (int)(lds_ * 4);
    float hreg[KT][16][NT];
#pragma unroll
    for (int jb = 0; jb < KT; ++jb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ju = (r & 3) + 8 * (r >> 2);
            buf_load<NT, AUXL>(hreg[jb][r], hrs[jb], vo(hoff, 32 * jb + ju), ju * ldh4);
        }
#pragma unroll
    for (int ob = 0; ob < KT; ++ob) {
        float sreg[16][NT];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ju = (r & 3) + 8 * (r >> 2);
            buf_load<NT, AUXL>(sreg[r], srs[ob], vo(soff, 32 * ob + ju), ju * lds4);
        }
        f32x16 acc[NT];                 // (zero, not eps, as the initial value: see mu_quot)
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
                    for (int ne = 0; ne < NT; ++ne) {
                        if constexpr (MMA) acc[ne] = MFMA32(a[e], hreg[jb][4 * g + e][ne], acc[ne]);
                        else acc[ne][4 * g + e] += a[e] * hreg[jb][4 * g + e][ne];   // tuning build: same traffic, no matrix work
                    }
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ju = (r & 3) + 8 * (r >> 2);
            float o[NT];
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) {
                const float q = mu_quot(sreg[r][ne], acc[ne][r] + eps);
                float v = hreg[ob][r][ne] * q;
                if (clamp) v = fmaxf(v, eps);
                o[ne] = v;
            }
            buf_store<NT, AUXS>(o, hrs[ob], vo(hoff, 32 * ob + ju), ju * ldh4);
        }
    }
}

// Workgroups walk the column tiles grid-stride (G is staged once per workgroup, not once per tile).  EDGE = false is the
// kernel for k == KP and n a whole number of tiles: it contains the interior tile code only (the edge tile's per-row offset
// selects cost ~40 registers, i.e. one to two waves per SIMD).
// Round 4, k = 128: the staged G is 66 KiB there, so the 160 KiB of LDS hold TWO workgroups per CU (2 waves per SIMD whatever the
// registers allow), and with one tile per wave every workgroup staged its 64 KiB of G for four tiles: MFMA busy 50 % at an
// unthrottled 2.28 GHz (profiles/r04b_elt128_*).  The launch now caps the grid at 1024 workgroups (two resident rounds), each
// walking its share of the tiles with ONE staging of G: 3.55 -> 4.25 TB/s (H; 4.55 with the two-column tile at two waves per
// SIMD, csrc/dnmf.hip dnmf_mu_update_h), 4.14 -> 4.46 TB/s (W) on the 6.4 GB pass.
// NWV = waves per workgroup (a template parameter since then): 8-wave workgroups sharing one G measured the same for H and
// 3-8 % slower for W than the capped 4-wave grid, so 4 stays.
constexpr int upd_waves(int /*kt*/) { return 4; }
// workgroups for `tiles` wave tiles: one tile per wave at k <= 64 (measured best: G is at most 16 KiB), capped at k = 128
inline unsigned upd_grid(long tiles, int kt) {
    const long cap = tune("DNMF_UPD_GRID", kt == 4 ? 1024 : (1L << 30));
    return (unsigned)std::min<long>(cdiv(tiles, upd_waves(kt)), cap);
}

template <int KT, int NT, int OCC, bool EDGE, bool MMA = true, int AUXL = 0, int AUXS = 0, int NWV = upd_waves(KT)>
__global__ __launch_bounds__(64 * NWV, EDGE ? edge_occ(KT, OCC) : OCC) void update_h_seq_kernel(float* __restrict__ H, int k, long n, long ldh,
                                                                const float* __restrict__ Sm, long lds_,
                                                                const float* __restrict__ G, float eps, int clamp, BatchTab bt) {
    REBASE(H); REBASE(Sm); REBASE(G);
    constexpr int KP = 32 * KT, GP = KP + 4;
    extern __shared__ __attribute__((aligned(16))) float gs[];
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int idx = threadIdx.x; idx < KP * KP / 4; idx += 64 * NWV) {
        const int gr = idx / (KP / 4), gc = (idx % (KP / 4)) * 4;
        *reinterpret_cast<f32x4*>(&gs[gr * GP + gc]) = *reinterpret_cast<const f32x4*>(G + gr * KP + gc);
    }
    __syncthreads();
    const long ntiles = cdiv(n, 32 * NT);
    for (long t = (long)blockIdx.x * NWV + wid; t < ntiles; t += (long)gridDim.x * NWV) {
        const long col0 = t * 32 * NT;
        if constexpr (!EDGE) update_h_seq_tile<KT, NT, true, MMA, AUXL, AUXS>(H, k, n, ldh, Sm, lds_, gs, eps, clamp, col0, li, h);
        else if (k == KP && col0 + 32 * NT <= n) update_h_seq_tile<KT, NT, true>(H, k, n, ldh, Sm, lds_, gs, eps, clamp, col0, li, h);
        else update_h_seq_tile<KT, NT, false>(H, k, n, ldh, Sm, lds_, gs, eps, clamp, col0, li, h);
    }
}

// W: wave tile = 32 rows x KP columns, lane (li, h) owns row row0 + li; wreg[s][e] = W[i][8s + 4h + e] (as above).  The
// descriptor is re-based to the tile's first row; the lane offset (li ldw + 4h) is shared by all pieces, the piece index
// is an immediate.  V = 4: 16-byte pieces (aligned rows, k % 4 == 0); V = 1: the same tile with dword accesses and a
// per-element column predicate (any k, any alignment).
// MODE UW_MU: the multiplicative update, in place.  MODE UW_HALS_T: the first pass of the HALS W sweep (dnmf_hals.h) --
// T[i][j] = S[i][j] - (W G')[i][j] with the staged G' = G masked to l > j, i.e. T[i][j] = AH[i][j] - sum_{l > j} W[i][l] G[l][j],
// written to a separate buffer; W is only read.
enum { UW_MU = 0, UW_HALS_T = 1 };

template <int KT, int V, bool INTERIOR, int MODE = UW_MU, int AUX = 0>
__device__ __forceinline__ void update_w_seq_tile(float* __restrict__ W, long m, int k, long ldw, const float* __restrict__ Sm,
                                                  long lds_, const float* gs, float eps, long row0, int li, int h,
                                                  float* __restrict__ T = nullptr, long ldt = 0) {
    constexpr int KP = 32 * KT, GP = KP + 4;
    const i32x4 wrs = buf_rsrc(W + row0 * ldw), srs = buf_rsrc(Sm + row0 * lds_);
    const i32x4 trs = MODE == UW_HALS_T ? buf_rsrc(T + row0 * ldt) : wrs;
    int woff = (int)(((long)li * ldw + 4 * h) * 4), soff = (int)(((long)li * lds_ + 4 * h) * 4);
    int toff = MODE == UW_HALS_T ? (int)(((long)li * ldt + 4 * h) * 4) : 0;
    if constexpr (!INTERIOR) {
        if (row0 + li >= m) { woff = BUF_OOB; soff = BUF_OOB; toff = BUF_OOB; }
    }
    // opaque to the optimiser: otherwise (x << 2) + 32 s is rewritten as (x + 8 s) << 2 and no longer folds into the
    // instruction's immediate offset (one address register per piece instead of one per tile)
    asm volatile("" : "+v"(woff));
    asm volatile("" : "+v"(soff));
    if constexpr (MODE == UW_HALS_T) asm volatile("" : "+v"(toff));
    else toff = woff;                        // (the laundered value: the stores fold their piece offsets like the loads)
    // piece (s, h) = columns 8s + 4h .. +3; the edge variant switches off columns >= k
    // The piece offset 32 s goes into the instruction's IMMEDIATE offset (a constant added to the lane offset), never into
    // the SGPR offset: on gfx950 a 16-byte buffer store with an SGPR offset can still be reading its data registers when
    // the next VALU instruction overwrites them, and hipcc only pads that hazard for stores WITHOUT an SGPR offset
    // (measured: lanes 12-15 / 28-31 of a store picked up the following group's denominators).
    auto ld4 = [&](float (&d)[4], i32x4 rs, int off, int s) {
        if constexpr (V == 4) {
            buf_load<4, AUX>(d, rs, ((INTERIOR || 8 * s + 4 * h < k) ? off : BUF_OOB) + 32 * s, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = buf_ld_f32(rs, ((8 * s + 4 * h + e < k) ? off : BUF_OOB) + 32 * s + 4 * e, 0, 0);
        }
    };
    auto st4 = [&](const float (&d)[4], i32x4 rs, int off, int s) {
        if constexpr (V == 4) {
            buf_store<4, AUX>(d, rs, ((INTERIOR || 8 * s + 4 * h < k) ? off : BUF_OOB) + 32 * s, 0);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) buf_st_f32(d[e], rs, ((8 * s + 4 * h + e < k) ? off : BUF_OOB) + 32 * s + 4 * e, 0, 0);
        }
    };
    float wreg[4 * KT][4];
#pragma unroll
    for (int s = 0; s < 4 * KT; ++s) ld4(wreg[s], wrs, woff, s);
#pragma unroll
    for (int jt = 0; jt < KT; ++jt) {
        float sreg[4][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) ld4(sreg[g], srs, soff, 4 * jt + g);
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
            for (int e = 0; e < 4; ++e) {
                if constexpr (MODE == UW_HALS_T) o[e] = sreg[g][e] - out[4 * g + e];
                else o[e] = wreg[4 * jt + g][e] * mu_quot(sreg[g][e], out[4 * g + e] + eps);
            }
            st4(o, trs, toff, 4 * jt + g);
        }
    }
}

template <int KT, int V, int OCC, bool EDGE, int MODE = UW_MU, int AUX = 0, int NWV = upd_waves(KT)>
__global__ __launch_bounds__(64 * NWV, (EDGE || V == 1) ? edge_occ(KT, OCC) : OCC) void update_w_seq_kernel(float* __restrict__ W, long m, int k, long ldw,
                                                                const float* __restrict__ Sm, long lds_,
                                                                const float* __restrict__ G, float eps,
                                                                float* __restrict__ T, long ldt, BatchTab bt) {
    REBASE(W); REBASE(Sm); REBASE(G); REBASE(T);
    constexpr int KP = 32 * KT, GP = KP + 4;
    extern __shared__ __attribute__((aligned(16))) float gs[];
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int idx = threadIdx.x; idx < KP * KP / 4; idx += 64 * NWV) {
        const int gr = idx / (KP / 4), gc = (idx % (KP / 4)) * 4;
        f32x4 g = *reinterpret_cast<const f32x4*>(G + gr * KP + gc);
        if constexpr (MODE == UW_HALS_T) {      // gs[j][l] feeds output column j with contraction index l: keep l > j
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = (gc + e > gr) ? g[e] : 0.f;
        }
        *reinterpret_cast<f32x4*>(&gs[gr * GP + gc]) = g;
    }
    __syncthreads();
    const long ntiles = cdiv(m, 32);
    for (long t = (long)blockIdx.x * NWV + wid; t < ntiles; t += (long)gridDim.x * NWV) {
        const long row0 = t * 32;
        if constexpr (!EDGE) update_w_seq_tile<KT, V, true, MODE, AUX>(W, m, k, ldw, Sm, lds_, gs, eps, row0, li, h, T, ldt);
        else if (k == KP && row0 + 32 <= m) update_w_seq_tile<KT, V, true, MODE>(W, m, k, ldw, Sm, lds_, gs, eps, row0, li, h, T, ldt);
        else update_w_seq_tile<KT, V, false, MODE>(W, m, k, ldw, Sm, lds_, gs, eps, row0, li, h, T, ldt);
    }
}

// =============================================================================================== W update, 16-row wave tiles
// Round 5.  update_w_seq_kernel puts 32 rows of W across the 32 lanes of a half-wave, so one 16-byte load
// instruction touches 32 rows x 32 bytes: every 128-byte line of W and S is completed by FOUR instructions, and with 20 waves'
// tiles in flight per CU the line is gone from the 32 KiB L1 before the fourth arrives (L2 hit rate 0.68 in profiles/r04c_elt_*:
// two of three re-touches are L2 requests).  The kernel is bound by that request stream + its MFMAs back to back: 520 us of
// traffic (what the MFMA-free ew_kernel<4> needs for the same bytes) + 221 us of matrix work = the 755 us measured.
// Here a wave tile is 16 rows on v_mfma_f32_16x16x4_f32: lane (i, q) owns row i and the 16-byte pieces at columns 16 c + 4 q, so an
// instruction touches 16 rows x 64 bytes -- a line is completed by TWO instructions, issued back to back.  Same arithmetic
// order per element? No: the contraction order inside a row differs from the 32-row kernel (sums agree to fp32 rounding; both are
// pinned against float64 in tests/test_gpu_kernels.py::test_mu_updates).
//   D[j][i] = sum_l G[j][l] W[i][l]:  A operand lane (j = l & 15, q) = G[16 jt + j][16 c + 4 q + e], B operand lane (i, q) =
//   W[i][16 c + 4 q + e] (e = 0..3: four steps per 16-byte piece), C/D lane (i, q) register r = (W G)[i][16 jt + 4 q + r] -- the
//   piece (jt, q) of row i: the register that fed the product is the value the epilogue scales, as in the 32-row kernel.
#define MFMA16F(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
template <int KT, int NWV, int OCC>
__global__ __launch_bounds__(64 * NWV, OCC) void update_w16_kernel(float* __restrict__ W, long m, long ldw, const float* __restrict__ Sm,
                                                                   long lds_, const float* __restrict__ G, float eps, BatchTab bt) {
    REBASE(W); REBASE(Sm); REBASE(G);
    constexpr int KP = 32 * KT, GP = KP + 4, NC = KP / 16;
    extern __shared__ __attribute__((aligned(16))) float gs[];
    for (int idx = threadIdx.x; idx < KP * KP / 4; idx += 64 * NWV) {
        const int gr = idx / (KP / 4), gc = (idx % (KP / 4)) * 4;
        *reinterpret_cast<f32x4*>(&gs[gr * GP + gc]) = *reinterpret_cast<const f32x4*>(G + gr * KP + gc);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long ntiles = m / 16;                                    // (m % 16 == 0: the host checks)
    for (long t = (long)blockIdx.x * NWV + wid; t < ntiles; t += (long)gridDim.x * NWV) {
        const long row0 = t * 16;
        const i32x4 wrs = buf_rsrc(W + row0 * ldw), srs = buf_rsrc(Sm + row0 * lds_);
        int woff = (int)(((long)i * ldw + 4 * q) * 4), soff = (int)(((long)i * lds_ + 4 * q) * 4);
        asm volatile("" : "+v"(woff));                             // (keeps the piece offsets in the instructions' immediates)
        asm volatile("" : "+v"(soff));
        float w[NC][4];
#pragma unroll
        for (int c = 0; c < NC; ++c) buf_load<4>(w[c], wrs, woff + 64 * c, 0);
#pragma unroll
        for (int jt = 0; jt < NC; ++jt) {
            float sv[4];
            buf_load<4>(sv, srs, soff + 64 * jt, 0);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float a[4];
                load_vec_raw<4>(a, &gs[(16 * jt + i) * GP + 16 * c + 4 * q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = MFMA16F(a[e], w[c][e], acc);
            }
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = w[jt][r] * mu_quot(sv[r], acc[r] + eps);
            buf_store<4>(o, wrs, woff + 64 * jt, 0);
        }
    }
}

// (The H-side mirror of this kernel -- KP rows x 32 columns per wave, lane (i, q) owning two columns and the rows 16 t + 4 q + e, 8-wave
// workgroups sharing the 66 KiB of G at k = 128 -- was built and measured in round 5: correct, and SLOWER than update_h_seq_kernel<4, 2,
// 2> on the 6.4 GB pass (1.445-1.467 ms against 1.406: the H kernel's loads were whole lines already, and its 32x32x2 tiles read G
// from LDS half as often per flop).  Not kept.)

}  // namespace
