// dnmf_hals.h -- HALS / Frobenius column and row sweeps.
// Part of libdnmf_hip.so (kernels live in anonymous namespaces of the headers; the translation units csrc/*.hip include what they launch).
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
                                                        double* __restrict__ ss2_out, BatchTab bt) {
    REBASE(W); REBASE(AH); REBASE(G); REBASE(prev_ss2); REBASE(ss2_out);
    __shared__ float gcol[DNMF_MAX_K];      // (any rank up to DNMF_MAX_K = 256)
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
                                                          const double* __restrict__ ss2, BatchTab bt) {
    REBASE(W); REBASE(ss2);
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
                                                    const float* __restrict__ G, float eps, BatchTab bt) {
    REBASE(H); REBASE(AtW); REBASE(G);
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
                                                       const float* __restrict__ G, int kp, float eps, BatchTab bt) {
    REBASE(H); REBASE(AtW); REBASE(G);
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

namespace {

// =============================================================================================== persistent W sweep
// The W sweep of a rank whose column norms are local (p_r == 1) in TWO launches instead of k.  The per-column kernels above
// re-read every row of W for every column (k m k 4 bytes: 4.3 GB per sweep at m = 262144, k = 64 against 67 MB of factor) and
// pay a launch per column.  Here:
//   1. an MFMA pass (update_w_seq_kernel<.., UW_HALS_T>, dnmf_update.h) forms T[i][j] = AH[i][j] - sum_{l > j} W_old[i][l] G[l][j]
//      -- the old columns' part of every column step, a masked W G product;
//   2. hals_w_sweep_kernel: one lane owns one row of T for the whole sweep and keeps it in registers; for kk = 0 .. k-1:
//      u = max(t[kk], eps);  the column's sum of squares over ALL rows (grid-wide, see below);  w = u / sqrt(ss2) (if
//      ss2 > 0);  t[kk] := w (the slot is dead: it now holds the result);  t[j] -= w G[kk][j] for j > kk -- first j = kk+1
//      alone (the next column's critical path), the rest after the next column's partial sum is on its way, so the
//      triangular update overlaps the wait;  finally the row (now W_new) is stored.
// This is the reference's column step with its cancelling pair removed: W[i][kk] G[kk][kk] + AH[i][kk] - (W G)[i][kk]
// (dist_nmf.py:887) = AH[i][kk] - sum_{l != kk} W[i][l] G[l][kk]; l < kk are the new columns, l > kk the old ones.
// The column steps are function templates expanded by a fold expression: a `#pragma unroll` loop over the columns is NOT
// unrolled by hipcc (body too large), the row then lives in scratch memory and every column costs 6.5 us even on a
// single workgroup (measured).  G rows come from LDS (staged once per workgroup) as broadcast 16-byte reads.  (A first
// version also did step 1 in this kernel, row-wise on the VALU: hipcc could not allocate registers for it -- thousands
// of spills in every formulation tried.)
//
// Grid-wide sum per column WITHOUT a separate barrier: every workgroup publishes its fp64 partial sum into its own slot
// slab[kk][wg] (the slab is preset to all-ones bit patterns = "not there yet") with an agent-scope store; four waves of
// every workgroup poll the slots of the column (all of a lane's slots requested in one round), and all reduce them in a
// FIXED order -- the result is bitwise identical in every workgroup and from run to run (no floating-point atomics).
// One memory round trip per column.  All workgroups must be co-resident (the host checks the occupancy and otherwise
// takes the per-column path).
// (HALS_WG / HALS_MAX_WG: dnmf_common.h)
constexpr unsigned long long HALS_EMPTY = ~0ull;

__device__ __forceinline__ double dshfl_xor(double v, int mask) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return __hiloint2double(hi, lo);
}

// partner exchange inside a row of 16 lanes with DPP (a register-to-register move, no LDS crossbar round trip):
// quad_perm [1,0,3,2] / [2,3,0,1] pair lanes inside a quad, row_half_mirror pairs the two quads of 8 lanes, row_mirror
// the two halves of the row.  Every pairing is symmetric, so after the four steps all 16 lanes hold the same sum.
template <int CTRL>
__device__ __forceinline__ double ddpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// wave-wide sum, identical in all 64 lanes (symmetric butterfly: a + b == b + a at every step)
__device__ __forceinline__ double wave_sum_all(double v) {
    v += ddpp<0xB1>(v);      // quad_perm [1,0,3,2]
    v += ddpp<0x4E>(v);      // quad_perm [2,3,0,1]
    v += ddpp<0x141>(v);     // row_half_mirror
    v += ddpp<0x140>(v);     // row_mirror
    v += dshfl_xor(v, 16);
    v += dshfl_xor(v, 32);
    return v;
}

// sum of one value per thread over the workgroup, the same association order everywhere; result in every thread
__device__ __forceinline__ double block_sum_fixed(double v, double* red) {
    v = wave_sum_all(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();                           // red[] may still be read from the previous use
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < HALS_WG / 64; ++w) s += red[w];
    return s;
}

// One wave collects the nwg <= 64 NQ slots of a column: every lane keeps NQ slots, ALL of them are requested back to back in
// each polling round (independent, unconditional loads -- a loop that waits for one slot after the other pays one memory
// round trip per slot: measured 16 us per column on 512 workgroups), slots that are still empty are asked for again.
// Returns the lane's sum in slot order.
// Sticky per-device word: set when a polling wave of the persistent sweep gave up (below).  The host reads and clears it
// with dnmf_hals_sweep_status (PyNMF does at the end of a HALS fit), so a sweep that lost its co-residency surfaces as an
// error instead of as NaN factors.
__device__ unsigned int g_hals_timeout = 0;

template <int NQ>
__device__ __forceinline__ double hals_poll(const unsigned long long* col, int nwg, unsigned long long patience = 0) {
    const int lane = threadIdx.x & 63;
    const bool sys = patience != 0;          // cross-rank sweep: the slots are written by peers (system scope), waits are bounded in time
    unsigned long long bits[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) bits[q] = (lane + 64 * q < nwg) ? HALS_EMPTY : 0ull;   // beyond the grid: +0.0, never awaited
    // Safety valve: a sweep whose workgroups are not all resident (two sweeps sharing the device, which the host-side
    // occupancy check cannot see) would wait forever.  After ~2^21 polling rounds (about a second) the wave gives up, sets
    // the sticky word g_hals_timeout (dnmf_hals_sweep_status reports it to the host) and lets the missing slots keep their
    // "empty" pattern, which is a NaN: the column norm -- and with it W -- turns NaN instead of the GPU hanging.
    bool complete = false;
    const unsigned long long t0 = sys ? wall_clock64() : 0ull;
    // a sweep on this device has already given up (sticky word, cleared by dnmf_hals_sweep_status): its factors are lost, and so are
    // the ones computed since -- do not spend a second per column of every later sweep waiting again (one missing slot then costs one
    // wait per fit, not k per iteration)
    const bool lost = __hip_atomic_load(&g_hals_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    for (unsigned spins = 0; !lost && (sys || spins < (1u << 21)); ++spins) {
        unsigned long long v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int s = lane + 64 * q;
            v[q] = sys ? __hip_atomic_load(col + (s < nwg ? s : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                       : __hip_atomic_load(col + (s < nwg ? s : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (sys && (spins & 1023) == 1023 && wall_clock64() - t0 > patience) break;     // a peer is gone: say so (g_hals_timeout)
        bool missing = false;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            bits[q] = bits[q] == HALS_EMPTY ? v[q] : bits[q];
            missing = missing || bits[q] == HALS_EMPTY;
        }
        if (!__any(missing)) { complete = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    if (!complete && lane == 0) __hip_atomic_store(&g_hals_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) s += __longlong_as_double((long long)bits[q]);
    return s;
}

// Column KK of the sweep (see hals_w_sweep_kernel).  A function template per column, expanded by a fold expression: a
// `#pragma unroll` loop over the columns is not unrolled by hipcc (body too large), the row then lives in scratch memory
// and every column costs 6.5 us even on a single workgroup.
template <int KP, int KK>
__device__ __forceinline__ void hals_col_step(float (&t)[KP], float& u, int k, bool live, float eps, const float* gs,
                                              unsigned long long* __restrict__ slab, double* __restrict__ ss2_out,
                                              double* red, int nwg, int dbg, const HalsPeers& pe) {
    asm volatile("" ::: "memory");
    if (KK >= k) return;                                  // uniform
    const double sq = live ? (double)u * (double)u : 0.0;
    const double part = block_sum_fixed(sq, red);
    unsigned long long* col = (pe.P ? pe.slab[pe.rank] : slab) + (long)KK * HALS_MAX_WG;
    {
        unsigned long long bits = (unsigned long long)__double_as_longlong(part);
        if (bits == HALS_EMPTY) bits = 0x7ff8000000000000ull;            // (a NaN with that payload cannot occur)
        if (pe.P == 0) {
            if (threadIdx.x == 0) __hip_atomic_store(col + blockIdx.x, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if ((int)threadIdx.x < pe.P) {             // thread q: this workgroup's slot in rank q's slab (the value is its own flag)
            __hip_atomic_store(pe.slab[threadIdx.x] + (long)KK * HALS_MAX_WG + pe.first + blockIdx.x, bits, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (pe.P) nwg = pe.total;                             // all ranks' workgroups, in rank order: the same sum on every rank
    // the one entry of G on the next column's critical path is read before the wait
    float g_next = 0.f;
    if constexpr (KK + 1 < KP) g_next = gs[KK * KP + KK + 1];
    // lagging part of the previous column's triangular update: overlaps the round trip of the partial sums
    if constexpr (KK > 0) {
        const float wp = t[KK - 1];
#pragma unroll
        for (int j0 = 4 * ((KK + 1) / 4); j0 < KP; j0 += 4) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(&gs[(KK - 1) * KP + j0]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (j0 + q > KK) t[j0 + q] = fmaf(-wp, g[q], t[j0 + q]);
            if ((j0 / 4) % 4 == 3) __builtin_amdgcn_sched_barrier(0);   // at most four pieces of the G row in flight
        }
    }
    // the first four waves of the workgroup fetch the slots of this column (wave w: slots 256 w .. 256 w + 255)
    double mine = 0.0;
    if (dbg & 1) {                                         // tuning build only: no exchange (each workgroup on its own)
        if (threadIdx.x == 0) mine = part;
    } else if (threadIdx.x < 256) {
        const int base = (threadIdx.x >> 6) * 256, cnt = nwg - base;
        if (cnt > 0) {
            const unsigned long long pat = pe.P ? pe.patience : 0ull;
            if (cnt <= 64) mine = hals_poll<1>(col + base, cnt, pat);
            else if (cnt <= 128) mine = hals_poll<2>(col + base, cnt, pat);
            else mine = hals_poll<4>(col + base, cnt, pat);
        }
    }
    const double ss2 = block_sum_fixed(mine, red);
    if (blockIdx.x == 0 && threadIdx.x == 0 && ss2_out) ss2_out[KK] = ss2;
    const float ss = (float)sqrt(ss2);
    const float w = ss > 0.f ? u / ss : u;
    t[KK] = w;
    if constexpr (KK + 1 < KP) {
        t[KK + 1] = fmaf(-w, g_next, t[KK + 1]);
        u = fmaxf(t[KK + 1], eps);
    }
}

template <int KP, int... Ks>
__device__ __forceinline__ void hals_sweep_all(float (&t)[KP], float& u, int k, bool live, float eps, const float* gs,
                                               unsigned long long* __restrict__ slab, double* __restrict__ ss2_out,
                                               double* red, int nwg, int dbg, const HalsPeers& pe, std::integer_sequence<int, Ks...>) {
    (hals_col_step<KP, Ks>(t, u, k, live, eps, gs, slab, ss2_out, red, nwg, dbg, pe), ...);
}

// waves per SIMD: what the row (KP registers) + the fp64 reductions + the slot polling hold; 8-wave workgroups, so the device keeps 4 / 4 / 2 x 256 CUs x 4 SIMDs x 64 rows = 262144 / 262144 / 131072 rows
// resident for KP = 32 / 64 / 128
template <int KP, bool VEC>
__global__ __launch_bounds__(HALS_WG, KP <= 64 ? 4 : 2) void hals_w_sweep_kernel(
    float* __restrict__ W, long m, int k, long ldw, const float* __restrict__ T, long ldt, const float* __restrict__ G,
    float eps, unsigned long long* __restrict__ slab, double* __restrict__ ss2_out, int dbg, HalsPeers pe, BatchTab bt) {
    REBASE(W); REBASE(T); REBASE(G); REBASE(slab); REBASE(ss2_out);
    __shared__ double red[HALS_WG / 64];
    // G (KP x KP, zero padded, symmetric) staged once per workgroup: every use is a row segment G[r][c0 .. c0+3] at a
    // wave-uniform address = a broadcast ds_read_b128
    extern __shared__ __attribute__((aligned(16))) float gs[];
    for (int idx = threadIdx.x; idx < KP * KP / 4; idx += HALS_WG)
        *reinterpret_cast<f32x4*>(&gs[idx * 4]) = *reinterpret_cast<const f32x4*>(G + idx * 4);
    __syncthreads();
    const long i = (long)blockIdx.x * HALS_WG + threadIdx.x;
    const bool live = i < m;
    const long ir = live ? i : m - 1;                 // dead lanes read a valid row and contribute nothing
    const int nwg = gridDim.x;
    float t[KP];
    const float* trow = T + ir * ldt;                  // 1. the row of T (the transform ran as its own MFMA kernel)
#pragma unroll
    for (int j = 0; j < KP; j += 4) {
        if constexpr (VEC) {
            if (j < k) {                              // VEC: k % 4 == 0, 16-byte aligned rows
                const f32x4 v = *reinterpret_cast<const f32x4*>(trow + j);
                t[j] = v[0]; t[j + 1] = v[1]; t[j + 2] = v[2]; t[j + 3] = v[3];
            } else { t[j] = t[j + 1] = t[j + 2] = t[j + 3] = 0.f; }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) t[j + e] = (j + e < k) ? trow[j + e] : 0.f;
        }
    }
    // 2. the column sweep
    float u = fmaxf(t[0], eps);
    hals_sweep_all<KP>(t, u, k, live, eps, gs, slab, ss2_out, red, nwg, dbg, pe, std::make_integer_sequence<int, KP>{});
    // 3. store the row
    if (live) {
        float* orow = W + i * ldw;
#pragma unroll
        for (int j = 0; j < KP; j += 4) {
            if constexpr (VEC) {
                if (j < k) *reinterpret_cast<f32x4*>(orow + j) = f32x4{t[j], t[j + 1], t[j + 2], t[j + 3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j + e < k) orow[j + e] = t[j + e];
            }
        }
    }
}

}  // namespace
