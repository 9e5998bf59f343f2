// dnmf_small.h -- whole fits of SMALL problems as ONE persistent kernel per batch (round 5): MU/KL (small_kl_fit_kernel, described first;
// small_kl_hfit_kernel when W is fixed), MU/FRO (small_fro_fit_kernel) and HALS (small_hals_fit_kernel).
// Part of libdnmf_hip.so (kernels live in anonymous namespaces of the headers; csrc/dnmf_fit.hip plans and launches them).
//
// Why: the reference's own examples factorise 1024 x 256 matrices (examples/dist_pynmfk_2d_Swim.py: k = 14..18, 5000 KL steps per
// fit, 20 perturbations per k).  At that size a step is ~11 dependent launches of 4-12 us each, whatever the batch: 90 us per
// batched step, host-bound and latency-bound at once (tools/dbg/fitgap.py under rocprofv3).  Here a problem's rows are cut into slabs of R = 16 NW
// rows, one workgroup per slab, all workgroups of all problems of a batch resident at once; a workgroup keeps its slab of A, its
// rows of W and the whole of H in LDS for the entire fit and runs the iterations itself:
//
//   W phase (dist_nmf.py:806,810,828-830), local to the slab: wave w owns 16 rows.  Per 16-column tile: S^T = H^T W^T on
//     v_mfma_f32_16x16x4_f32 -- its C registers are U = A / (S + eps) for (row = lane, four consecutive columns), which is exactly the
//     A operand of the next product U H^T (the contraction runs over those columns) -- accumulated over the tiles, then
//     W *= U H^T / (rowsum(H) + eps).
//   H phase (:806,808,847-849): wave w owns the column tiles w, w + NW, ...; per 16-row tile S = W H with the operands swapped, so that
//     U lands as (column = lane, four consecutive rows) = the B operand of W^T U.  The slab's k x n partial goes to global memory.
//   grid barrier over the problem's workgroups; every workgroup sums the partials (fixed order: the same sum for any grid) for its share of
//     H's elements, updates them (clamped every tenth step, pyDNMF.py:155-156), writes them; second barrier; everybody re-reads H.
//
// Two barriers per step among the <= 64 workgroups of ONE problem (an arrival counter per problem; the exchanged data move as
// device-scope atomic stores / loads, see st_dev); waits are bounded by the wall clock and report through a sticky word, as the
// persistent HALS sweep does.
// Measured (20 problems of 1024 x 256, k <= 16, 8 workgroups of 8 waves each; tuning build with phases switched off): a step is 19 us
// = W phase 6-7 us + H phase 6-7 us (4.2 MFLOP per workgroup and step on one CU's matrix pipe: 160 of the 256 CUs are in use) + 6 us
// of H update, re-read of H and row / column sums + 2 us for the two barriers; the per-step path needs 90 us for the same step.
// The sums are fp32 MFMA accumulations in a different association than the big kernels' (dnmf_kl16.h): results agree with the step
// path to fp32 rounding, not bit for bit (tests/test_gpu_fit.py pins both against float64).
#pragma once
#include "dnmf_common.h"

namespace {

#define SM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct SmallKlArgs {
    const float* A; long lda, a_stride;          // strides between the problems of a batch, in elements
    float* W; long ldw, w_stride;
    float* H; long ldh, h_stride;
    int m, n, k;
    float eps;
    int itr, w_update;
    float* part; long part_stride;               // per problem: [P][KP][NS] partial W^T U, then [P][KP] column sums of the slabs' W
    unsigned* bar; long bar_stride;              // per problem: arrival counter (zeroed before the launch)
    unsigned long long patience;                 // ticks of the 100 MHz wall clock a barrier may wait
    int z0;                                      // first problem of this launch
    unsigned* slots; long slots_stride;          // HALS: per problem [2][KP][P] column-norm partials (float bits; SLOT_EMPTY = not there yet)
    int cw;                                      // HALS: columns of H a workgroup sweeps (ceil(NS / P))
    float* hg; long hg_stride;                   // MU fits: per problem [KP][NS] granules {H element, step tag} (zeroed before the launch)
};

__device__ unsigned int g_small_timeout = 0;     // sticky: a barrier of a persistent fit gave up (dnmf_hals_sweep_status reports it)

// sum over the 16 lanes of a DPP row (lanes 16 q .. 16 q + 15), the same value in all of them, fixed association
// four quotients a / (d + eps) with the additions and products as PACKED fp32 instructions (v_pk_add_f32 / v_pk_mul_f32: two lanes'
// worth per issue; the fp32 vector instructions of the KL quotient cost about half the time of the step's MFMAs at k <= 16, none of it
// overlaps them); the same IEEE operations as the scalar form, so the same bits
typedef float f32x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 quot4(f32x4 a, f32x4 d, float eps) {
    const f32x2s e2 = {eps, eps};
    f32x2s lo = f32x2s{d[0], d[1]} + e2, hi = f32x2s{d[2], d[3]} + e2;
    lo = f32x2s{__builtin_amdgcn_rcpf(lo[0]), __builtin_amdgcn_rcpf(lo[1])};
    hi = f32x2s{__builtin_amdgcn_rcpf(hi[0]), __builtin_amdgcn_rcpf(hi[1])};
    lo = f32x2s{a[0], a[1]} * lo;
    hi = f32x2s{a[2], a[3]} * hi;
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

__device__ __forceinline__ float row16_sum(float v) {
    auto dpp = [](float x, auto ctrl) {
        const int b = __builtin_bit_cast(int, x);
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, decltype(ctrl)::value, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>{});     // row_mirror
    return v;
}

// Ordering contract of the in-kernel exchanges of this file: none of the atomics below carries release / acquire semantics --
// the hand-off is the form MI355X_MICROARCH.md lists as measured-valid on gfx950 ("Valid forms", row 1 of its table): EVERY handed-off
// byte is written by a write-through (sc1) store and read by an L1-bypassing (sc1) load (st_dev / ld_dev), every storing wave drains its
// stores (s_waitcnt 0) before the workgroup barrier behind which ONE lane adds to the arrival counter, the consumer's polling lane reads
// only after its poll has matched and the other waves only after the workgroup barrier it then joins; hipMalloc memory, one workgroup
// per CU.  An agent-scope acquire after each barrier (buffer_inv sc1, ~1.7 us) would make the form architectural at +20 % per step; the
// guard below keeps the kernels from being built for a target the form has not been measured on.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "csrc/dnmf_small.h: the sc1 store / sc1 load hand-off of the persistent kernels is validated on gfx950 (and gfx942) only"
#endif
// The data that cross workgroups inside the kernel (the partials, the updated H) move as relaxed DEVICE-scope atomic stores / loads:
// coherent at the device level by themselves (write-through, no stale lines), so the barrier needs no cache write-back / invalidate
// -- the first version bracketed the barriers with __threadfence() (an L2 write-back + invalidate each) and a step took 107 us.
__device__ __forceinline__ void st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one element of A as stored -> fp32 (bf16 -> fp32 is a 16-bit shift)
template <typename TA> __device__ __forceinline__ float sm_ld(const TA* p);
template <> __device__ __forceinline__ float sm_ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float sm_ld<bf16_t>(const bf16_t* p) { return __builtin_bit_cast(float, (unsigned)(*p) << 16); }

// arrive at / wait for the `gen`-th barrier of the problem's nwg workgroups (counter monotonic: gen nwg arrivals in all).  bar[1] is the
// problem's abort word (zeroed with the counter before the launch): the first wait that exceeds `patience` sets it and the sticky
// device word; every wait of the problem -- this one and all later ones, in every workgroup -- then returns at once, so a fit that lost
// its co-residency ends within one patience instead of one patience per barrier.  Its factors are garbage; the host raises.
__device__ __forceinline__ void small_barrier(unsigned* bar, unsigned target, unsigned long long patience) {
    __builtin_amdgcn_s_waitcnt(0);                         // this thread's device-scope stores have completed
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 15u) == 15u && __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (wall_clock64() - t0 > patience) {          // (the clock is read every round: the round trip of the counter dominates)
                __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&g_small_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

// the row sums of the rows a wave has just copied (lane partials rs[u] of row wv + u NW) -> xs
template <int NW, int RB>
__device__ __forceinline__ void hs_row_sums(const float (&rs)[RB], float* xs, int k, int wv, int lane) {
#pragma unroll
    for (int u = 0; u < RB; ++u) {
        float t = row16_sum(rs[u]);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if (lane == 0 && wv + u * NW < k) xs[wv + u * NW] = t;
    }
}
// H (k x n, written by the other workgroups of the problem with write-through stores; read here with sc1 loads) -> rows [0, k) of Hs
// (the rows [k, KP) are zeroed once, before the first step).  Wave wv takes the rows wv, wv + NW, ..., a lane four consecutive columns;
// every load of a pass is in flight before the first value is used and no index is divided (round 6: the element-indexed copy, eight
// loads in flight and a division per element, was 2.4 us of an 18.7 us step at k <= 16 and 5.6 of 29.3 at k = 17, tools/swimbench.py
// under DNMF_SMALL_ABL).
// xs != nullptr: also the row sums of H (a wave owns whole rows: lane partials, then the wave's reduction) -> xs[j], j < k.
template <int NW, int KP>
__device__ __forceinline__ void hs_fill(const float* H, long ldh, int k, int n, int NS, int LDH, float* Hs, int wv, int lane, float* xs = nullptr) {
    const i32x4 hd = buf_rsrc(H);
    constexpr int RB = (KP + NW - 1) / NW;                 // a wave's rows in one pass
    float rs[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) rs[u] = 0.f;
    for (int c0 = 0; c0 < NS; c0 += 256) {
        const int c = c0 + 4 * lane;
        for (int j0 = wv; j0 < k; j0 += NW * RB) {
            f32x4 v[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int j = j0 + u * NW;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[u][e] = buf_ld_f32(hd, (j < k && c + e < n) ? (int)((j * ldh + c + e) * 4) : BUF_OOB, 0, 16);
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int j = j0 + u * NW;
                if (j < k && c < NS) *reinterpret_cast<f32x4*>(&Hs[j * LDH + c]) = v[u];
                rs[u] += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
            }
        }
    }
    if (xs) hs_row_sums<NW, RB>(rs, xs, k, wv, lane);
}
// The same copy from the GRANULES of H: the owners of the H update publish every new element as one 8-byte {value, tag = step} store
// (MI355X_MICROARCH.md "R2 granule": an aligned 8-byte write-through store is observed whole), and a reader takes an element only with the
// tag of the step it is waiting for -- the data is its own flag, so the device-wide barrier that used to stand between the H update and this
// copy (1.2-1.5 us of a 19 us step) is gone.  No second buffer is needed: an owner overwrites an element for step t + 1 only after the
// partials of step t have ALL arrived, and a workgroup writes those after it has read the whole H of step t.  Waits are bounded like
// small_barrier's (bar[1] = the problem's abort word).
template <int NW, int KP>
__device__ __forceinline__ void hs_fill_granules(const float* Hg, int k, int n, int NS, int LDH, float* Hs, int wv, int lane, float want,
                                                 unsigned* bar, unsigned long long patience, float* xs = nullptr) {
    const i32x4 gd = buf_rsrc(Hg);
    constexpr int RB = (KP + NW - 1) / NW;
    float rs[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) rs[u] = 0.f;
    for (int c0 = 0; c0 < NS; c0 += 256) {
        const int c = c0 + 4 * lane;
        f32x2 g[RB][4];
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int j = wv + u * NW;
#pragma unroll
                for (int e = 0; e < 4; ++e) g[u][e] = buf_ld_f32x2(gd, (j < k && c + e < n) ? (j * NS + c + e) * 8 : BUF_OOB, 0, 16);
            }
            bool ok = true;
#pragma unroll
            for (int u = 0; u < RB; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) ok = ok && (!(wv + u * NW < k && c + e < n) || g[u][e][1] == want);
            if (__all(ok)) break;
            if ((++spins & 7u) == 7u && __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (wall_clock64() - t0 > patience) {
                __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&g_small_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int j = wv + u * NW;
            if (j < k && c < NS) *reinterpret_cast<f32x4*>(&Hs[j * LDH + c]) = f32x4{g[u][0][0], g[u][1][0], g[u][2][0], g[u][3][0]};
            rs[u] += (g[u][0][0] + g[u][1][0]) + (g[u][2][0] + g[u][3][0]);
        }
    }
    if (xs) hs_row_sums<NW, RB>(rs, xs, k, wv, lane);
}
__device__ __forceinline__ void hs_zero_tail(float* Hs, int k, int KP, int LDH, int tid, int T) {
    for (int idx = k * LDH + tid; idx < KP * LDH; idx += T) Hs[idx] = 0.f;
}

// ALDS: the slab of A lives in LDS for the whole fit.  !ALDS (k > 16 at the example sizes: slab + H + W do not fit 160 KiB with 128-row
// slabs, and 64-row slabs of 20 problems do not fit the device in one launch): A streams from the L2 / Infinity Cache -- a problem is
// ~1 MB -- with the next group's elements requested before this group's MFMAs; LDS then holds H and the slab's W only.
template <int KP, int NW, bool ALDS>
__global__ __launch_bounds__(64 * NW, 1) void small_kl_fit_kernel(SmallKlArgs a) {
    constexpr int JT = KP / 16, KS = KP / 4, R = 16 * NW, T = 64 * NW, LDW = KP + 1;
    const int z = a.z0 + blockIdx.z, p = blockIdx.x, P = gridDim.x;
    const float* __restrict__ A = a.A + (long)z * a.a_stride;
    float* __restrict__ W = a.W + (long)z * a.w_stride;
    float* H = a.H + (long)z * a.h_stride;
    float* part = a.part + (long)z * a.part_stride;
    unsigned* bar = a.bar + (long)z * a.bar_stride;
    float* Hg = a.hg + (long)z * a.hg_stride;             // [KP][NS] granules of H (hs_fill_granules)
    const int m = a.m, n = a.n, k = a.k;
    const float eps = a.eps;
    const int ksteps = (k + 3) >> 2;                      // contraction steps of 4 that hold real columns of W
#ifdef DNMF_TUNING
    const int abl = a.w_update >> 8;                       // ablations (WRONG results, timing only): 1 H without waiting for its granules, 2 no barrier before the H update, 4 no reload of H, 8 no H update
    a.w_update &= 1;
#else
    constexpr int abl = 0;
#endif
    const int NS = (n + 15) & ~15, nct = NS / 16, LDA = NS + 4, LDH = NS + 4;
    float* pcs = part + (long)P * KP * NS;                 // [P][KP] column sums of W per slab
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                      // [R][LDA]   the slab of A (zero beyond m / n); ALDS only
    float* Hs = As + (ALDS ? R * LDA : 0);                 // [KP][LDH]  all of H (zero beyond k / n)
    float* Ws = Hs + KP * LDH;                             // [R][LDW]   the slab's rows of W (zero beyond m / k)
    float* xs = Ws + R * LDW;                              // [KP]       row sums of H
    float* cs = xs + KP;                                   // [NW][KP]   column sums of W per wave
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long r0 = (long)p * R;

    if constexpr (ALDS) {
        for (int idx = tid; idx < R * NS; idx += T) {
            const int r = idx / NS, c = idx - r * NS;
            As[r * LDA + c] = (r0 + r < m && c < n) ? A[(r0 + r) * a.lda + c] : 0.f;
        }
    }
    // streamed A (!ALDS): four consecutive elements of the wave's row (W phase) / one element of each of four rows (H phase), zero outside
    const bool avec = (a.lda % 4 == 0) && (((unsigned long)A & 15) == 0);
    auto a_row4 = [&](int c) -> f32x4 {                    // A[r0 + 16 wv + i][c .. c + 3]
        const long row = r0 + 16 * wv + i;
        if (row < m && c + 4 <= n && avec) return *reinterpret_cast<const f32x4*>(A + row * a.lda + c);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (row < m && c + e < n) ? A[row * a.lda + c + e] : 0.f;
        return v;
    };
    auto a_col4 = [&](int rbase, int c) -> f32x4 {         // A[r0 + rbase + r][c], r = 0..3
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (r0 + rbase + e < m && c < n) ? A[(r0 + rbase + e) * a.lda + c] : 0.f;
        return v;
    };
    for (int idx = tid; idx < R * KP; idx += T) {
        const int r = idx / KP, j = idx - r * KP;
        Ws[r * LDW + j] = (r0 + r < m && j < k) ? W[(r0 + r) * a.ldw + j] : 0.f;
    }
    auto load_h = [&](float want) {                        // H -> LDS (want = 0: the caller's H; else the granules of that step), then its row sums
        if (want == 0.f) hs_fill<NW, KP>(H, a.ldh, k, n, NS, LDH, Hs, wv, lane, xs);
        else hs_fill_granules<NW, KP>(Hg, k, n, NS, LDH, Hs, wv, lane, want, bar, a.patience, xs);
        __syncthreads();
    };
    hs_zero_tail(Hs, k, KP, LDH, tid, T);
    if (tid >= k && tid < KP) xs[tid] = 0.f;
    load_h(0.f);
    // the wave's rows of W as an MFMA operand: lane (row i, q) holds W[16 wv + i][4 s + q]
    float wreg[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) wreg[s] = Ws[(16 * wv + i) * LDW + 4 * s + q];
    const bool rowok = r0 + 16 * wv + i < m;
    unsigned gen = 0;

    for (int it = 0; it < a.itr; ++it) {
        const bool clamp = (it % 10 == 0);
        if (a.w_update) {
            // ---------------------------------------------------------------- W phase: rows 16 wv .. 16 wv + 15
            // four column tiles at a time: four independent MFMA chains in flight (one tile after the other was a chain of dependent
            // MFMAs, LDS reads and reciprocals -- 5.9 us of a 21 us step); U H^T accumulates in two registers sets, added at the end
            f32x4 acc2[2][JT];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc2[h2][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 apre[4];                                 // (!ALDS) the next group's elements of A
            if constexpr (!ALDS) {
#pragma unroll
                for (int t = 0; t < 4; ++t) apre[t] = a_row4(16 * (t < nct ? t : nct - 1) + 4 * q);
            }
            auto wgroup = [&](int ct0, auto tail) __attribute__((always_inline)) {      // TAIL: fewer than four tiles left (a tile beyond the last one repeats it with U = 0)
                constexpr bool TAIL = decltype(tail)::value;
                f32x4 acur[4];
                if constexpr (!ALDS) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) acur[t] = apre[t];
                    if (ct0 + 4 < nct) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) apre[t] = a_row4(16 * (ct0 + 4 + t < nct ? ct0 + 4 + t : nct - 1) + 4 * q);
                    }
                }
                int c0[4];
                f32x4 d[4];                                // lane (row i, q) reg r = (W H)[row i][c0 + 4 q + r]
#pragma unroll
                for (int t = 0; t < 4; ++t) { c0[t] = 16 * (!TAIL || ct0 + t < nct ? ct0 + t : nct - 1); d[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                float hop[KS][4];                          // all operands of the group first (one block of LDS reads: a read behind the
#pragma unroll                                             // uniform `break` below would be waited for right in front of its MFMAs)
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int t = 0; t < 4; ++t) hop[s][t] = Hs[(4 * s + q) * LDH + c0[t] + i];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (s >= ksteps) break;                // (the zero-padded steps beyond k: uniform)
#pragma unroll
                    for (int t = 0; t < 4; ++t) d[t] = SM_MFMA(hop[s][t], wreg[s], d[t]);
                }
                f32x4 u[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f32x4 av;
                    if constexpr (ALDS) av = *reinterpret_cast<const f32x4*>(&As[(16 * wv + i) * LDA + c0[t] + 4 * q]);
                    else av = acur[t];
                    u[t] = quot4(av, d[t], eps);
                    if constexpr (TAIL) { if (!(ct0 + t < nct)) u[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                }
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)            // lane (j = i, q) reg r = (U H^T)[row 4 q + r][16 jt + i]
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const f32x4 hv = *reinterpret_cast<const f32x4*>(&Hs[(16 * jt + i) * LDH + c0[t] + 4 * q]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc2[t & 1][jt] = SM_MFMA(u[t][r], hv[r], acc2[t & 1][jt]);
                    }
            };
            {
                int ct0 = 0;
                for (; ct0 + 4 <= nct; ct0 += 4) wgroup(ct0, std::false_type{});
                if (ct0 < nct) wgroup(ct0, std::true_type{});
            }
            // U H^T goes through the wave's own rows of Ws into the operand layout (W itself is in wreg)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Ws[(16 * wv + 4 * q + r) * LDW + 16 * jt + i] = acc2[0][jt][r] + acc2[1][jt][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the tile is in LDS (same wave reads it back)
            __builtin_amdgcn_wave_barrier();
            float tt[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) tt[s] = Ws[(16 * wv + i) * LDW + 4 * s + q];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                wreg[s] = wreg[s] * (tt[s] * __builtin_amdgcn_rcpf(xs[4 * s + q] + eps));
                Ws[(16 * wv + i) * LDW + 4 * s + q] = wreg[s];
            }
        }
        // column sums of the slab's W (the H update divides by the sums over all rows, :847-849)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float v = row16_sum(wreg[s]);
            if (i == 0) cs[wv * KP + 4 * s + q] = v;
        }
        __syncthreads();
        if (tid < KP) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += cs[w * KP + tid];
            st_dev(&pcs[p * KP + tid], v);
        }
        // -------------------------------------------------------------------- H phase: column tiles wv, wv + NW, ...
        for (int ct = wv; ct < nct; ct += NW) {
            const int c0 = 16 * ct;
            f32x4 acc3[2][JT];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc3[h2][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 hpre[4];
            if constexpr (!ALDS) {
#pragma unroll
                for (int t = 0; t < 4; ++t) hpre[t] = a_col4(16 * t + 4 * q, c0 + i);
            }
            static_for<0, (NW + 3) / 4>([&](auto gc) {     // four row tiles at a time (independent chains, as in the W phase); NW = 6: four, then two
                constexpr int rt0 = 4 * decltype(gc)::value, C = NW - rt0 < 4 ? NW - rt0 : 4;
                f32x4 hcur[4];
                if constexpr (!ALDS) {
#pragma unroll
                    for (int t = 0; t < C; ++t) hcur[t] = hpre[t];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (rt0 + 4 + t < NW) hpre[t] = a_col4(16 * (rt0 + 4 + t) + 4 * q, c0 + i);
                }
                f32x4 d[4];                                // lane (col i, q) reg r = (W H)[16 rt + 4 q + r][c0 + i]
#pragma unroll
                for (int t = 0; t < C; ++t) d[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                float hbv[KS], wop[KS][4];                 // (operands first, as in the W phase)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    hbv[s] = Hs[(4 * s + q) * LDH + c0 + i];
#pragma unroll
                    for (int t = 0; t < C; ++t) wop[s][t] = Ws[(16 * (rt0 + t) + i) * LDW + 4 * s + q];
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    if (s >= ksteps) break;
#pragma unroll
                    for (int t = 0; t < C; ++t) d[t] = SM_MFMA(wop[s][t], hbv[s], d[t]);
                }
                f32x4 u[4];
#pragma unroll
                for (int t = 0; t < C; ++t) {
                    f32x4 av;
#pragma unroll
                    for (int r = 0; r < 4; ++r) av[r] = ALDS ? As[(16 * (rt0 + t) + 4 * q + r) * LDA + c0 + i] : hcur[t][r];
                    u[t] = quot4(av, d[t], eps);
                }
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)            // lane (col i, q) reg r = (W^T U)[16 jt + 4 q + r][c0 + i]
#pragma unroll
                    for (int t = 0; t < C; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc3[t & 1][jt] = SM_MFMA(Ws[(16 * (rt0 + t) + 4 * q + r) * LDW + 16 * jt + i], u[t][r], acc3[t & 1][jt]);
            });
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * jt + 4 * q + r < k) st_dev(&part[((long)p * KP + 16 * jt + 4 * q + r) * NS + c0 + i], acc3[0][jt][r] + acc3[1][jt][r]);   // (only the k real rows are read)
        }
        if (!(abl & 2)) small_barrier(bar, (unsigned)P * ++gen, a.patience);
        // -------------------------------------------------------------------- H update: this workgroup's share of the elements
        for (int e = p * T + tid; e < ((abl & 8) ? 0 : k * NS); e += P * T) {      // (the rows [k, KP) of H do not exist)
            const int j = e / NS, c = e - j * NS;
            float sum = 0.f, x = 0.f;                      // W^T U and the column sum of W over all slabs, slab order
            auto add = [&](auto wide, int g0) __attribute__((always_inline)) {      // `wide` partials of each in flight
                constexpr int WD = decltype(wide)::value;
                float v[WD], y[WD];
#pragma unroll
                for (int u = 0; u < WD; ++u) {
                    const int g = g0 + u < P ? g0 + u : P - 1;
                    v[u] = ld_dev(&part[((long)g * KP + j) * NS + c]);
                    y[u] = ld_dev(&pcs[g * KP + j]);
                }
#pragma unroll
                for (int u = 0; u < WD; ++u) { sum += (g0 + u < P) ? v[u] : 0.f; x += (g0 + u < P) ? y[u] : 0.f; }
            };
            if (P <= 8) add(std::integral_constant<int, 8>{}, 0);                      // (one round trip for the slabs of a problem)
            else for (int g0 = 0; g0 < P; g0 += 12) add(std::integral_constant<int, 12>{}, g0);
            float h = Hs[j * LDH + c] * (sum * __builtin_amdgcn_rcpf(x + eps));
            if (j < k && c < n) {
                if (clamp) h = fmaxf(h, eps);
                H[(long)j * a.ldh + c] = h;                                                    // (read by the host side only, after the launch)
                buf_st_f32x2(f32x2{h, (float)(it + 1)}, buf_rsrc(Hg), (j * NS + c) * 8, 0, 16);   // published: value and step in one granule
            }
        }
        __syncthreads();                                   // (every thread has read its old elements of Hs)
        if (!(abl & 4)) load_h((abl & 1) ? 0.f : (float)(it + 1));                              // (no device-wide barrier: the granules carry their step)
        if (clamp) {                                       // W = max(W, eps) after both updates (pyDNMF.py:155)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (rowok && 4 * s + q < k) wreg[s] = fmaxf(wreg[s], eps);
                Ws[(16 * wv + i) * LDW + 4 * s + q] = wreg[s];
            }
            __syncthreads();
        }
    }
    if (a.w_update || a.itr > 0) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
            if (rowok && 4 * s + q < k) W[(r0 + 16 * wv + i) * a.ldw + 4 * s + q] = wreg[s];
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The Frobenius twin (MU/FRO, dist_nmf.py:716-751): the same slabs, barriers and exchange, without the quotient.
//   W phase: A H^T over the column tiles (the wave's elements of A are the A operand as they come), G = H H^T by the first JT^2 waves
//     from the H in LDS (every workgroup the same G), W *= A H^T / (W G + eps) with W G on the matrix cores in the layout A H^T
//     accumulated in.
//   H phase: W^T A per column tile + the slab's W^T W (first JT^2 waves) -> global partials; barrier; every workgroup sums the Gram
//     partials (slab order), then for its share of H's elements the W^T A partials, and H *= W^T A / (G H + eps) with the k-term dot
//     product from LDS; barrier; re-read H.
// TA: the storage type of A (float, or bf16_t for params.precision = 'bfloat16'); ALDS: the slab in LDS as stored, else streamed from the L2
template <int KP, int NW, bool ALDS, typename TA>
__global__ __launch_bounds__(64 * NW, 1) void small_fro_fit_kernel(SmallKlArgs a) {
    constexpr int JT = KP / 16, KS = KP / 4, R = 16 * NW, T = 64 * NW, LDW = KP + 1, LDG = KP + 1;
    static_assert(JT * JT <= NW, "one wave per Gram tile");
    const int z = a.z0 + blockIdx.z, p = blockIdx.x, P = gridDim.x;
    const TA* __restrict__ A = reinterpret_cast<const TA*>(a.A) + (long)z * a.a_stride;
    float* __restrict__ W = a.W + (long)z * a.w_stride;
    float* H = a.H + (long)z * a.h_stride;
    float* part = a.part + (long)z * a.part_stride;
    unsigned* bar = a.bar + (long)z * a.bar_stride;
    float* Hg = a.hg + (long)z * a.hg_stride;             // [KP][NS] granules of H (hs_fill_granules)
    const int m = a.m, n = a.n, k = a.k;
    const float eps = a.eps;
    const int NS = (n + 15) & ~15, nct = NS / 16, LDA = NS + (sizeof(TA) == 4 ? 4 : 8), LDH = NS + 4;
    float* pg = part + (long)P * KP * NS;                  // [P][KP][KP] W^T W of the slabs
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                                      // [KP][LDH]
    float* Ws = Hs + KP * LDH;                             // [R][LDW]
    float* Gs = Ws + R * LDW;                              // [KP][LDG]  H H^T (W phase), then W^T W (H update)
    TA* As = reinterpret_cast<TA*>(Gs + ((KP * LDG + 3) & ~3));       // [R][LDA]   ALDS: the slab of A as stored (16-byte aligned rows)
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long r0 = (long)p * R;
    if constexpr (ALDS) {
        for (int idx = tid; idx < R * NS; idx += T) {
            const int r = idx / NS, c = idx - r * NS;
            As[r * LDA + c] = (r0 + r < m && c < n) ? A[(r0 + r) * a.lda + c] : TA(0);
        }
    }
    const bool avec = (a.lda % 4 == 0) && (((unsigned long)A & (4 * sizeof(TA) - 1)) == 0);
    auto a_row4 = [&](int c) -> f32x4 {                    // A[r0 + 16 wv + i][c .. c + 3] (zero outside)
        f32x4 v;
        if constexpr (ALDS) {
            if constexpr (sizeof(TA) == 4) v = *reinterpret_cast<const f32x4*>(&As[(16 * wv + i) * LDA + c]);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = sm_ld<TA>(&As[(16 * wv + i) * LDA + c + e]);
            }
            return v;
        }
        const long row = r0 + 16 * wv + i;
        if (row < m && c + 4 <= n && avec) {
            if constexpr (sizeof(TA) == 4) v = *reinterpret_cast<const f32x4*>(A + row * a.lda + c);
            else {
                const uint2 raw = *reinterpret_cast<const uint2*>(A + row * a.lda + c);
                v[0] = __builtin_bit_cast(float, raw.x << 16); v[1] = __builtin_bit_cast(float, raw.x & 0xffff0000u);
                v[2] = __builtin_bit_cast(float, raw.y << 16); v[3] = __builtin_bit_cast(float, raw.y & 0xffff0000u);
            }
            return v;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (row < m && c + e < n) ? sm_ld<TA>(A + row * a.lda + c + e) : 0.f;
        return v;
    };
    auto a_col4 = [&](int rbase, int c) -> f32x4 {         // A[r0 + rbase + r][c], r = 0..3
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if constexpr (ALDS) v[e] = sm_ld<TA>(&As[(rbase + e) * LDA + c]);
            else v[e] = (r0 + rbase + e < m && c < n) ? sm_ld<TA>(A + (r0 + rbase + e) * a.lda + c) : 0.f;
        }
        return v;
    };
    for (int idx = tid; idx < R * KP; idx += T) {
        const int r = idx / KP, j = idx - r * KP;
        Ws[r * LDW + j] = (r0 + r < m && j < k) ? W[(r0 + r) * a.ldw + j] : 0.f;
    }
    auto load_h = [&](float want) {                        // H -> LDS (want = 0: the caller's H; else the granules of that step), then G = H H^T (wave t = jt1 JT + jt2 owns a 16 x 16 tile)
        if (want == 0.f) hs_fill<NW, KP>(H, a.ldh, k, n, NS, LDH, Hs, wv, lane);
        else hs_fill_granules<NW, KP>(Hg, k, n, NS, LDH, Hs, wv, lane, want, bar, a.patience);
        __syncthreads();
        if (wv < JT * JT) {
            const int j1 = wv / JT, j2 = wv - j1 * JT;
            f32x4 g[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            for (int ct = 0; ct < nct; ++ct) {             // lane (i, q) reg r = G[16 j1 + 4 q + r][16 j2 + i]
                const f32x4 h1 = *reinterpret_cast<const f32x4*>(&Hs[(16 * j1 + i) * LDH + 16 * ct + 4 * q]);
                const f32x4 h2 = *reinterpret_cast<const f32x4*>(&Hs[(16 * j2 + i) * LDH + 16 * ct + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) g[ct & 1] = SM_MFMA(h1[r], h2[r], g[ct & 1]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Gs[(16 * j1 + 4 * q + r) * LDG + 16 * j2 + i] = g[0][r] + g[1][r];
        }
        __syncthreads();
    };
    hs_zero_tail(Hs, k, KP, LDH, tid, T);
    load_h(0.f);
    float wreg[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) wreg[s] = Ws[(16 * wv + i) * LDW + 4 * s + q];
    const bool rowok = r0 + 16 * wv + i < m;
    unsigned gen = 0;

    for (int it = 0; it < a.itr; ++it) {
        const bool clamp = (it % 10 == 0);
        if (a.w_update) {
            // ---------------------------------------------------------------- W phase: rows 16 wv .. 16 wv + 15
            f32x4 acc2[2][JT];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc2[h2][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 apre[4];
            if constexpr (!ALDS) {
#pragma unroll
                for (int t = 0; t < 4; ++t) apre[t] = a_row4(16 * (t < nct ? t : nct - 1) + 4 * q);
            }
            for (int ct0 = 0; ct0 < nct; ct0 += 4) {
                f32x4 av[4];
                if constexpr (!ALDS) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) av[t] = apre[t];
                    if (ct0 + 4 < nct) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) apre[t] = a_row4(16 * (ct0 + 4 + t < nct ? ct0 + 4 + t : nct - 1) + 4 * q);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int c0 = 16 * (ct0 + t < nct ? ct0 + t : nct - 1);
                    if constexpr (ALDS) av[t] = a_row4(c0 + 4 * q);
                    if (ct0 + t >= nct) av[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {      // lane (j = i, q) reg r = (A H^T)[row 4 q + r][16 jt + i]
                        const f32x4 hv = *reinterpret_cast<const f32x4*>(&Hs[(16 * jt + i) * LDH + c0 + 4 * q]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc2[t & 1][jt] = SM_MFMA(av[t][r], hv[r], acc2[t & 1][jt]);
                    }
                }
            }
            // W G in the same layout, then the update through the wave's own rows of Ws
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};            // lane (j = i, q) reg r = (W G)[row 4 q + r][16 jt + i]
#pragma unroll
                for (int s = 0; s < KS; ++s) d = SM_MFMA(wreg[s], Gs[(4 * s + q) * LDG + 16 * jt + i], d);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float* wp = &Ws[(16 * wv + 4 * q + r) * LDW + 16 * jt + i];
                    *wp = *wp * ((acc2[0][jt][r] + acc2[1][jt][r]) * __builtin_amdgcn_rcpf(d[r] + eps));
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < KS; ++s) wreg[s] = Ws[(16 * wv + i) * LDW + 4 * s + q];
        }
        __syncthreads();                                   // every wave's rows of Ws are the updated ones
        // -------------------------------------------------------------------- H phase: column tiles wv, wv + NW, ...
        for (int ct = wv; ct < nct; ct += NW) {
            const int c0 = 16 * ct;
            f32x4 acc3[2][JT];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc3[h2][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 hpre[4];
            if constexpr (!ALDS) {
#pragma unroll
                for (int t = 0; t < 4; ++t) hpre[t] = a_col4(16 * t + 4 * q, c0 + i);
            }
            static_for<0, (NW + 3) / 4>([&](auto gc) {     // four row tiles at a time; NW = 6: four, then two
                constexpr int rt0 = 4 * decltype(gc)::value, C = NW - rt0 < 4 ? NW - rt0 : 4;
                f32x4 hcur[4];
                if constexpr (!ALDS) {
#pragma unroll
                    for (int t = 0; t < C; ++t) hcur[t] = hpre[t];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (rt0 + 4 + t < NW) hpre[t] = a_col4(16 * (rt0 + 4 + t) + 4 * q, c0 + i);
                } else {
#pragma unroll
                    for (int t = 0; t < C; ++t) hcur[t] = a_col4(16 * (rt0 + t) + 4 * q, c0 + i);
                }
#pragma unroll
                for (int t = 0; t < C; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float av = hcur[t][r];
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt)    // lane (col i, q) reg r = (W^T A)[16 jt + 4 q + r][c0 + i]
                            acc3[t & 1][jt] = SM_MFMA(Ws[(16 * (rt0 + t) + 4 * q + r) * LDW + 16 * jt + i], av, acc3[t & 1][jt]);
                    }
            });
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * jt + 4 * q + r < k) st_dev(&part[((long)p * KP + 16 * jt + 4 * q + r) * NS + c0 + i], acc3[0][jt][r] + acc3[1][jt][r]);   // (only the k real rows are read)
        }
        if (wv < JT * JT) {                                // the slab's W^T W: lane (i, q) reg r = G[16 j1 + 4 q + r][16 j2 + i]
            const int j1 = wv / JT, j2 = wv - j1 * JT;
            f32x4 g[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            for (int rt = 0; rt < NW; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    g[rt & 1] = SM_MFMA(Ws[(16 * rt + 4 * q + r) * LDW + 16 * j1 + i], Ws[(16 * rt + 4 * q + r) * LDW + 16 * j2 + i], g[rt & 1]);
#pragma unroll
            for (int r = 0; r < 4; ++r) st_dev(&pg[((long)p * KP + 16 * j1 + 4 * q + r) * KP + 16 * j2 + i], g[0][r] + g[1][r]);
        }
        small_barrier(bar, (unsigned)P * ++gen, a.patience);
        // -------------------------------------------------------------------- H update
        for (int e = tid; e < KP * KP; e += T) {           // G = W^T W over all slabs, slab order
            float x = 0.f;
            for (int g0 = 0; g0 < P; g0 += 8) {
                float y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) y[u] = ld_dev(&pg[(long)(g0 + u < P ? g0 + u : P - 1) * KP * KP + e]);
#pragma unroll
                for (int u = 0; u < 8; ++u) x += (g0 + u < P) ? y[u] : 0.f;
            }
            Gs[(e / KP) * LDG + e % KP] = x;
        }
        __syncthreads();
        for (int e = p * T + tid; e < k * NS; e += P * T) {  // (the rows [k, KP) of H do not exist)
            const int j = e / NS, c = e - j * NS;
            float sum = 0.f;
            auto add = [&](auto wide, int g0) __attribute__((always_inline)) {      // `wide` partials in flight
                constexpr int WD = decltype(wide)::value;
                float v[WD];
#pragma unroll
                for (int u = 0; u < WD; ++u) v[u] = ld_dev(&part[((long)(g0 + u < P ? g0 + u : P - 1) * KP + j) * NS + c]);
#pragma unroll
                for (int u = 0; u < WD; ++u) sum += (g0 + u < P) ? v[u] : 0.f;
            };
            if (P <= 8) add(std::integral_constant<int, 8>{}, 0);
            else for (int g0 = 0; g0 < P; g0 += 12) add(std::integral_constant<int, 12>{}, g0);
            float dot = 0.f;
#pragma unroll
            for (int l = 0; l < KP; ++l) dot = fmaf(Gs[j * LDG + l], Hs[l * LDH + c], dot);
            float h = Hs[j * LDH + c] * (sum * __builtin_amdgcn_rcpf(dot + eps));
            if (j < k && c < n) {
                if (clamp) h = fmaxf(h, eps);
                H[(long)j * a.ldh + c] = h;                                                    // (read by the host side only, after the launch)
                buf_st_f32x2(f32x2{h, (float)(it + 1)}, buf_rsrc(Hg), (j * NS + c) * 8, 0, 16);   // published: value and step in one granule
            }
        }
        __syncthreads();                                   // (every thread has read its old elements of Hs and Gs)
        load_h((float)(it + 1));                           // (no device-wide barrier: the granules carry their step)
        if (clamp) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (rowok && 4 * s + q < k) wreg[s] = fmaxf(wreg[s], eps);
                Ws[(16 * wv + i) * LDW + 4 * s + q] = wreg[s];
            }
            __syncthreads();
        }
    }
    if (a.itr > 0) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
            if (rowok && 4 * s + q < k) W[(r0 + 16 * wv + i) * a.ldw + 4 * s + q] = wreg[s];
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// HALS / Frobenius (dist_nmf.py:873-934) on the same slabs.  A is streamed from the L2 (fp32 or bf16-stored: TA), LDS holds H, the
// slab's W, the Gram matrix and the workgroup's share of W^T A.
//   W phase: A H^T as in the Frobenius kernel, T = A H^T - W G' with G' = G masked to l > j (the l = j term of the reference's
//     W G[:, j] cancels its W[:, j] G[j][j]; the l < j terms are the rank-1 corrections below) on the matrix cores, into the slab's rows
//     in LDS.  Then the k columns in sequence (:884-891): u = max(T[:, kk], eps); the slab's sum of squares goes to the problem's slot
//     [parity][kk][slab] -- the value is its own flag -- and every workgroup collects the P slots (the same butterfly sum everywhere);
//     w = u / norm; T[:, j] -= w G[kk][j] for j > kk.  One exchange per column: the price of the global column norm.
//   H phase: W^T A + the slab's W^T W as in the Frobenius kernel; barrier; Gram and the workgroup's columns of W^T A summed in slab
//     order, then a thread per column runs the k rows in sequence (:905-909) on the LDS copy of H, published as {value, step} granules;
//     re-read H from the granules (no second barrier: hs_fill_granules).
//   The slots of a parity are reset by their owner after the first barrier of the step that used them: every reader is done with them
//   by then, and they are not written again before the step after next.
constexpr unsigned SLOT_EMPTY = 0xffffffffu;               // (a NaN pattern: a sum of squares of finite data never has it; NaN data would read as "not there yet" until the wait times out and the fit reports it)

// ALDS: the slab of A in LDS in its storage type (bf16: 66 KiB for 128 rows x 256 columns -- it fits beside the rest; fp32 slabs stream)
template <int KP, int NW, typename TA, bool ALDS>
__global__ __launch_bounds__(64 * NW, 1) void small_hals_fit_kernel(SmallKlArgs a) {
    constexpr int JT = KP / 16, KS = KP / 4, R = 16 * NW, T = 64 * NW, LDW = KP + 1, LDG = KP + 1;
    static_assert(JT * JT <= NW, "one wave per Gram tile");
    const int z = a.z0 + blockIdx.z, p = blockIdx.x, P = gridDim.x;
    const TA* __restrict__ A = reinterpret_cast<const TA*>(a.A) + (long)z * a.a_stride;
    float* __restrict__ W = a.W + (long)z * a.w_stride;
    float* H = a.H + (long)z * a.h_stride;
    float* part = a.part + (long)z * a.part_stride;
    unsigned* bar = a.bar + (long)z * a.bar_stride;
    unsigned* slots = a.slots + (long)z * a.slots_stride;
    float* Hg = a.hg + (long)z * a.hg_stride;             // [KP][NS] granules of H (hs_fill_granules)
    const int m = a.m, n = a.n, k = a.k, cw = a.cw;
    const float eps = a.eps;
    const int NS = (n + 15) & ~15, nct = NS / 16, LDH = NS + 4;
    float* pg = part + (long)P * KP * NS;                  // [P][KP][KP] W^T W of the slabs
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                                      // [KP][LDH]
    float* Ws = Hs + KP * LDH;                             // [R][LDW]   W; during the W sweep: T
    float* Gs = Ws + R * LDW;                              // [KP][LDG]  H H^T (W phase), then W^T W (H sweep)
    float* atw = Gs + KP * LDG;                            // [KP][cw]   this workgroup's columns of W^T A
    float* red = atw + KP * cw;                            // [NW + 8]   wave sums / the broadcast norm
    float* Ts = red + NW + 8;                              // [R][LDW]   T of the W sweep (the new columns go straight into Ws)
    TA* As = reinterpret_cast<TA*>(Ts + R * LDW);          // [R][LDA]   ALDS: the slab of A as stored (zero beyond m / n)
    const int LDA = NS + 8;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long r0 = (long)p * R;
    const bool avec = (a.lda % 4 == 0) && (((unsigned long)A & (4 * sizeof(TA) - 1)) == 0);
    auto a_row4 = [&](int c) -> f32x4 {                    // A[r0 + 16 wv + i][c .. c + 3] (zero outside)
        const long row = r0 + 16 * wv + i;
        f32x4 v;
        if constexpr (ALDS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = sm_ld<TA>(&As[(16 * wv + i) * LDA + c + e]);
            return v;
        }
        if (row < m && c + 4 <= n && avec) {               // one 16-byte (fp32) / 8-byte (bf16) access
            if constexpr (sizeof(TA) == 4) v = *reinterpret_cast<const f32x4*>(A + row * a.lda + c);
            else {
                const uint2 raw = *reinterpret_cast<const uint2*>(A + row * a.lda + c);
                v[0] = __builtin_bit_cast(float, raw.x << 16); v[1] = __builtin_bit_cast(float, raw.x & 0xffff0000u);
                v[2] = __builtin_bit_cast(float, raw.y << 16); v[3] = __builtin_bit_cast(float, raw.y & 0xffff0000u);
            }
            return v;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (row < m && c + e < n) ? sm_ld<TA>(A + row * a.lda + c + e) : 0.f;
        return v;
    };
    auto a_col4 = [&](int rbase, int c) -> f32x4 {         // A[r0 + rbase + r][c], r = 0..3
        f32x4 v;
        if constexpr (ALDS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = sm_ld<TA>(&As[(rbase + e) * LDA + c]);
            return v;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (r0 + rbase + e < m && c < n) ? sm_ld<TA>(A + (r0 + rbase + e) * a.lda + c) : 0.f;
        return v;
    };
    for (int idx = tid; idx < R * KP; idx += T) {
        const int r = idx / KP, j = idx - r * KP;
        Ws[r * LDW + j] = (r0 + r < m && j < k) ? W[(r0 + r) * a.ldw + j] : 0.f;
    }
    if constexpr (ALDS) {
        for (int idx = tid; idx < R * NS; idx += T) {
            const int r = idx / NS, c = idx - r * NS;
            As[r * LDA + c] = (r0 + r < m && c < n) ? A[(r0 + r) * a.lda + c] : TA(0);
        }
    }
    auto load_h = [&](float want) {                        // H -> LDS (want = 0: the caller's H; else the granules of that step), then G = H H^T (wave t = jt1 JT + jt2 owns a 16 x 16 tile)
        if (want == 0.f) hs_fill<NW, KP>(H, a.ldh, k, n, NS, LDH, Hs, wv, lane);
        else hs_fill_granules<NW, KP>(Hg, k, n, NS, LDH, Hs, wv, lane, want, bar, a.patience);
        __syncthreads();
        if (wv < JT * JT) {
            const int j1 = wv / JT, j2 = wv - j1 * JT;
            f32x4 g[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            for (int ct = 0; ct < nct; ++ct) {
                const f32x4 h1 = *reinterpret_cast<const f32x4*>(&Hs[(16 * j1 + i) * LDH + 16 * ct + 4 * q]);
                const f32x4 h2 = *reinterpret_cast<const f32x4*>(&Hs[(16 * j2 + i) * LDH + 16 * ct + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) g[ct & 1] = SM_MFMA(h1[r], h2[r], g[ct & 1]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Gs[(16 * j1 + 4 * q + r) * LDG + 16 * j2 + i] = g[0][r] + g[1][r];
        }
        __syncthreads();
    };
    hs_zero_tail(Hs, k, KP, LDH, tid, T);
    load_h(0.f);
    float wreg[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) wreg[s] = Ws[(16 * wv + i) * LDW + 4 * s + q];
    const bool rowok = r0 + 16 * wv + i < m;
    unsigned gen = 0;
    bool lost = false;                                     // (thread 0 of wave 0: a slot wait gave up -- no more waiting in this fit)

    for (int it = 0; it < a.itr; ++it) {
        const bool clamp = (it % 10 == 0);
        unsigned* slot = slots + (long)(it & 1) * KP * P;
        if (a.w_update) {
            // ---------------------------------------------------------------- A H^T for rows 16 wv .. 16 wv + 15
            f32x4 acc2[2][JT];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc2[h2][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 apre[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) apre[t] = a_row4(16 * (t < nct ? t : nct - 1) + 4 * q);
            for (int ct0 = 0; ct0 < nct; ct0 += 4) {
                f32x4 av[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) av[t] = apre[t];
                if (ct0 + 4 < nct) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) apre[t] = a_row4(16 * (ct0 + 4 + t < nct ? ct0 + 4 + t : nct - 1) + 4 * q);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int c0 = 16 * (ct0 + t < nct ? ct0 + t : nct - 1);
                    if (ct0 + t >= nct) av[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {      // lane (j = i, q) reg r = (A H^T)[row 4 q + r][16 jt + i]
                        const f32x4 hv = *reinterpret_cast<const f32x4*>(&Hs[(16 * jt + i) * LDH + c0 + 4 * q]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc2[t & 1][jt] = SM_MFMA(av[t][r], hv[r], acc2[t & 1][jt]);
                    }
                }
            }
            // T = A H^T - W G' (G'[l][j] = G[l][j] for l > j), into the wave's rows of Ws
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < KS; ++s) d = SM_MFMA(wreg[s], (4 * s + q > 16 * jt + i) ? Gs[(4 * s + q) * LDG + 16 * jt + i] : 0.f, d);
#pragma unroll
                for (int r = 0; r < 4; ++r) Ts[(16 * wv + 4 * q + r) * LDW + 16 * jt + i] = (acc2[0][jt][r] + acc2[1][jt][r]) - d[r];
            }
            __syncthreads();
            // ---------------------------------------------------------------- the k columns in sequence
            const int srow = tid % R, sgrp = tid / R;      // the rank-1 corrections: T / R threads per row, columns sgrp, sgrp + T / R, ...
            constexpr int SG = T / R;                      // (= 4)
            for (int kk = 0; kk < k; ++kk) {                // two workgroup barriers and one exchange per column
                float sq = 0.f;
                if (tid < R) {
                    const float u = fmaxf(Ts[tid * LDW + kk], eps);
                    sq = (r0 + tid < m) ? u * u : 0.f;
                }
                if (wv * 64 < R) {                         // the waves that hold rows: sum of the wave, lane 0 -> red
                    float v = row16_sum(sq);
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (lane == 0) red[wv] = v;
                }
                __syncthreads();
                if (wv == 0) {
                    float mine = 0.f;
#pragma unroll
                    for (int w = 0; w < (R + 63) / 64; ++w) mine += red[w];
                    if (lane == 0) __hip_atomic_store(&slot[kk * P + p], __builtin_bit_cast(unsigned, mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // collect the P slots: lane g polls slot g (P <= 64)
                    unsigned bits = lane < P ? SLOT_EMPTY : 0u;
                    const unsigned long long t0 = wall_clock64();
                    unsigned spins = 0;
                    while (!lost) {
                        if (bits == SLOT_EMPTY) bits = __hip_atomic_load(&slot[kk * P + (lane < P ? lane : 0)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (!__any(bits == SLOT_EMPTY)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if ((++spins & 15u) == 15u && __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) lost = true;
                        if (wall_clock64() - t0 > a.patience) {
                            if (lane == 0) {
                                __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(&g_small_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            lost = true;
                        }
                    }
                    float v = (lane < P && bits != SLOT_EMPTY) ? __builtin_bit_cast(float, bits) : 0.f;
                    v = row16_sum(v);
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (lane == 0) red[NW + (kk & 1)] = v;  // (two cells: the next column's norm never lands on one still being read)
                }
                __syncthreads();
                const float ss = sqrtf(red[NW + (kk & 1)]);  // utils.py:388-391: the 2-norm of the whole column
                const float u = fmaxf(Ts[srow * LDW + kk], eps);
                const float w = ss > 0.f ? u / ss : u;
                for (int j = kk + 1 + sgrp; j < k; j += SG) Ts[srow * LDW + j] = fmaf(-w, Gs[kk * LDG + j], Ts[srow * LDW + j]);
                if (sgrp == 0) Ws[srow * LDW + kk] = (r0 + srow < m) ? w : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < KS; ++s) wreg[s] = Ws[(16 * wv + i) * LDW + 4 * s + q];
        }
        __syncthreads();
        // -------------------------------------------------------------------- H phase: W^T A per column tile, the slab's W^T W
        for (int ct = wv; ct < nct; ct += NW) {
            const int c0 = 16 * ct;
            f32x4 acc3[2][JT];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc3[h2][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 hpre[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) hpre[t] = a_col4(16 * t + 4 * q, c0 + i);
            for (int rt0 = 0; rt0 < NW; rt0 += 4) {
                f32x4 hcur[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) hcur[t] = hpre[t];
                if (rt0 + 4 < NW) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) hpre[t] = a_col4(16 * (rt0 + 4 + t) + 4 * q, c0 + i);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int jt = 0; jt < JT; ++jt)    // lane (col i, q) reg r = (W^T A)[16 jt + 4 q + r][c0 + i]
                            acc3[t & 1][jt] = SM_MFMA(Ws[(16 * (rt0 + t) + 4 * q + r) * LDW + 16 * jt + i], hcur[t][r], acc3[t & 1][jt]);
            }
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) st_dev(&part[((long)p * KP + 16 * jt + 4 * q + r) * NS + c0 + i], acc3[0][jt][r] + acc3[1][jt][r]);
        }
        if (wv < JT * JT) {
            const int j1 = wv / JT, j2 = wv - j1 * JT;
            f32x4 g[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            for (int rt = 0; rt < NW; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    g[rt & 1] = SM_MFMA(Ws[(16 * rt + 4 * q + r) * LDW + 16 * j1 + i], Ws[(16 * rt + 4 * q + r) * LDW + 16 * j2 + i], g[rt & 1]);
#pragma unroll
            for (int r = 0; r < 4; ++r) st_dev(&pg[((long)p * KP + 16 * j1 + 4 * q + r) * KP + 16 * j2 + i], g[0][r] + g[1][r]);
        }
        small_barrier(bar, (unsigned)P * ++gen, a.patience);
        if (a.w_update && tid < k) __hip_atomic_store(&slot[tid * P + p], SLOT_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // -------------------------------------------------------------------- H sweep on this workgroup's columns [cb, ce)
        for (int e = tid; e < KP * KP; e += T) {
            float x = 0.f;
            for (int g0 = 0; g0 < P; g0 += 8) {
                float y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) y[u] = ld_dev(&pg[(long)(g0 + u < P ? g0 + u : P - 1) * KP * KP + e]);
#pragma unroll
                for (int u = 0; u < 8; ++u) x += (g0 + u < P) ? y[u] : 0.f;
            }
            Gs[(e / KP) * LDG + e % KP] = x;
        }
        const int cb = p * cw, ce = (cb + cw < NS) ? cb + cw : NS;
        for (int e = tid; e < KP * cw; e += T) {
            const int j = e / cw, c = cb + e - j * cw;
            float sum = 0.f;
            if (c < ce) {
                for (int g0 = 0; g0 < P; g0 += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = ld_dev(&part[((long)(g0 + u < P ? g0 + u : P - 1) * KP + j) * NS + c]);
#pragma unroll
                    for (int u = 0; u < 8; ++u) sum += (g0 + u < P) ? v[u] : 0.f;
                }
            }
            atw[e] = sum;
        }
        __syncthreads();
        // KP lanes per column (lane l holds H[l][c]): the dot product of row kk is a product per lane + a butterfly sum over the 16-lane
        // DPP row(s); lane kk takes the new value -- rows in sequence, updated rows used at once (:905-909)
        {
            constexpr int CPW = 64 / KP;                    // columns per wave: 4 (KP = 16) or 2 (KP = 32)
            const int l = lane % KP, cl = lane / KP;
            for (int c = cb + wv * CPW + cl; c < cb + cw; c += NW * CPW) {       // (uniform trip count per wave: cw is the same everywhere)
                const bool cok = c < ce;
                float h = (cok && l < k) ? Hs[l * LDH + c] : 0.f;
                const float aw = (cok && l < k) ? atw[l * cw + (c - cb)] : 0.f;
                for (int kk = 0; kk < k; ++kk) {
                    float d = Gs[kk * LDG + l] * h;         // (G rows beyond k are zero)
                    d = row16_sum(d);
                    if constexpr (KP == 32) d += __shfl_xor(d, 16, 64);
                    const float t = fmaxf(h + aw - d, eps);
                    if (l == kk) h = t;
                }
                if (cok && l < k) {
                    const float v = (c < n) ? h : 0.f;
                    Hs[l * LDH + c] = v;
                    if (c < n) {
                        H[(long)l * a.ldh + c] = v;                                               // (read by the host side only, after the launch)
                        buf_st_f32x2(f32x2{v, (float)(it + 1)}, buf_rsrc(Hg), (l * NS + c) * 8, 0, 16);   // published: value and step in one granule
                    }
                }
            }
        }
        __syncthreads();                                   // (the sweep of every wave is over: Hs may be overwritten)
        load_h((float)(it + 1));                           // (no device-wide barrier: the granules carry their step, small_kl_fit_kernel)
        if (clamp) {                                       // W = max(W, eps) (pyDNMF.py:170-172; H is >= eps after its sweep)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (rowok && 4 * s + q < k) wreg[s] = fmaxf(wreg[s], eps);
                Ws[(16 * wv + i) * LDW + 4 * s + q] = wreg[s];
            }
            __syncthreads();
        }
    }
    if (a.itr > 0) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
            if (rowok && 4 * s + q < k) W[(r0 + 16 * wv + i) * a.ldw + 4 * s + q] = wreg[s];
    }
}
inline size_t small_hals_lds(int kp, int nw, long n, int cw, size_t a_elem /* bytes per element of an LDS-resident A slab, 0: streamed */) {
    const long ns = (n + 15) & ~15L;
    return ((size_t)kp * (ns + 4) + 2 * (size_t)(16 * nw) * (kp + 1) + (size_t)kp * (kp + 1) + (size_t)kp * cw + nw + 8) * sizeof(float) +
           (size_t)(16 * nw) * (ns + 8) * a_elem;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// MU/KL with W FIXED (w_update = 0: the regression fit of every k of an NMFk sweep, pyDNMFk.py:243-247).  The H update of a column
// reads nothing but that column of A and of H (and all of W): the columns are independent problems.  A workgroup owns 16 columns for the
// whole fit -- all of W in LDS, its columns of H in registers / LDS, its 16 columns of A streamed from the L2 (m x 64 bytes per step) -- and
// iterates on its own: NO grid barrier, no exchange, no residency requirement.  Wave w takes the row tiles w, w + NW, ...: S = W H
// (B operand = the wave's copy of the H columns), U = A / (S + eps) in the C registers = the B operand of W^T U; the waves' partial
// k x 16 results are summed in wave order through LDS, H *= W^T U / (colsum(W) + eps), clamp every tenth step.  The LDS copy of W is
// clamped once, after step 0 (pyDNMF.py:155: idempotent afterwards), and its column sums are taken again.
template <int KP, int NW>
__global__ __launch_bounds__(64 * NW, 1) void small_kl_hfit_kernel(SmallKlArgs a) {
    constexpr int JT = KP / 16, KS = KP / 4, T = 64 * NW, LDW = KP + 1, LDR = 17;
    const int z = a.z0 + blockIdx.z, c0 = 16 * blockIdx.x;
    const float* __restrict__ A = a.A + (long)z * a.a_stride;
    const float* __restrict__ W = a.W + (long)z * a.w_stride;
    float* H = a.H + (long)z * a.h_stride;
    const int m = a.m, n = a.n, k = a.k;
    const float eps = a.eps;
    const int ksteps = (k + 3) >> 2;
    const int m16 = (m + 15) & ~15, nrt = m16 / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                                      // [m16][LDW]     all of W (zero beyond m / k)
    float* red = Wl + m16 * LDW;                           // [NW][KP][LDR]  the waves' partial W^T U (also scratch of the column sums)
    float* hs = red + NW * KP * LDR;                       // [KP][LDR]      the 16 columns of H
    float* xw = hs + KP * LDR;                             // [KP]           column sums of W
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int idx = tid; idx < m16 * KP; idx += T) {
        const int r = idx / KP, j = idx - r * KP;
        Wl[r * LDW + j] = (r < m && j < k) ? W[(long)r * a.ldw + j] : 0.f;
    }
    for (int idx = tid; idx < KP * 16; idx += T) {
        const int j = idx >> 4, c = idx & 15;
        hs[j * LDR + c] = (j < k && c0 + c < n) ? H[(long)j * a.ldh + c0 + c] : 0.f;
    }
    __syncthreads();
    auto colsums = [&]() {                                 // xw[j] = sum_r Wl[r][j]: T / KP strided parts, added in part order
        constexpr int PARTS = T / KP;
        static_assert(PARTS <= NW * LDR, "scratch");
        const int j = tid % KP, part = tid / KP;
        float v = 0.f;
        for (int r = part; r < m16; r += PARTS) v += Wl[r * LDW + j];
        red[part * KP + j] = v;
        __syncthreads();
        if (tid < KP) {
            float x = 0.f;
            for (int g = 0; g < PARTS; ++g) x += red[g * KP + tid];
            xw[tid] = x;
        }
        __syncthreads();
    };
    colsums();
    float hb[KS];                                          // lane (col i, q): H[4 s + q][c0 + i]
#pragma unroll
    for (int s = 0; s < KS; ++s) hb[s] = hs[(4 * s + q) * LDR + i];
    auto a_col4 = [&](int rt) -> f32x4 {                   // A[16 rt + 4 q + r][c0 + i], r = 0..3 (zero outside)
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 16 * rt + 4 * q + e;
            v[e] = (rt < nrt && row < m && c0 + i < n) ? A[(long)row * a.lda + c0 + i] : 0.f;
        }
        return v;
    };
    for (int it = 0; it < a.itr; ++it) {
        const bool clamp = (it % 10 == 0);
        f32x4 acc[2][JT];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) acc[h2][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 apre[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) apre[t] = a_col4(wv + NW * t);
        for (int g0 = 0; wv + NW * g0 < nrt; g0 += 4) {    // four of the wave's row tiles at a time
            f32x4 acur[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acur[t] = apre[t];
#pragma unroll
            for (int t = 0; t < 4; ++t) apre[t] = a_col4(wv + NW * (g0 + 4 + t));
            int rt[4];
            f32x4 d[4];                                    // lane (col i, q) reg r = (W H)[16 rt + 4 q + r][c0 + i]
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int x = wv + NW * (g0 + t);
                rt[t] = x < nrt ? x : nrt - 1;             // (a tile beyond the last repeats it; its U is zero: A reads as zero there)
                d[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            float wop[KS][4];                              // (operands first: small_kl_fit_kernel)
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t) wop[s][t] = Wl[(16 * rt[t] + i) * LDW + 4 * s + q];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s >= ksteps) break;                    // (the zero-padded steps beyond k: uniform)
#pragma unroll
                for (int t = 0; t < 4; ++t) d[t] = SM_MFMA(wop[s][t], hb[s], d[t]);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 u = quot4(acur[t], d[t], eps);
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)            // lane (col i, q) reg r = (W^T U)[16 jt + 4 q + r][c0 + i]
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[t & 1][jt] = SM_MFMA(Wl[(16 * rt[t] + 4 * q + r) * LDW + 16 * jt + i], u[r], acc[t & 1][jt]);
            }
        }
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wv * KP + 16 * jt + 4 * q + r) * LDR + i] = acc[0][jt][r] + acc[1][jt][r];
        __syncthreads();
        for (int e = tid; e < KP * 16; e += T) {
            const int j = e >> 4, c = e & 15;
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += red[(w * KP + j) * LDR + c];
            float h = hs[j * LDR + c] * (sum * __builtin_amdgcn_rcpf(xw[j] + eps));
            if (clamp && j < k && c0 + c < n) h = fmaxf(h, eps);
            hs[j * LDR + c] = h;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KS; ++s) hb[s] = hs[(4 * s + q) * LDR + i];
        if (it == 0) {                                     // W = max(W, eps), once (idempotent at the later clamps)
            for (int idx = tid; idx < m16 * KP; idx += T) {
                const int r = idx / KP, j = idx - r * KP;
                if (r < m && j < k) Wl[r * LDW + j] = fmaxf(Wl[r * LDW + j], eps);
            }
            __syncthreads();
            colsums();
        }
    }
    for (int e = tid; e < KP * 16; e += T) {
        const int j = e >> 4, c = e & 15;
        if (j < k && c0 + c < n) H[(long)j * a.ldh + c0 + c] = hs[j * LDR + c];
    }
    // (W itself is only read here: a workgroup that starts late must still find the caller's W, not a clamped one -- the host clamps W
    // with its own launch after this kernel, csrc/dnmf_fit.hip)
}
inline size_t small_kl_hfit_lds(int kp, int nw, long m) {
    const long m16 = (m + 15) & ~15L;
    return ((size_t)m16 * (kp + 1) + (size_t)nw * kp * 17 + (size_t)kp * 17 + kp) * sizeof(float);
}

// LDS of either kernel: [slab of A] + H + the slab's W + (KL: row sums, per-wave column sums; FRO: the k x k Gram matrix)
inline size_t small_fro_lds(int kp, int nw, long n, bool alds, size_t a_elem) {
    const long ns = (n + 15) & ~15L;
    const size_t f = (size_t)kp * (ns + 4) + (size_t)(16 * nw) * (kp + 1) + (((size_t)kp * (kp + 1) + 3) & ~size_t(3));
    return f * sizeof(float) + (alds ? (size_t)(16 * nw) * (ns + (a_elem == 4 ? 4 : 8)) * a_elem : 0);
}
inline size_t small_kl_lds(int kp, int nw, long n, bool alds) {
    const long ns = (n + 15) & ~15L;
    const size_t tail = std::max<size_t>((size_t)kp + (size_t)nw * kp, (size_t)kp * (kp + 1));
    return ((alds ? (size_t)(16 * nw) * (ns + 4) : 0) + (size_t)kp * (ns + 4) + (size_t)(16 * nw) * (kp + 1) + tail) * sizeof(float);
}

}  // namespace
