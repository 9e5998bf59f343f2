// dnmf_team2.h -- the one-pass MU/Frobenius team kernel with the waves of a workgroup split by ROLE (round 6, second design).
// Same algorithm, same team exchange and the same arithmetic order as team_fro_kernel (csrc/dnmf_team.h, which documents them); what
// changes is who does what inside a workgroup, because of what that kernel's counters showed (profiles/r06a_team_*, r06b_team_*):
// with all eight waves running the same program between workgroup barriers, every wave is in its vector / LDS / wait section at the
// same time, and on gfx950 an fp32 vector instruction does not overlap an fp32 MFMA of its SIMD (DESIGN "Round 3") -- 23 % of the
// stage had no MFMA issuing anywhere.
//
//   waves 0-3, the P role (wave p owns columns 128 p .. 128 p + 127 of the member's 512): stream the slab pieces of A from HBM into
//     registers and into the LDS ring, form the partial of A H^T (64 MFMAs per slab; H in 64 registers), publish the member's partial.
//   waves 4-7, the Q role (wave p + 4 owns the same columns, and sits on the same SIMD as wave p where the hardware deals waves to
//     SIMDs cyclically -- speed only): gather the team's partials, update the slab's W rows, add W_new^T A from the LDS copy (64 MFMAs
//     per slab; the 32 x 128 accumulator in 64 registers).
// The roles meet only through LDS words: `staged[p]` (P wave p has written slab t's piece), `consumed[p]` (Q wave p + 4 has read it: the
// ring slot may be overwritten), and one arrival counter per role for the two role-wide steps (the four partials of a slab are complete;
// the new W rows are complete).  No workgroup barrier in the loop: a role waits on the OTHER role only when the ring is full or empty,
// so while one wave of a SIMD sits in a wait or in its vector section the other one keeps the matrix pipe busy.
#pragma once
#include "dnmf_team.h"

namespace {

constexpr int TM_LDW = tm_ldw(TM_KP);
constexpr size_t TM2_LDS_BYTES = (size_t)(TM_NBUF * TM_R * TM_LDA + 2 * 4 * TM_R * TM_KP + 2 * 2 * TM_R * TM_KP + 2 * TM_R * TM_LDW + 16) * sizeof(float);

// The LDS words are relaxed atomics between compiler barriers: the ORDER they rely on is the hardware's (the DS operations of a wave
// execute in issue order; all waves of a workgroup share one LDS), and a release / acquire at workgroup scope would make hipcc drain the
// wave's vector-memory queue -- the slab pieces in flight -- at every flag.
#define TM_CBAR() asm volatile("" ::: "memory")
__device__ __forceinline__ unsigned tm_lds_ld(const unsigned* p) {
    const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    TM_CBAR();
    return v;
}
__device__ __forceinline__ void tm_lds_st(unsigned* p, unsigned v, int lane) {
    TM_CBAR();
    if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    TM_CBAR();
}
// wait until the LDS word reaches `target` (the other waves of this workgroup are resident: no time-out needed here)
__device__ __forceinline__ void tm_wait_ge(const unsigned* p, unsigned target) {
    while (tm_lds_ld(p) < target) __builtin_amdgcn_s_sleep(1);
}
// arrival of one wave of a role at its `gen`-th meeting (4 waves): the wave's LDS writes are ahead of the add (DS operations of a wave
// execute in issue order), the others' are visible once the counter has reached 4 gen
__device__ __forceinline__ void tm_role_sync(unsigned* cnt, unsigned gen, int lane) {
    TM_CBAR();
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    TM_CBAR();
    tm_wait_ge(cnt, 4u * gen);
}

template <int SD, int NT>
__global__ __launch_bounds__(64 * TM_NW, 2) void team_split_kernel(TeamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tm_smem[];
    constexpr int EL = TM_R * TM_KP;
    float* Ab = tm_smem;                                   // [NBUF][R][LDA]
    float* red = Ab + TM_NBUF * TM_R * TM_LDA;             // [2][4][EL]    the P waves' partials of A H^T (by slab parity)
    float* dred = red + 2 * 4 * EL;                        // [2][2][EL]    the owner's partial denominators
    float* Wn = dred + 2 * 2 * EL;                         // [2][R][LDW]   the slab's new W rows
    unsigned* sw = (unsigned*)(Wn + 2 * TM_R * TM_LDW);    // sw[0] P arrivals, sw[1] Q arrivals, sw[4 + p] staged, sw[8 + p] consumed
    __shared__ unsigned s_flag;

    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, q4 = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = wv & 3;                                  // column block of the wave pair
    const bool prole = wv < 4;
    const int T = a.T;
    const int b = blockIdx.x, xcd = b & 7, qq = b >> 3;
    const int team = xcd * a.tpx + qq / T, member = qq % T;
    const long m = a.m;
    const int n = a.n, k = a.k;
    const long row0 = (long)team * a.rpt;
    const long rows = min(a.rpt, m - row0);
    const int nsl = rows > 0 ? (int)((rows + TM_R - 1) / TM_R) : 0;
    const int nfull = rows > 0 ? (int)(rows / TM_R) : 0;
    const int cw = 128 * p;                                // the pair's first column inside the member's piece
    const int cb = member * TM_C + cw;                     // ... inside the matrix
    const int rt = tid & 255;                              // thread of the role: elements rt and rt + 256 of a slab's 16 x 32 tile

    if (tid == 0) __hip_atomic_fetch_add(a.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // census, first half
    if (tid < 16) sw[tid] = 0u;

    const i32x4 arsrc = buf_rsrc((const float*)a.A + row0 * a.lda);
    const i32x4 wrsrc = buf_rsrc(a.W + row0 * a.ldw);
    const i32x4 rrsrc = buf_rsrc(a.ring + (long)team * TM_D * (T + 2) * EL);
    constexpr int PLANE = EL * 8;
    const int slotb = (T + 2) * PLANE;
    const int wslab = (int)(a.ldw * 4 * TM_R);
    int wevoff[2];                                         // the role thread's two elements of a slab of W: rows rt >> 5 and 8 + (rt >> 5)
#pragma unroll
    for (int h = 0; h < 2; ++h) wevoff[h] = (rt & 31) < k ? (int)((((rt >> 5) + 8 * h) * a.ldw + (rt & 31)) * 4) : BUF_OOB;
    const int gvoff = rt * 8;

    // census, second half
    if (tid == 0) {
        const unsigned long long t0 = wall_clock64();
        unsigned ok = 1, spins = 0;
        while (__hip_atomic_load(a.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 15u) == 15u && __hip_atomic_load(a.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
            if (wall_clock64() - t0 > a.patience) {
                __hip_atomic_store(a.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&g_team_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        s_flag = ok;
    }
    __syncthreads();
    if (s_flag == 0u || nsl == 0) return;                  // nothing has been written

    if (prole) {
        // =================================================================================== P role
        f32x4 hreg[2][8];                                  // the pair's 32 x 128 block of H as B operand (k-slot q4, kk = lane & 15)
#pragma unroll
        for (int tk = 0; tk < 2; ++tk)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int kk = 16 * tk + i;
                hreg[tk][g] = kk < k ? *reinterpret_cast<const f32x4*>(a.H + (long)kk * a.ldh + cb + 16 * g + 4 * q4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        const int dtk = p & 1, dks = 4 * (p >> 1);         // this wave's share of the owner's W G: tile dtk, contraction steps dks .. dks + 3
        float gden[4];
        int wdvoff[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            gden[u] = a.G[(4 * (dks + u) + q4) * TM_KP + 16 * dtk + i];
            wdvoff[u] = 4 * (dks + u) + q4 < k ? (int)((i * a.ldw + 4 * (dks + u) + q4) * 4) : BUF_OOB;
        }
        const int avoff = (int)(((long)q4 * a.lda + cb + 4 * i) * 4);
        const int arow4 = (int)(a.lda * 16);
        f32x4 stg[SD][8];
        float wold[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, wdn[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        auto load_slab = [&](int t, f32x4 (&d)[8], auto gc) {
            constexpr bool GD = decltype(gc)::value;
            const int soff = t * 4 * arow4;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int vo = (!GD || (long)t * TM_R + 4 * ks + q4 < rows) ? avoff : BUF_OOB;
                d[2 * ks] = buf_ld_f32x4(arsrc, vo, soff + ks * arow4, NT);
                d[2 * ks + 1] = buf_ld_f32x4(arsrc, vo, soff + ks * arow4 + 256, NT);
            }
        };
        auto load_w = [&](int t, int own, float (&wo)[2], float (&wd)[4], auto gc) {     // the old W of slab t for its owner (BUF_OOB otherwise)
            constexpr bool GD = decltype(gc)::value;
            const int soff = t * wslab;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                wo[h] = buf_ld_f32(wrsrc, (own && (!GD || (long)t * TM_R + (rt >> 5) + 8 * h < rows)) ? wevoff[h] : BUF_OOB, soff, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) wd[u] = buf_ld_f32(wrsrc, (own && (!GD || (long)t * TM_R + i < rows)) ? wdvoff[u] : BUF_OOB, soff, 0);
        };
#pragma unroll
        for (int u = 0; u < SD; ++u)
            if (u < nsl) load_slab(u, stg[u], std::true_type{});
        int o0 = 0, o1 = 1 % T;                            // owners of slabs t and t + 1
        load_w(0, o0 == member, wold[0], wdn[0], std::true_type{});
        unsigned gen = 0;

        auto stage = [&](int t, auto uc, auto gc) {
            constexpr int U = decltype(uc)::value, PAR = U & 1;
            constexpr bool GD = decltype(gc)::value;
            if (GD && t >= nsl) return;
            float* const redp = red + PAR * (4 * EL);
            float* const dredp = dred + PAR * (2 * EL);
            float* const buf = Ab + (U % 3) * (TM_R * TM_LDA);
            if (t >= TM_NBUF) tm_wait_ge(sw + 8 + p, (unsigned)(t - TM_NBUF + 1));        // the ring slot is free again
            {   // S: the pair's 16 x 128 piece of slab t into the ring
                float* dst = buf + q4 * TM_LDA + cw + 4 * i;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    *reinterpret_cast<f32x4*>(dst + 4 * ks * TM_LDA) = stg[U % SD][2 * ks];
                    *reinterpret_cast<f32x4*>(dst + 4 * ks * TM_LDA + 64) = stg[U % SD][2 * ks + 1];
                }
                tm_lds_st(sw + 4 + p, (unsigned)(t + 1), lane);
            }
            if (!GD || t + SD < nsl) load_slab(t + SD, stg[U % SD], gc);
            if (!GD || t + 1 < nsl) load_w(t + 1, o1 == member, wold[(U + 1) & 1], wdn[(U + 1) & 1], gc);
            {   // P: 16 x 32 partial of A H^T over the pair's 128 columns
                const float* src = buf + i * TM_LDA + cw + 4 * q4;
                f32x4 pacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(src + 16 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pacc[0] = TM_MFMA(av[e], hreg[0][g][e], pacc[0]);
                        pacc[1] = TM_MFMA(av[e], hreg[1][g][e], pacc[1]);
                    }
                }
                float* dst = redp + p * EL + 4 * q4 * TM_KP + i;
#pragma unroll
                for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[r * TM_KP + 16 * tk] = pacc[tk][r];
            }
            if (o0 == member) {                            // the owner's W G for slab t
                f32x4 dacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 4; ++u) dacc = TM_MFMA(wdn[U & 1][u], gden[u], dacc);
                float* dd = dredp + (p >> 1) * EL + 4 * q4 * TM_KP + 16 * dtk + i;
#pragma unroll
                for (int r = 0; r < 4; ++r) dd[r * TM_KP] = dacc[r];
            }
            tm_role_sync(sw, ++gen, lane);
            {   // publish slab t: the role thread's two elements (the owner adds w_old and den)
                float rr[2][4], dd[2][2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) rr[h][w] = redp[w * EL + rt + 256 * h];
#pragma unroll
                    for (int w = 0; w < 2; ++w) dd[h][w] = dredp[w * EL + rt + 256 * h];
                }
                const float tag = __uint_as_float((unsigned)(t + 1));
                const int slot = (t & (TM_D - 1)) * slotb;
                const int ov = (o0 == member && !(a.xflags & 2)) ? gvoff : BUF_OOB;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float v = ((rr[h][0] + rr[h][1]) + rr[h][2]) + rr[h][3];
                    const float den = (dd[h][0] + dd[h][1]) + a.eps;
                    buf_st_f32x2(f32x2{v, tag}, rrsrc, (a.xflags & 2) ? BUF_OOB : gvoff, slot + member * PLANE + 2048 * h, 16);
                    buf_st_f32x2(f32x2{wold[U & 1][h], tag}, rrsrc, ov, slot + T * PLANE + 2048 * h, 16);
                    buf_st_f32x2(f32x2{den, tag}, rrsrc, ov, slot + (T + 1) * PLANE + 2048 * h, 16);
                }
            }
            o0 = o1;
            o1 = o1 + 1 == T ? 0 : o1 + 1;
        };
        auto group = [&](int t0, auto gc) {
            stage(t0, std::integral_constant<int, 0>{}, gc);
            stage(t0 + 1, std::integral_constant<int, 1>{}, gc);
            stage(t0 + 2, std::integral_constant<int, 2>{}, gc);
            stage(t0 + 3, std::integral_constant<int, 3>{}, gc);
            stage(t0 + 4, std::integral_constant<int, 4>{}, gc);
            stage(t0 + 5, std::integral_constant<int, 5>{}, gc);
        };
        int t0 = 0;
        for (; t0 + 5 + SD < nfull; t0 += 6) group(t0, std::false_type{});
        for (; t0 < nsl; t0 += 6) group(t0, std::true_type{});
    } else {
        // =================================================================================== Q role
        f32x4 qacc[2][8];
#pragma unroll
        for (int tk = 0; tk < 2; ++tk)
#pragma unroll
            for (int j = 0; j < 8; ++j) qacc[tk][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        float tmask[TM_MAXT];
#pragma unroll
        for (int j = 0; j < TM_MAXT; ++j) tmask[j] = j < T ? 1.f : 0.f;
        f32x2 gat[2][TM_MAXT + 2];
        auto issue_gather = [&](int t) {
            const int slot = (t & (TM_D - 1)) * slotb;
#pragma unroll
            for (int j = 0; j < TM_MAXT + 2; ++j) {
                const int pl = j < TM_MAXT ? (j < T ? j : T - 1) : T + (j - TM_MAXT);
#pragma unroll
                for (int h = 0; h < 2; ++h) gat[h][j] = buf_ld_f32x2(rrsrc, gvoff, slot + pl * PLANE + 2048 * h, 16);
            }
        };
        auto tags_ok = [&](unsigned want) {
            unsigned tsum = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < TM_MAXT + 2; ++j) tsum += __float_as_uint(gat[h][j][1]);
            return tsum == 2u * (TM_MAXT + 2) * want;      // a slot's earlier contents carry smaller tags, never larger ones
        };
        auto regather = [&](int t) {                       // a late member: read the slab's planes again, bounded by the wall clock
            const unsigned long long t0 = wall_clock64();
            unsigned spins = 0;
            for (;;) {
                issue_gather(t);
                if (tags_ok((unsigned)(t + 1))) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 15u) == 15u && __hip_atomic_load(a.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (wall_clock64() - t0 > a.patience) {
                    __hip_atomic_store(a.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&g_team_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        };
        if (!(a.xflags & 2)) issue_gather(0);
        else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < TM_MAXT + 2; ++j) gat[h][j] = f32x2{1.f, 0.f};
        }
        int o0 = 0;
        unsigned gen = 0;

        auto stage = [&](int t, auto uc, auto gc) {
            constexpr int U = decltype(uc)::value, PAR = U & 1;
            constexpr bool GD = decltype(gc)::value;
            if (GD && t >= nsl) return;
            float* const Wnp = Wn + PAR * (TM_R * TM_LDW);
            const float* const buf = Ab + (U % 3) * (TM_R * TM_LDA);
            // W update of slab t from the granules asked for a stage ago
            if (!tags_ok((unsigned)(t + 1)) && !(a.xflags & 1)) regather(t);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float ah = gat[h][0][0];
#pragma unroll
                for (int j = 1; j < TM_MAXT; ++j) ah = fmaf(tmask[j], gat[h][j][0], ah);
                const float wn = div_pos(gat[h][TM_MAXT][0] * ah, gat[h][TM_MAXT + 1][0]);
                Wnp[((rt >> 5) + 8 * h) * TM_LDW + (rt & 31)] = wn;
                const bool st = o0 == member && (!GD || (long)t * TM_R + (rt >> 5) + 8 * h < rows);
                buf_st_f32(wn, wrsrc, st ? wevoff[h] : BUF_OOB, t * wslab, 0);
            }
            if ((!GD || t + 1 < nsl) && !(a.xflags & 2)) issue_gather(t + 1);
            tm_role_sync(sw + 1, ++gen, lane);
            tm_wait_ge(sw + 4 + p, (unsigned)(t + 1));     // the P wave of the pair has written the piece
            {   // Q: the team's W^T A gains slab t (this pair's 128 columns, from the LDS copy)
                float wop[2][4];
#pragma unroll
                for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) wop[tk][ks] = Wnp[(4 * ks + q4) * TM_LDW + 16 * tk + i];
                const float* src = buf + q4 * TM_LDA + cw + 4 * i;
                f32x4 bv[4][2];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    bv[ks][0] = *reinterpret_cast<const f32x4*>(src + 4 * ks * TM_LDA);
                    bv[ks][1] = *reinterpret_cast<const f32x4*>(src + 4 * ks * TM_LDA + 64);
                }
                tm_lds_st(sw + 8 + p, (unsigned)(t + 1), lane);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            qacc[0][4 * c2 + j] = TM_MFMA(wop[0][ks], bv[ks][c2][j], qacc[0][4 * c2 + j]);
                            qacc[1][4 * c2 + j] = TM_MFMA(wop[1][ks], bv[ks][c2][j], qacc[1][4 * c2 + j]);
                        }
            }
            o0 = o0 + 1 == T ? 0 : o0 + 1;
        };
        auto group = [&](int t0, auto gc) {
            stage(t0, std::integral_constant<int, 0>{}, gc);
            stage(t0 + 1, std::integral_constant<int, 1>{}, gc);
            stage(t0 + 2, std::integral_constant<int, 2>{}, gc);
            stage(t0 + 3, std::integral_constant<int, 3>{}, gc);
            stage(t0 + 4, std::integral_constant<int, 4>{}, gc);
            stage(t0 + 5, std::integral_constant<int, 5>{}, gc);
        };
        int t0 = 0;
        for (; t0 + 5 + 1 < nfull; t0 += 6) group(t0, std::false_type{});
        for (; t0 < nsl; t0 += 6) group(t0, std::true_type{});

        // the team's partial: accumulator register r of tile (tk, 4 c2 + j) is row 16 tk + 4 q4 + r, column cb + 64 c2 + 4 i + j
        float* Pt = a.P + (long)team * TM_KP * n + cb + 4 * i;
#pragma unroll
        for (int tk = 0; tk < 2; ++tk)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
                    *reinterpret_cast<f32x4*>(Pt + (long)(16 * tk + 4 * q4 + r) * n + 64 * c2) =
                        f32x4{qacc[tk][4 * c2][r], qacc[tk][4 * c2 + 1][r], qacc[tk][4 * c2 + 2][r], qacc[tk][4 * c2 + 3][r]};
    }
}

}  // namespace
