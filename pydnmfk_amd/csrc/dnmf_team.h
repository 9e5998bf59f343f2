// dnmf_team.h -- ONE pass over A per MU/Frobenius iteration for 16 < k <= 32 (round 6): team_fro_kernel.
// Part of libdnmf_hip.so (csrc/dnmf_team.hip plans and launches it; csrc/dnmf.hip: mu_fro_step_impl takes it when the shape allows).
//
// What it computes (dist_nmf.py:716-732 feeding :736-749): for every row r of this rank's block
//     W[r] <- W[r] * (A[r] H^T) / (W[r] (H H^T) + eps)          and, with the NEW rows,          P += W[r]^T A[r]
// -- the W update of a row needs that row's WHOLE A[r] H^T before W^T A may touch the row again, so the row block has to stay on
// chip between its two uses.  A 16-row slab of a 4096-column matrix is 256 KiB: more than one CU's LDS, so a TEAM of T = n / 512
// workgroups (one per CU, all of one XCD where the dispatcher deals blocks round-robin -- speed only) shares it by COLUMNS:
//
//   member j keeps its 16 x 512 piece of the slab in LDS (32 KiB, four slabs deep), wave w of its eight waves owning 64 columns for the
//   whole launch: the matching 32 x 64 block of H lives in 32 registers as the B operand of v_mfma_f32_16x16x4_f32, the 32 x 64 block of
//   the team's W^T A partial in 32 accumulator registers.
//   P stage (slab t + 2): the wave's 16 x 32 partial of A H^T over its 64 columns (32 MFMAs; the A fragments are 16-byte LDS reads, row
//     per lane); the eight waves' partials are summed through LDS (thread e = element e of the 16 x 32 tile) and the member's partial is
//     PUBLISHED: one 8-byte {value, tag} granule per element, one write-through (sc1) store each -- no flag, no fence, no drain
//     (MI355X_MICROARCH.md "R2 granule": an aligned 8-byte sc1 store is observed whole).
//   exchange: every member reads the T partials of a slab (sc1 loads, issued one stage ahead of their use so the round trip through
//     the fabric is behind MFMA work), checks the tags (a late member: re-read that granule, bounded by the wall clock), sums them in
//     member order -- every member of a team, and every run, forms the same bits.
//   W update (slab t): the rows of a slab have ONE owner in the team (member t mod T): it alone reads the old W of the slab -- two stages
//     before anybody may overwrite it, which is why the others do not read it themselves: a faster member's store of the new rows could
//     overtake their load -- forms den = W G + eps on the matrix pipe (two MFMAs per wave) and publishes {w_old}, {den} as two more
//     granule planes beside its partial.  Every member then computes w <- (w_old * ah) / den for all 512 elements (the same bits
//     everywhere), writes it to LDS as the A operand of the next product; the owner also writes it to HBM.
//   Q stage (slab t): P[32 x 64 per wave] += W_new^T A_piece from the LDS copy of the slab (32 MFMAs) -- no second read of A.
//
// A is read ONCE (streamed into registers two slabs ahead, written to the LDS ring by the wave that will read it: the pieces are wave
// private, no barrier guards them); what crosses the team per slab is (T + 2) x 4 KiB of granules against 32 KiB of A per member.  The k x n
// partials of the teams (at most 32) go through the existing fixed-order reduction (launch_reduce) into W^T A; H H^T before and
// W^T W after are the existing Gram launches.
// Residency: every workgroup waits for its team mates, so all of them must be on the device at once.  The launch takes one workgroup per
// CU at most and begins with a CENSUS (an arrival counter all workgroups wait on, bounded by the wall clock): nothing is written before
// every workgroup has been seen running, so a launch that cannot become resident (a co-tenant holds CUs) leaves W untouched, sets the
// sticky time-out word of the persistent kernels (dnmf_hals_sweep_status) and ends.
#pragma once
#include "dnmf_common.h"

namespace {

#define TM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ unsigned int g_team_timeout = 0;   // sticky: a wait of the team kernel gave up (dnmf_hals_sweep_status reports it with the other persistent kernels')

constexpr int TM_R = 16;                 // rows per slab (one MFMA tile of 16x16x4)
constexpr int TM_C = 512;                // columns per team member
constexpr int TM_NW = 8;                 // waves per workgroup, 64 columns each
constexpr int TM_KP = 32;                // padded rank
constexpr int TM_LDA = TM_C + 8;         // LDS row pitch of a slab piece: 130 sixteen-byte chunks = 2 (mod 16) -> the row-per-lane ds_read_b128 is conflict free
constexpr int TM_NBUF = 4;               // slabs in the LDS ring: t (Q), t + 1, t + 2 (P), t + 3 (being written)
constexpr int TM_D = 8;                  // granule ring depth (a member is at most three slabs ahead of the slowest reader: >= 6)
constexpr int TM_LDW = 48;               // pitch of the new W rows in LDS (the two 16-lane halves of a scalar read land on different banks)
constexpr int TM_MAXT = 8;               // members per team at most (n <= 4096)
constexpr size_t TM_LDS_BYTES = (size_t)(TM_NBUF * TM_R * TM_LDA + TM_NW * TM_R * TM_KP + 4 * TM_R * TM_KP + TM_R * TM_LDW) * sizeof(float);

struct TeamArgs {
    const void* A; long lda;             // the data block (fp32), rows 16-byte aligned
    long m; int n, k;
    const float* H; long ldh;            // k x n
    const float* G;                      // H H^T, 32 x 32 zero padded
    float* W; long ldw;                  // m x k, updated in place
    float eps;
    float* P;                            // [teams][32][n] partial W^T A per team
    unsigned long long* ring;            // [teams][TM_D][T + 2][512] granules (T partials, the owner's w_old and den), zeroed before the launch
    unsigned* ctl;                       // ctl[0] census counter, ctl[1] abort word; zeroed before the launch
    int T, tpx;                          // members per team, teams per XCD-residue class (grid = 8 tpx T)
    long rpt;                            // rows per team (a multiple of 16)
    unsigned long long patience;         // ticks of the 100 MHz wall clock a wait may last
};

__device__ __forceinline__ void tm_st64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long tm_ld64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// SD: slabs in flight in registers ahead of the LDS ring (1, 2 or 4); NT: cache policy bits of the loads of A (2 = streaming)
template <int SD, int NT>
__global__ __launch_bounds__(64 * TM_NW, 2) void team_fro_kernel(TeamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tm_smem[];
    float* Ab = tm_smem;                                   // [NBUF][R][LDA]
    float* red = Ab + TM_NBUF * TM_R * TM_LDA;             // [NW][R * KP]   the waves' partials of A H^T
    float* dred = red + TM_NW * TM_R * TM_KP;              // [4][R * KP]    partial denominators (two waves fill one plane)
    float* Wn = dred + 4 * TM_R * TM_KP;                   // [R][LDW]       the slab's new W rows
    __shared__ unsigned s_flag;

    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, q4 = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T;
    const int b = blockIdx.x, xcd = b & 7, qq = b >> 3;
    const int team = xcd * a.tpx + qq / T, member = qq % T;
    const long m = a.m;
    const int n = a.n, k = a.k;
    const long row0 = (long)team * a.rpt;
    const long rows = min(a.rpt, m - row0);
    const int nsl = rows > 0 ? (int)((rows + TM_R - 1) / TM_R) : 0;
    const int cw = 64 * wv;                                // the wave's first column inside the member's piece
    const int cb = member * TM_C + cw;                     // ... inside the matrix

    // ---- census: arrive now, look again after the prologue's loads are under way
    if (tid == 0) __hip_atomic_fetch_add(a.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // loop invariants in registers: the wave's block of H (B operand: k-slot q4, kk = lane & 15), two fragments of G for the denominator
    f32x4 hreg[2][4];
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kk = 16 * tk + i;
            hreg[tk][g] = kk < k ? *reinterpret_cast<const f32x4*>(a.H + (long)kk * a.ldh + cb + 16 * g + 4 * q4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const int dtk = wv & 1, dks = 2 * (wv >> 1);           // this wave's share of W G: tile dtk, contraction steps dks, dks + 1
    float gden[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) gden[u] = a.G[(4 * (dks + u) + q4) * TM_KP + 16 * dtk + i];

    // the slab pieces stream through a MUBUF descriptor at the team's first row: lane offset + scalar slab offset
    const i32x4 arsrc = buf_rsrc((const float*)a.A + row0 * a.lda);
    const int avoff = (int)(((long)q4 * a.lda + cb + 4 * i) * 4);          // row q4 of a group of four, the lane's four columns
    const int arow4 = (int)(a.lda * 16);                                   // bytes between groups of four rows
    f32x4 stg[SD][4];
    auto load_slab = [&](int t, f32x4 (&d)[4]) {                           // 16 x 64 floats of slab t: four rows x 256 bytes per instruction
        const int soff = t * 4 * arow4;
        if ((long)(t + 1) * TM_R <= rows) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) d[ks] = buf_ld_f32x4(arsrc, avoff, soff + ks * arow4, NT);
        } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                d[ks] = buf_ld_f32x4(arsrc, (long)t * TM_R + 4 * ks + q4 < rows ? avoff : BUF_OOB, soff + ks * arow4, NT);
        }
    };
#pragma unroll
    for (int u = 0; u + 1 < SD; ++u)
        if (u < nsl) load_slab(u, stg[u]);

    // W as it is before the update, read by the slab's OWNER only: thread e's own element, and the wave's two A fragments of W G
    const int er = tid >> 5, ekk = tid & 31;               // thread e = element (er, ekk) of a slab's 16 x 32 tile
    float wold[2] = {0.f, 0.f}, wdn[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    auto load_w = [&](int t, float& wo, float (&wd)[2]) {
        const long r = row0 + (long)t * TM_R;
        wo = (r + er < m && ekk < k) ? a.W[(r + er) * a.ldw + ekk] : 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = 4 * (dks + u) + q4;
            wd[u] = (r + i < m && j < k) ? a.W[(r + i) * a.ldw + j] : 0.f;
        }
    };

    // ---- census, second half
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

    f32x4 qacc[2][4];
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int j = 0; j < 4; ++j) qacc[tk][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int EL = TM_R * TM_KP;                       // elements of a slab's tile = granules of a plane = threads
    unsigned long long gat[TM_MAXT + 2];
    unsigned long long* const ring_team = a.ring + (long)team * TM_D * (T + 2) * EL;
    auto issue_gather = [&](int t) {                       // planes 0 .. T - 1: the members' partials; T: w_old; T + 1: den
        const unsigned long long* src = ring_team + (long)(t & (TM_D - 1)) * (T + 2) * EL + tid;
#pragma unroll
        for (int j = 0; j < TM_MAXT; ++j)
            if (j < T) gat[j] = tm_ld64(src + j * EL);
        gat[TM_MAXT] = tm_ld64(src + T * EL);
        gat[TM_MAXT + 1] = tm_ld64(src + (T + 1) * EL);
    };
    // a granule that is not there yet (a late member): read again, bounded by the wall clock
    auto granule = [&](unsigned long long g, const unsigned long long* src, unsigned want) -> float {
        if ((unsigned)(g >> 32) != want) {
            const unsigned long long t0 = wall_clock64();
            unsigned spins = 0;
            for (;;) {
                g = tm_ld64(src);
                if ((unsigned)(g >> 32) == want) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 15u) == 15u && __hip_atomic_load(a.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (wall_clock64() - t0 > a.patience) {
                    __hip_atomic_store(a.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&g_team_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        return __uint_as_float((unsigned)g);
    };
    auto pack = [](float v, int t) { return ((unsigned long long)(unsigned)(t + 1) << 32) | (unsigned long long)__float_as_uint(v); };

    // one stage: u = s mod 4 fixes every LDS ring slot and staging register at compile time
    // owners without a division per stage: o3 / o2 / o0 = (s + 3) / (s + 2) / s modulo T, stepped with s
    int o3 = (T - 1) % T, o2 = (2 * T - 2) % T, o0 = (4 * T - 4) % T;      // s = -4
    auto stage = [&](int s, auto uc) {
        constexpr int u = decltype(uc)::value;
        // S: slab s + 3 from the staging registers into its ring slot (the wave's own 64 columns)
        if (s + 3 >= 0 && s + 3 < nsl) {
            float* dst = Ab + ((u + 3) & 3) * (TM_R * TM_LDA) + q4 * TM_LDA + cw + 4 * i;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) *reinterpret_cast<f32x4*>(dst + 4 * ks * TM_LDA) = stg[(u + 3) % SD][ks];
        }
        // L: slab s + 3 + SD into the registers just freed; the owner of slab s + 3 reads its old W (used by the next stage)
        if (s + 3 + SD < nsl) load_slab(s + 3 + SD, stg[(u + 3) % SD]);
        if (s + 3 >= 0 && s + 3 < nsl && o3 == member) load_w(s + 3, wold[(u + 3) & 1], wdn[(u + 3) & 1]);
        // P: the wave's partial of A H^T for slab s + 2
        const bool pub = s + 2 >= 0 && s + 2 < nsl;
        const bool own = pub && o2 == member;
        if (pub) {
            const float* src = Ab + ((u + 2) & 3) * (TM_R * TM_LDA) + i * TM_LDA + cw + 4 * q4;
            f32x4 pacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(src + 16 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pacc[0] = TM_MFMA(av[e], hreg[0][g][e], pacc[0]);
                    pacc[1] = TM_MFMA(av[e], hreg[1][g][e], pacc[1]);
                }
            }
            float* dst = red + wv * EL + 4 * q4 * TM_KP + i;
#pragma unroll
            for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r * TM_KP + 16 * tk] = pacc[tk][r];
        }
        // the owner's W G for slab s + 2: this wave's tile and contraction steps
        if (own) {
            f32x4 dacc = TM_MFMA(wdn[u & 1][0], gden[0], (f32x4{0.f, 0.f, 0.f, 0.f}));
            dacc = TM_MFMA(wdn[u & 1][1], gden[1], dacc);
            float* dst = dred + (wv >> 1) * EL + 4 * q4 * TM_KP + 16 * dtk + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r * TM_KP] = dacc[r];
        }
        __syncthreads();
        // publish slab s + 2: thread e sums the eight waves' values of element e (the owner adds w_old and den)
        if (pub) {
            float v = red[tid];
#pragma unroll
            for (int w = 1; w < TM_NW; ++w) v += red[w * EL + tid];
            unsigned long long* dst = ring_team + (long)((s + 2) & (TM_D - 1)) * (T + 2) * EL + tid;
            tm_st64(dst + member * EL, pack(v, s + 2));
            if (own) {
                const float den = (((dred[tid] + dred[EL + tid]) + dred[2 * EL + tid]) + dred[3 * EL + tid]) + a.eps;
                tm_st64(dst + T * EL, pack(wold[u & 1], s + 2));
                tm_st64(dst + (T + 1) * EL, pack(den, s + 2));
            }
        }
        // W update of slab s, then the reads of slab s + 1's granules (consumed by the next stage)
        const bool cur = s >= 0 && s < nsl;
        if (cur) {
            const unsigned want = (unsigned)(s + 1);
            const unsigned long long* src = ring_team + (long)(s & (TM_D - 1)) * (T + 2) * EL + tid;
            float ah = 0.f;
#pragma unroll
            for (int j = 0; j < TM_MAXT; ++j)
                if (j < T) ah += granule(gat[j], src + j * EL, want);
            const float wo = granule(gat[TM_MAXT], src + T * EL, want);
            const float den = granule(gat[TM_MAXT + 1], src + (T + 1) * EL, want);
            const float wn = (wo * ah) / den;
            Wn[er * TM_LDW + ekk] = wn;
            const long r = row0 + (long)s * TM_R + er;
            if (o0 == member && r < m && ekk < k) a.W[r * a.ldw + ekk] = wn;
        }
        if (s + 1 >= 0 && s + 1 < nsl) issue_gather(s + 1);
        __syncthreads();
        // Q: the team's W^T A gains the slab (this wave: its 64 columns, from the LDS copy)
        if (cur) {
            float wop[2][4];
#pragma unroll
            for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wop[tk][ks] = Wn[(4 * ks + q4) * TM_LDW + 16 * tk + i];
            const float* src = Ab + u * (TM_R * TM_LDA) + q4 * TM_LDA + cw + 4 * i;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(src + 4 * ks * TM_LDA);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    qacc[0][j] = TM_MFMA(wop[0][ks], bv[j], qacc[0][j]);
                    qacc[1][j] = TM_MFMA(wop[1][ks], bv[j], qacc[1][j]);
                }
            }
        }
        o3 = o3 + 1 == T ? 0 : o3 + 1;
        o2 = o2 + 1 == T ? 0 : o2 + 1;
        o0 = o0 + 1 == T ? 0 : o0 + 1;
    };

    for (int sb = -4; sb < nsl; sb += 4) {
        stage(sb, std::integral_constant<int, 0>{});
        stage(sb + 1, std::integral_constant<int, 1>{});
        stage(sb + 2, std::integral_constant<int, 2>{});
        stage(sb + 3, std::integral_constant<int, 3>{});
    }

    // the team's partial: accumulator register r of tile (tk, j) is row 16 tk + 4 q4 + r, column cb + 4 i + j
    float* Pt = a.P + (long)team * TM_KP * n + cb + 4 * i;
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<f32x4*>(Pt + (long)(16 * tk + 4 * q4 + r) * n) = f32x4{qacc[tk][0][r], qacc[tk][1][r], qacc[tk][2][r], qacc[tk][3][r]};
}

}  // namespace
