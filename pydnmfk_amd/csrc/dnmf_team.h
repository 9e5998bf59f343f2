// dnmf_team.h -- ONE pass over A per MU/Frobenius iteration for 16 < k <= 32 (round 6): team_fro_kernel.
// Part of libdnmf_hip.so (csrc/dnmf_team.hip plans and launches it; csrc/dnmf.hip: mu_fro_step_impl takes it when the shape allows).
//
// What it computes (dist_nmf.py:716-732 feeding :736-749): for every row r of this rank's block
//     W[r] <- W[r] * (A[r] H^T) / (W[r] (H H^T) + eps)          and, with the NEW rows,          P += W[r]^T A[r]
// -- the W update of a row needs that row's WHOLE A[r] H^T before W^T A may touch the row again, so the row block has to stay on
// chip between its two uses.  A 16-row slab of a 4096-column matrix is 256 KiB: more than one CU's LDS, so a TEAM of T = ceil(n / 512)
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
// partials of the teams (at most 32) go through the existing fixed-order reduction (launch_reduce) into W^T A; H H^T before is the
// existing Gram launch; W_new^T W_new rides along (the slab's owner adds 16 MFMAs on operands that are in registers anyway; the members'
// sums meet once at the end of the launch, the reduction launch's Gram tail adds the teams').
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
constexpr int TM_KP = 32;                // padded rank at 16 < k <= 32 (the kernel's KT = 2; KT = 1: 16), and the pitch of the Gram buffers at every k <= 32
constexpr int TM_LDA = TM_C + 8;         // LDS row pitch of a slab piece: 130 sixteen-byte chunks = 2 (mod 16) -> the row-per-lane ds_read_b128 is conflict free
constexpr int TM_NBUF = 3;               // slabs in the LDS ring at lookahead 2: t (Q), t + 1, t + 2 (written and read by P in the same stage); LA + 1 in general
constexpr int TM_D = 8;                  // granule ring depth (slabs s - 1 .. s + 2 are live while a member publishes s + 2: >= 4)
constexpr int tm_ldw(int kp) { return kp == 16 ? 40 : 48; }   // pitch of the new W rows in LDS: 48 puts the two 16-lane halves of a scalar read on different banks; 40 at k <= 16, whose four-slab ring fills the 160 KiB to the last KiB
constexpr int TM_MAXT = 8;               // members per team at most (n <= 4096)
constexpr int tm_la(int kp) { return kp == 16 ? 3 : 2; }      // lookahead: slabs between the partial (P) and its use (Q); k <= 16 has the LDS for 3
constexpr size_t tm_lds_bytes(int kp, int la = 0) {
    return (size_t)(((la ? la : tm_la(kp)) + 1) * TM_R * TM_LDA + 2 * TM_NW * TM_R * kp + 2 * 4 * TM_R * kp + 2 * TM_R * tm_ldw(kp)) * sizeof(float);
}
constexpr size_t TM_LDS_BYTES = tm_lds_bytes(TM_KP, 2);

struct TeamArgs {
    const void* A; long lda;             // the data block (fp32), rows 16-byte aligned
    long m; int n, k;
    const float* H; long ldh;            // k x n
    const float* G;                      // H H^T, 32 x 32 zero padded
    float* W; long ldw;                  // m x k, updated in place
    float eps;
    float* P;                            // [teams][KP][n] partial W^T A per team (KP = 16 for k <= 16, else 32)
    float* Pg;                           // [teams][KP][KP] partial W_new^T W_new per team (NULL: the caller forms the Gram matrix itself)
    unsigned long long* gx;              // [teams][T][KP * KP] granules: the members' Gram partials meet here at the end (zeroed with the ring)
    unsigned long long* ring;            // [teams][TM_D][T + 2][16 KP] granules (T partials, the owner's w_old and den), zeroed before the launch
    unsigned* ctl;                       // ctl[0] census counter, ctl[1] abort word; zeroed before the launch
    int T, tpx;                          // members per team, teams per XCD-residue class (grid = 8 tpx T)
    long rpt;                            // rows per team (a multiple of 16)
    unsigned long long patience;         // ticks of the 100 MHz wall clock a wait may last
    int xflags;                          // tuning build only (0 in the shipped library): 1 = take granules as they are, 2 = no exchange at all (timing ablations, wrong results)
};

// SD: slabs in flight in registers ahead of the LDS ring (1, 2 or 3); NT: cache policy bits of the loads of A (2 = streaming)
//
// Memory instructions and the in-order counter.  A wave's loads and stores retire in issue order (s_waitcnt vmcnt(N) = "all but my N
// youngest"), and this kernel has three streams per wave with very different latencies: the slab pieces of A (HBM, SD stages ahead),
// the granules (fabric, one stage ahead) and the owner's W elements.  The stage body therefore issues every one of them UNCONDITIONALLY
// and in the order they are consumed -- a member that is not the owner of a slab still executes the owner's loads and stores, with the
// lane offset BUF_OOB (the hardware drops such an access), the gather always reads TM_MAXT + 2 planes (plane min(j, T - 1) for j >= T) --
// so that the compiler's counts are exact and a wait for the granules leaves the younger loads of A in flight.  The first version had
// these accesses under `if (owner)` / `if (j < T)`: the counts became path dependent, every wait degenerated to vmcnt(0..3) and each
// stage paid a full fabric round trip (0.46 ms per launch at 65536 x 4096 against 0.22 ms of matrix time; profiles/r06a_team_*).
// KT = 16-wide tiles of the padded rank: 2 for 16 < k <= 32; 1 for k <= 16, where the two-pass kernels are HBM-bound (AI = k / 2 flop per byte
// against a balance of 20) and reading A once pays most
// LA = lookahead: the partial of slab s + LA is formed and published in stage s, i.e. LA stages before the slab's W update and Q: the
// team's granules have LA - 1 stages to arrive (requested one stage after their publication).  LA + 1 slabs live in LDS: 2 at k > 16 (the
// ring fills the LDS), 3 at k <= 16 -- there a stage is short (0.9 us of MFMA) and one stage did not cover the fabric round trip: the
// exchange cost 0.074 of 0.29 ms (profiles/r06_team_ab.txt)
// MT = members per team the register sets are sized for: 8 (n <= 4096); 16 at k <= 16 only (n <= 8192: the registers allow it there)
template <int SD, int NT, int KT = 2, int LA = 2, int MT = TM_MAXT>
__global__ __launch_bounds__(64 * TM_NW, 2) void team_fro_kernel(TeamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float tm_smem[];
    constexpr int TM_LDW = tm_ldw(16 * KT);
    constexpr int KP = 16 * KT, NB = LA + 1, NG = LA - 1, PER = NB % 2 ? 2 * NB : NB;   // ring slots, granule register sets, stages per unrolled group
    static_assert(PER % SD == 0 && PER % NB == 0 && PER % 2 == 0 && (NG == 1 || PER % NG == 0), "the unrolled group must return every ring to its start");
    constexpr int EL = TM_R * KP;                          // elements of a slab's tile = granules of a plane (512 = one per thread; KT = 1: 256, waves 0-3)
    float* Ab = tm_smem;                                   // [NBUF][R][LDA]
    float* red = Ab + NB * TM_R * TM_LDA;             // [2][NW][EL]   the waves' partials of A H^T (by stage parity)
    float* dred = red + 2 * TM_NW * EL;                    // [2][4][EL]    partial denominators (two waves fill one plane)
    float* Wn = dred + 2 * 4 * EL;                         // [2][R][LDW]   the slab's new W rows
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
    const int nfull = rows > 0 ? (int)(rows / TM_R) : 0;   // slabs with all 16 rows
    const int cw = 64 * wv;                                // the wave's first column inside the member's piece
    const int cb = member * TM_C + cw;                     // ... inside the matrix

    // ---- census: arrive now, look again after the prologue's loads are under way
    if (tid == 0) __hip_atomic_fetch_add(a.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // loop invariants in registers: the wave's block of H (B operand: k-slot q4, kk = lane & 15), two fragments of G for the denominator
    f32x4 hreg[KT][4];
#pragma unroll
    for (int tk = 0; tk < KT; ++tk)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kk = 16 * tk + i;
            // (columns beyond n -- the last member of a matrix whose width is not a multiple of 512: n % 4 == 0, so a lane's four are in or out -- are zeros)
            hreg[tk][g] = (kk < k && cb + 16 * g + 4 * q4 < n) ? *reinterpret_cast<const f32x4*>(a.H + (long)kk * a.ldh + cb + 16 * g + 4 * q4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    // this wave's share of the owner's W G (16 x KP, contraction over KP): KT = 2: tile wv & 1, steps 2 (wv >> 1), + 1 -> plane wv >> 1;
    // KT = 1: one tile, step wv of four (waves 0-3) -> plane wv.  The four planes add up to the product either way.
    constexpr int ND = KT;                                 // MFMAs of this share
    const bool dact = KT == 2 || wv < 4;
    const int dtk = KT == 2 ? (wv & 1) : 0, dks = KT == 2 ? 2 * (wv >> 1) : (wv & 3), dpl = KT == 2 ? (wv >> 1) : (wv & 3);
    float gden[ND];
#pragma unroll
    for (int u = 0; u < ND; ++u) gden[u] = a.G[(4 * (dks + u) + q4) * TM_KP + 16 * dtk + i];

    // MUBUF descriptors at the team's first row / the team's granules: one lane offset per stream, everything else scalar
    const i32x4 arsrc = buf_rsrc((const float*)a.A + row0 * a.lda);
    const int avoff = cb + 4 * i < n ? (int)(((long)q4 * a.lda + cb + 4 * i) * 4) : BUF_OOB;   // row q4 of a group of four, the lane's four columns (beyond n: zeros)
    const int arow4 = (int)(a.lda * 16);                                   // bytes between groups of four rows
    const i32x4 wrsrc = buf_rsrc(a.W + row0 * a.ldw);
    const bool eon = tid < EL;                                             // thread e = element (er, ekk) of a slab's 16 x KP tile (KT = 1: waves 0-3)
    const int er = tid / KP, ekk = tid % KP, te = eon ? tid : 0;
    const int wevoff = (eon && ekk < k) ? (int)((er * a.ldw + ekk) * 4) : BUF_OOB;   // thread e's element of a slab of W
    int wdvoff[ND];                                                        // the wave's A fragments of the product W G
#pragma unroll
    for (int u = 0; u < ND; ++u) wdvoff[u] = (dact && 4 * (dks + u) + q4 < k) ? (int)((i * a.ldw + 4 * (dks + u) + q4) * 4) : BUF_OOB;
    const int wslab = (int)(a.ldw * 4 * TM_R);                             // bytes between slabs of W
    const i32x4 rrsrc = buf_rsrc(a.ring + (long)team * TM_D * (T + 2) * EL);
    const int gvoff = eon ? tid * 8 : BUF_OOB;                             // (threads without an element: every granule access is dropped)
    constexpr int PLANE = EL * 8;
    const int slotb = (T + 2) * PLANE;                                     // bytes of a ring slot

    f32x4 stg[SD][4];
    float wold[2] = {0.f, 0.f}, wdn[2][ND] = {};
    f32x2 gat[NG][MT + 2];                                           // granules of the slabs s + 1 .. s + LA - 1 on their way
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int j = 0; j < MT + 2; ++j) gat[g][j] = f32x2{0.f, 0.f};

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

    f32x4 qacc[KT][4];
#pragma unroll
    for (int tk = 0; tk < KT; ++tk)
#pragma unroll
        for (int j = 0; j < 4; ++j) qacc[tk][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool plain = (a.xflags & 4) != 0;
    // W_new^T W_new (the Gram matrix of the H phase, dist_nmf.py:748) rides along: the new rows of a slab are in every member's LDS, and as
    // the A operand of Q they ARE both operands of the product -- the slab's owner adds its 16 MFMAs (KT = 2, two per wave: tile (wv >> 1 & 1,
    // wv & 1), contraction steps 2 (wv >> 2), + 1; KT = 1: four MFMAs, one in each of the waves 0-3: step wv), no loads
    f32x4 gacc = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool gt1 = KT == 2 && ((wv >> 1) & 1), gt2 = KT == 2 && (wv & 1), gkh = (wv >> 2) != 0;

    // granules that are not there yet (a late member): read the slab's planes again until they are, bounded by the wall clock
    auto regather = [&](int t, f32x2 (&gat)[MT + 2]) __attribute__((always_inline)) {
        const unsigned want = (unsigned)(t + 1);
        const int slot = __builtin_amdgcn_readfirstlane((t & (TM_D - 1)) * slotb);
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < MT + 2; ++j) {
                const int pl = j < MT ? (j < T ? j : T - 1) : T + (j - MT);
                gat[j] = buf_ld_f32x2(rrsrc, gvoff, slot + pl * PLANE, 16);
            }
#pragma unroll
            for (int j = 0; j < MT + 2; ++j) ok = ok && __float_as_uint(gat[j][1]) == want;
            if (ok) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 15u) == 15u && __hip_atomic_load(a.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (wall_clock64() - t0 > a.patience) {
                __hip_atomic_store(a.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&g_team_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    };

    float tmask[MT];                                                  // 1 for the planes of real members (uniform values)
#pragma unroll
    for (int j = 0; j < MT; ++j) tmask[j] = j < T ? 1.f : 0.f;
    // owners without a division per stage: o3 / o2 / o0 = (s + LA + 1) / (s + LA) / s modulo T, stepped with s
    constexpr int S0 = -2 * PER;                                           // first stage: the pipeline fills in the two guarded groups
    int o3 = ((S0 + LA + 1) % T + T) % T, o2 = ((S0 + LA) % T + T) % T, o0 = (S0 % T + T) % T;
    // One stage.  U = s mod PER fixes every LDS slot and staging register at compile time; GD: the guarded form (first and last
    // stages of a team's rows: slabs that do not exist, a last slab with fewer than 16 rows) -- the steady form has no such test.
    // ONE barrier per stage: before it the wave stages slab s + LA and forms its partial (matrix work that needs nothing from the
    // team), then finishes the W update of slab s from the granules it asked for a stage ago; after it the partial is published,
    // the next slab's granules are requested and the product W_new^T A of slab s runs.
    auto stage = [&](int s, auto uc, auto gc) __attribute__((always_inline)) {
        constexpr int U = decltype(uc)::value, PAR = U & 1;
        constexpr bool GD = decltype(gc)::value;
        float* const redp = red + PAR * (TM_NW * EL);
        float* const dredp = dred + PAR * (4 * EL);
        float* const Wnp = Wn + PAR * (TM_R * TM_LDW);
        const bool pub = !GD || (s + LA >= 0 && s + LA < nsl);
        f32x2 (&gcur)[MT + 2] = gat[U % NG];                           // slab s's granules (requested LA - 1 stages ago); refilled below for slab s + LA - 1
        const bool cur = !GD || (s >= 0 && s < nsl);
        // S: slab s + LA from the staging registers into its ring slot (the wave's own 64 columns)
        if (pub) {
            float* dst = Ab + ((U + LA) % NB) * (TM_R * TM_LDA) + q4 * TM_LDA + cw + 4 * i;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) *reinterpret_cast<f32x4*>(dst + 4 * ks * TM_LDA) = stg[(U + LA) % SD][ks];
        }
        // L: slab s + LA + SD into the registers just freed (four rows x 256 bytes per instruction)
        {
            const int t = s + LA + SD;
            if (!GD || (t >= 0 && t < nsl)) {
                const int soff = __builtin_amdgcn_readfirstlane(t * 4 * arow4);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int vo = (!GD || (long)t * TM_R + 4 * ks + q4 < rows) ? avoff : BUF_OOB;
                    stg[(U + LA) % SD][ks] = buf_ld_f32x4(arsrc, vo, soff + ks * arow4, NT);
                }
            }
        }
        // the old W of slab s + LA + 1, for its owner (used by the next stage; everybody else reads nothing: BUF_OOB)
        {
            const int t = s + LA + 1;
            if (!GD || (t >= 0 && t < nsl)) {
                const bool mine = o3 == member;
                // (scalar offsets are pinned to SGPRs: under scalar-register pressure hipcc kept these induction values in vector registers and
                // wrapped every access in a waterfall loop -- 0.385 -> 0.414 ms per step at config 2 when the kernel gained its KT parameter)
                const int soff = __builtin_amdgcn_readfirstlane(t * wslab);
                wold[(U + LA + 1) & 1] = buf_ld_f32(wrsrc, (mine && (!GD || (long)t * TM_R + er < rows)) ? wevoff : BUF_OOB, soff, 0);
#pragma unroll
                for (int u = 0; u < ND; ++u)
                    wdn[(U + LA + 1) & 1][u] = buf_ld_f32(wrsrc, (mine && (!GD || (long)t * TM_R + i < rows)) ? wdvoff[u] : BUF_OOB, soff, 0);
            }
        }
        // P, first half: the wave's partial of A H^T for slab s + LA over its first 32 columns
        f32x4 pacc[KT];
#pragma unroll
        for (int tk = 0; tk < KT; ++tk) pacc[tk] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* psrc = Ab + ((U + LA) % NB) * (TM_R * TM_LDA) + i * TM_LDA + cw + 4 * q4;
        auto p_half = [&](int g0) {
#pragma unroll
            for (int g = g0; g < g0 + 2; ++g) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(psrc + 16 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int tk = 0; tk < KT; ++tk) pacc[tk] = TM_MFMA(av[e], hreg[tk][g][e], pacc[tk]);
            }
        };
        if (pub) p_half(0);
        __builtin_amdgcn_sched_barrier(0);
        // W update of slab s from the granules read a stage ago -- between the two halves of P, so that neither the wait for the granules nor
        // the LDS round trip behind it leaves the matrix pipe without queued work.  All tags right <=> their sum is (MT + 2) x the tag:
        // a slot's earlier contents carry SMALLER tags (zero after the launch's memset, then s + 1 - 8, s + 1 - 16, ...), never larger ones.
        if (cur) {
            const unsigned want = (unsigned)(s + 1);
            unsigned tsum = __float_as_uint(gcur[0][1]);
#pragma unroll
            for (int j = 1; j < MT + 2; ++j) tsum += __float_as_uint(gcur[j][1]);
            if (eon && tsum != (MT + 2) * want && !(a.xflags & 1)) regather(s, gcur);
            float ah = gcur[0][0];
#pragma unroll
            for (int j = 1; j < MT; ++j) ah = fmaf(tmask[j], gcur[j][0], ah);     // planes j >= T: x 0 (a re-read of plane T - 1)
            const float wn = div_pos(gcur[MT][0] * ah, gcur[MT + 1][0]);
            if (eon) Wnp[er * TM_LDW + ekk] = wn;
            const bool st = o0 == member && (!GD || (long)s * TM_R + er < rows);
            buf_st_f32(wn, wrsrc, st ? wevoff : BUF_OOB, __builtin_amdgcn_readfirstlane(s * wslab), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (pub) {
            p_half(2);
            float* dst = redp + wv * EL + 4 * q4 * KP + i;
#pragma unroll
            for (int tk = 0; tk < KT; ++tk)
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[r * KP + 16 * tk] = pacc[tk][r];
            // the owner's W G for slab s + LA: this wave's tile and contraction steps
            if (o2 == member && dact) {
                f32x4 dacc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < ND; ++u) dacc = TM_MFMA(wdn[(U + LA) & 1][u], gden[u], dacc);
                float* dd = dredp + dpl * EL + 4 * q4 * KP + 16 * dtk + i;
#pragma unroll
                for (int r = 0; r < 4; ++r) dd[r * KP] = dacc[r];
            }
        }
        __syncthreads();
        // after the barrier: every LDS read of this half first (the eight partials and four denominator parts of element e, the new W rows
        // as A operand, the slab piece as B operand), then half of Q, the publication and the request for the next granules, the other half
        float rr[TM_NW], dd[4], wop[KT][4];
        f32x4 bv[4];
        const float* qsrc = Ab + (U % NB) * (TM_R * TM_LDA) + q4 * TM_LDA + cw + 4 * i;
        if (cur) {
#pragma unroll
            for (int tk = 0; tk < KT; ++tk)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) wop[tk][ks] = Wnp[(4 * ks + q4) * TM_LDW + 16 * tk + i];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) bv[ks] = *reinterpret_cast<const f32x4*>(qsrc + 4 * ks * TM_LDA);
        }
        if (pub) {
#pragma unroll
            for (int w = 0; w < TM_NW; ++w) rr[w] = redp[w * EL + te];
#pragma unroll
            for (int w = 0; w < 4; ++w) dd[w] = dredp[w * EL + te];
        }
        auto q_half = [&](int k0) {                        // Q: the team's W^T A gains slab s (this wave: its 64 columns, from the LDS copy)
#pragma unroll
            for (int ks = k0; ks < k0 + 2; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tk = 0; tk < KT; ++tk) qacc[tk][j] = TM_MFMA(wop[tk][ks], bv[ks][j], qacc[tk][j]);
        };
        if (cur) q_half(0);
        __builtin_amdgcn_sched_barrier(0);
        // publish slab s + LA: thread e sums the eight waves' values of element e (the owner adds w_old and den)
        if (pub) {
            float v = rr[0];
#pragma unroll
            for (int w = 1; w < TM_NW; ++w) v += rr[w];
            const float den = (((dd[0] + dd[1]) + dd[2]) + dd[3]) + a.eps;
            const float tag = __uint_as_float((unsigned)(s + LA + 1));
            const int slot = __builtin_amdgcn_readfirstlane(((s + LA) & (TM_D - 1)) * slotb);
            const int ov = (o2 == member && !(a.xflags & 2)) ? gvoff : BUF_OOB;
            if (plain) {        // the whole team on one XCD: its L2 is the point of coherence, the granules need not leave it
                buf_st_f32x2(f32x2{v, tag}, rrsrc, (a.xflags & 2) ? BUF_OOB : gvoff, slot + member * PLANE, 0);
                buf_st_f32x2(f32x2{wold[(U + LA) & 1], tag}, rrsrc, ov, slot + T * PLANE, 0);
                buf_st_f32x2(f32x2{den, tag}, rrsrc, ov, slot + (T + 1) * PLANE, 0);
            } else {
                buf_st_f32x2(f32x2{v, tag}, rrsrc, (a.xflags & 2) ? BUF_OOB : gvoff, slot + member * PLANE, 16);
                buf_st_f32x2(f32x2{wold[(U + LA) & 1], tag}, rrsrc, ov, slot + T * PLANE, 16);
                buf_st_f32x2(f32x2{den, tag}, rrsrc, ov, slot + (T + 1) * PLANE, 16);
            }
        }
        // the reads of slab s + LA - 1's granules (published a stage ago, used LA - 1 stages from now) into the set slab s has just left
        if ((!GD || (s + LA - 1 >= 0 && s + LA - 1 < nsl)) && !(a.xflags & 2)) {
            const int slot = __builtin_amdgcn_readfirstlane(((s + LA - 1) & (TM_D - 1)) * slotb);
#pragma unroll
            for (int j = 0; j < MT + 2; ++j) {
                const int pl = j < MT ? (j < T ? j : T - 1) : T + (j - MT);
                gcur[j] = buf_ld_f32x2(rrsrc, gvoff, slot + pl * PLANE, 16);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (cur) q_half(2);
        if (cur && a.Pg && o0 == member) {
            if constexpr (KT == 2) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float x1 = gkh ? (gt1 ? wop[1][2 + u] : wop[0][2 + u]) : (gt1 ? wop[1][u] : wop[0][u]);
                    const float x2 = gkh ? (gt2 ? wop[1][2 + u] : wop[0][2 + u]) : (gt2 ? wop[1][u] : wop[0][u]);
                    gacc = TM_MFMA(x1, x2, gacc);
                }
            } else if (wv < 4) {
                const float x = (wv & 2) ? ((wv & 1) ? wop[0][3] : wop[0][2]) : ((wv & 1) ? wop[0][1] : wop[0][0]);
                gacc = TM_MFMA(x, x, gacc);
            }
        }
        o3 = o3 + 1 == T ? 0 : o3 + 1;
        o2 = o2 + 1 == T ? 0 : o2 + 1;
        o0 = o0 + 1 == T ? 0 : o0 + 1;
    };
    auto group = [&](int sb, auto gc) __attribute__((always_inline)) {
        stage(sb, std::integral_constant<int, 0>{}, gc);
        stage(sb + 1, std::integral_constant<int, 1>{}, gc);
        stage(sb + 2, std::integral_constant<int, 2>{}, gc);
        stage(sb + 3, std::integral_constant<int, 3>{}, gc);
        if constexpr (PER == 6) {
            stage(sb + 4, std::integral_constant<int, 4>{}, gc);
            stage(sb + 5, std::integral_constant<int, 5>{}, gc);
        }
        static_assert(PER == 4 || PER == 6, "group");
    };

    int sb = S0;
    group(sb, std::true_type{});                                           // the pipeline fills
    group(sb + PER, std::true_type{});
    sb = 0;
    for (; sb + PER - 1 + LA + SD < nfull; sb += PER) group(sb, std::false_type{});   // every slab a stage touches exists and is whole
    for (; sb < nsl; sb += PER) group(sb, std::true_type{});               // the pipeline drains

    // the team's partial: accumulator register r of tile (tk, j) is row 16 tk + 4 q4 + r, column cb + 4 i + j
    float* Pt = a.P + (long)team * KP * n + cb + 4 * i;
#pragma unroll
    for (int tk = 0; tk < KT; ++tk)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (cb + 4 * i < n)
                *reinterpret_cast<f32x4*>(Pt + (long)(16 * tk + 4 * q4 + r) * n) = f32x4{qacc[tk][0][r], qacc[tk][1][r], qacc[tk][2][r], qacc[tk][3][r]};

    // the Gram partials: the two contraction halves of a tile meet in LDS, the members' sums in the team's granule planes (once per
    // launch), member 0 adds them in member order and writes the team's 32 x 32 partial for the reduction launch's Gram tail
    if (a.Pg) {
        __syncthreads();
        constexpr int GE = KP * KP, NPL = KT == 2 ? 2 : 4, GH = (GE + 511) / 512;      // elements of the tile, partial planes, elements per thread
        float* gl = red;                                   // [NPL][GE]
        if (KT == 2 || wv < 4) {
            float* dst = gl + (KT == 2 ? (wv >> 2) : wv) * GE + (16 * (int)gt1 + 4 * q4) * KP + 16 * (int)gt2 + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r * KP] = gacc[r];
        }
        __syncthreads();
        const i32x4 grsrc = buf_rsrc(a.gx + (long)team * T * GE);
        const float gtag = __uint_as_float(0x7fffffffu);
#pragma unroll
        for (int h = 0; h < GH; ++h) {
            const int e = tid + 512 * h;
            float v = gl[e < GE ? e : 0];
#pragma unroll
            for (int pl = 1; pl < NPL; ++pl) v += gl[pl * GE + (e < GE ? e : 0)];
            buf_st_f32x2(f32x2{v, gtag}, grsrc, e < GE ? e * 8 : BUF_OOB, member * (GE * 8), 16);
        }
        if (member == 0 && tid < GE) {                       // (GE < 512: whole waves drop out)
            const unsigned long long t0 = wall_clock64();
            f32x2 g[GH][MT];
            auto fetch = [&]() {                           // all planes of the thread's elements in flight together
#pragma unroll
                for (int h = 0; h < GH; ++h)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        g[h][j] = buf_ld_f32x2(grsrc, (tid + 512 * h) * 8, (j < T ? j : T - 1) * (GE * 8), 16);
            };
            unsigned spins = 0;
            for (;;) {
                fetch();
                bool ok = true;
#pragma unroll
                for (int h = 0; h < GH; ++h)
#pragma unroll
                    for (int j = 0; j < MT; ++j) ok = ok && __float_as_uint(g[h][j][1]) == 0x7fffffffu;
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 15u) == 15u && __hip_atomic_load(a.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                if (wall_clock64() - t0 > a.patience) {
                    __hip_atomic_store(a.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&g_team_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
#pragma unroll
            for (int h = 0; h < GH; ++h) {
                float sum = g[h][0][0];
#pragma unroll
                for (int j = 1; j < MT; ++j) sum = fmaf(tmask[j], g[h][j][0], sum);
                a.Pg[(long)team * GE + tid + 512 * h] = sum;
            }
        }
    }
}

}  // namespace
