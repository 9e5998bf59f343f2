// dnmf_team.hip -- plan and launch of the one-pass MU/Frobenius team kernel (csrc/dnmf_team.h); part of libdnmf_hip.so.
// csrc/dnmf.hip (mu_fro_step_impl) asks dnmf_team_plan_ whether a step of this shape takes it, sizes the workspace with
// dnmf_team_ws_bytes_ and calls dnmf_team_fro_; the reduction of the teams' partials and the H update stay where they were.
#include "dnmf_team.h"
#ifdef DNMF_TUNING
#include "dnmf_team2.h"       // the role-split variant: measured slower (profiles/r06b_team2_*), kept for the A/B only
#endif
#include "dnmf_host.h"

namespace {

struct TeamPlan {
    bool ok;
    int T, tpx, teams, teams_used, kp;
    long rpt;
    size_t p_bytes, pg_off, ctl_off, ring_off, gx_off, total;
};

int g_team_on = 1;                       // dnmf_set_onepass: 0 never, 1 where it measured faster (team_pays), 2 wherever the shape allows
unsigned long long g_team_patience = 200000000ull;

// CUs the census may count on, per device (a process may drive several GPUs): the kernel takes one workgroup per CU; 0 = no device (the
// CPU tier loads the library for its symbol checks) or the kernel cannot be resident at all -- plans then say no
constexpr int TEAM_MAX_DEVICES = 64;
int team_cus() {
    static int cus_of[TEAM_MAX_DEVICES];
    static bool have[TEAM_MAX_DEVICES] = {};
    int dev = 0, cus = 0, nb = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TEAM_MAX_DEVICES) { clear_hip_error(); return 0; }
    if (have[dev]) return cus_of[dev];
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { clear_hip_error(); return 0; }
    const auto kern = team_fro_kernel<2, 2, 2, 2>;
    allow_lds(kern, TM_LDS_BYTES + 64);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 64 * TM_NW, TM_LDS_BYTES) != hipSuccess || nb < 1) { clear_hip_error(); cus = 0; }
    cus_of[dev] = cus;
    have[dev] = true;
    return cus;
}

// Where the one-pass step measured faster than the two passes on an MI355X (tools/team_shapes.sh on the final kernel, third block of
// profiles/r06b_team_shapes.txt): short blocks (two passes are launch- and latency-bound there: 48 / 39 / 26 / 17 % at m = 4096 / 8192 /
// 16384 / 24576, n = 4096; 5-10 % up to 49152), ranks that are not a whole 32-wide tile (the two-pass kernels take their edge paths: 7 % at
// 65536 x 4096, k = 24; 41 % at 16384 x 4096, k = 17) and very tall blocks (1.4-2.9 % from 196608 rows); at k = 32 in between the two
// are within 1 % of each other either way, and the two-pass sequence needs no residency.
// k <= 16 (round 6, KT = 1): the two-pass kernels are HBM-bound there and the one-pass step won at every shape of the sweep (fourth and fifth
// block of the file: 2-46 %, 17-22 % at 65536 ... 262144 x 4096).
// Widths beyond 4096 (teams of 9-16 members, k <= 16 only): the exchange grows with the team (T + 2 planes per slab and member) and ate the
// gain on tall blocks (65536 / 262144 x 8192: 1-3 %; 32768 x 6144, where 12-member teams leave a quarter of the CUs idle: 27 % SLOWER), so only
// short blocks take it there (16384 x 8192: 22 %).
// Inside a BATCHED fit the two-pass kernels cover all problems with every launch (nothing launch-bound is left for the one-pass step to win on
// short blocks, and it runs the problems one after the other): tools/batchbench.py, 8 x 32768 x 4096, k = 16: +24 % iterations per second;
// 8 x 16384 x 4096: +18 %; 8 x 32768 x 2048, k = 8: +14 %; 6 x 65536 x 4096, k = 24: +8 %; 20 x 16384 x 2048, k = 10: -2 % -> from 2^26 elements, k < 32.
bool team_pays(long m, long n, int k, bool batched) {
    if (batched) return n >= 2048 && n <= 4096 && k < 32 && (double)m * n >= 67108864.0;
    if (n > 4096) return m <= 24576;
    return n >= 2048 && (k < 32 || m <= 24576 || (n >= 4096 && m <= 49152) || m >= 196608);
}

// shape-only part of the decision (the workspace query has no pointers); `cus` = 0: ask the device
TeamPlan team_plan(long m, long n, int k, int cus = 0, bool sizing = false, bool batched = false) {
    TeamPlan p{};
    if ((!g_team_on || !dnmf_persistent_on_() || (g_team_on == 1 && !team_pays(m, n, k, batched))) && !sizing) return p;
    if (k < 1 || k > TM_KP || n % 4 != 0 || n < 4 || cdiv(n, TM_C) > (k <= 16 ? 2 * TM_MAXT : TM_MAXT) || m < 4096) return p;
    p.kp = k <= 16 ? 16 : 32;
    if (cus <= 0) cus = team_cus();
    p.T = (int)cdiv(n, TM_C);                                           // the last member's piece may be narrower (its columns beyond n read as zeros)
    p.tpx = (cus / 8) / p.T;
    if (p.tpx < 1) return p;
    p.teams = 8 * p.tpx;
    p.rpt = round_up(cdiv(m, p.teams), TM_R);
    p.teams_used = (int)cdiv(m, p.rpt);
    if (p.teams_used > 64) return p;                                   // one-stage reduction of the partials
    p.p_bytes = align256((size_t)p.teams * p.kp * n * sizeof(float));
    p.pg_off = p.p_bytes;                                              // the teams' partial Gram tiles
    p.ctl_off = p.pg_off + align256((size_t)p.teams * p.kp * p.kp * sizeof(float));
    p.ring_off = p.ctl_off + 256;                                      // ctl | ring | gx: zeroed by ONE memset per launch
    p.gx_off = p.ring_off + (size_t)p.teams * TM_D * (p.T + 2) * (TM_R * p.kp) * sizeof(unsigned long long);
    p.total = p.gx_off + (size_t)p.teams * p.T * (p.kp * p.kp) * sizeof(unsigned long long);
    p.ok = true;
    return p;
}

}  // namespace

__attribute__((visibility("hidden"))) size_t dnmf_team_ws_bytes_(long m, long n, int k) {
    // sized for the largest device this library is built for (256 CUs), so that the answer does not depend on a device query
    const TeamPlan p = team_plan(m, n, k, 256, true);
    return p.ok ? p.total : 0;
}

// 1: not taken (the caller keeps the two-pass sequence); DNMF_OK: W is updated, the teams' partials of W^T A are in
// *P_out ([*nparts][32][n], part of `part`) for the caller's reduction
__attribute__((visibility("hidden"))) int dnmf_team_fro_(const float* A, long m, long n, long lda, float* W, long ldw, const float* H, long ldh,
                                                          const float* G, int k, float eps, void* part, size_t part_bytes, void* stream,
                                                          const float** P_out, int* nparts, const float** Pg_out, int* kp_out) {
    const TeamPlan p = team_plan(m, n, k, 0, false, dnmf_batch_()->B > 1);
    if (!p.ok) return 1;
    // a batched fit (blockIdx.z = problem in the other kernels): the team kernel takes the whole device, so the problems of a batch run one
    // after the other -- each exactly as a fit of its own (bit-identical, as the batched entry points promise)
    const int B = dnmf_batch_()->B;
    if (!(aligned16(A) && lda % 4 == 0 && lda >= n && aligned16(H) && ldh % 4 == 0 && ldh >= n && ldw >= k && aligned16(part))) return 1;
    if ((double)p.rpt * lda * 4.0 >= 2147483648.0 || part_bytes < p.total) return 1;
    hipStream_t st = S(stream);
    for (int z = 0; z < B; ++z) {
    char* base = (char*)batch_ptr(part, z);
    TeamArgs a{};
    a.A = batch_ptr(A, z); a.lda = lda; a.m = m; a.n = (int)n; a.k = k;
    a.H = (const float*)batch_ptr(H, z); a.ldh = ldh; a.G = (const float*)batch_ptr(G, z); a.W = (float*)batch_ptr(W, z); a.ldw = ldw; a.eps = eps;
    a.P = (float*)base;
    a.Pg = (float*)(base + p.pg_off);
    a.gx = (unsigned long long*)(base + p.gx_off);
    a.ctl = (unsigned*)(base + p.ctl_off);
    a.ring = (unsigned long long*)(base + p.ring_off);
    a.T = p.T; a.tpx = p.tpx; a.rpt = p.rpt;
    a.patience = g_team_patience;
    if (hipMemsetAsync(base + p.ctl_off, 0, p.total - p.ctl_off, st) != hipSuccess) return fail(DNMF_EHIP, "team: memset of the granule ring failed");
    static const int sd = (int)tune("DNMF_TEAM_SD", 2), nt = (int)tune("DNMF_TEAM_NT", 2);
    const dim3 grid((unsigned)(8 * p.tpx * p.T)), block(64 * TM_NW);
#ifdef DNMF_TUNING
#define TEAM2_LAUNCH(SD, NT)                                                         \
    do {                                                                             \
        static bool once = false;                                                    \
        if (!once) { allow_lds(team_split_kernel<SD, NT>, TM2_LDS_BYTES + 64); once = true; } \
        hipLaunchKernelGGL((team_split_kernel<SD, NT>), grid, block, TM2_LDS_BYTES, st, a); \
    } while (0)
#endif
#define TEAM_LAUNCH(SD, NT)                                                          \
    do {                                                                             \
        constexpr int SD16 = (SD) == 3 ? 2 : (SD);       /* the k <= 16 kernel's group is four stages long */ \
        static bool once = false;                                                    \
        if (!once) {                                                                 \
            allow_lds(team_fro_kernel<SD, NT, 2, 2>, TM_LDS_BYTES + 64);             \
            allow_lds(team_fro_kernel<SD16, NT, 1, 3>, tm_lds_bytes(16) + 64);       \
            allow_lds(team_fro_kernel<SD16, NT, 1, 3, 2 * TM_MAXT>, tm_lds_bytes(16) + 64); \
            once = true;                                                             \
        }                                                                            \
        if (p.kp == 16 && p.T > TM_MAXT) hipLaunchKernelGGL((team_fro_kernel<SD16, NT, 1, 3, 2 * TM_MAXT>), grid, block, tm_lds_bytes(16), st, a); \
        else if (p.kp == 16) hipLaunchKernelGGL((team_fro_kernel<SD16, NT, 1, 3>), grid, block, tm_lds_bytes(16), st, a); \
        else hipLaunchKernelGGL((team_fro_kernel<SD, NT, 2, 2>), grid, block, TM_LDS_BYTES, st, a); \
    } while (0)
#ifdef DNMF_TUNING
    a.xflags = (int)tune("DNMF_TEAM_X", 0);
    static const int ver = (int)tune("DNMF_TEAM_V", 1);
    if (ver == 2) {
        if (sd == 1) TEAM2_LAUNCH(1, 2);
        else if (sd == 3) TEAM2_LAUNCH(3, 2);
        else TEAM2_LAUNCH(2, 2);
    } else
    if (sd == 1 && nt == 2) TEAM_LAUNCH(1, 2);
    else if (sd == 3 && nt == 2) TEAM_LAUNCH(3, 2);
    else if (sd == 2 && nt == 0) TEAM_LAUNCH(2, 0);
    else
#endif
        TEAM_LAUNCH(2, 2);
#undef TEAM_LAUNCH
    (void)sd; (void)nt;
    if (int rc = check_launch("team_fro_kernel")) return rc;
    }
    *P_out = (const float*)part;                                      // problem 0's (the reduction launch moves to its own problem by blockIdx.z)
    *Pg_out = (const float*)((const char*)part + p.pg_off);
    *kp_out = p.kp;
    *nparts = p.teams_used;
    return DNMF_OK;
}

// read-and-clear of the team kernel's sticky time-out word (csrc/dnmf_hals.hip: dnmf_hals_sweep_status reports it)
__attribute__((visibility("hidden"))) int dnmf_team_timeout_take_(unsigned* out) {
    unsigned v = 0;
    const unsigned zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_team_timeout), sizeof(v), 0, hipMemcpyDeviceToHost) != hipSuccess) return DNMF_EHIP;
    if (v && hipMemcpyToSymbol(HIP_SYMBOL(g_team_timeout), &zero, sizeof(zero), 0, hipMemcpyHostToDevice) != hipSuccess) return DNMF_EHIP;
    *out = v;
    return DNMF_OK;
}

__attribute__((visibility("hidden"))) void dnmf_team_set_patience_(unsigned long long ticks) { g_team_patience = ticks; }

extern "C" {

int dnmf_set_onepass(int on) {
    const int was = g_team_on;
    g_team_on = on < 0 ? 0 : (on > 2 ? 2 : on);
    return was;
}

int dnmf_mu_fro_onepass(long m, long n, int k) { return (m >= 1 && n >= 1 && team_plan(m, n, k).ok) ? 1 : 0; }

}  // extern "C"
