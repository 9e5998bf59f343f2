// dnmf_hals.hip -- C ABI of the HALS sweeps (csrc/dnmf_hals.h).  A translation unit of its own (see csrc/dnmf_kl.hip).
#include <mutex>

#include "dnmf_common.h"
#include "dnmf_host.h"
#include "dnmf_stream.h"
#include "dnmf_update.h"
#include "dnmf_hals.h"

// csrc/dnmf_wide.hip
__attribute__((visibility("hidden"))) int dnmf_wide_hals_update_h_(float* H, int k, long n, long ldh, const float* AtW, long ldatw, const float* G,
                                                                   long ldg, float eps, void* stream);

extern "C" {

static int hals_w_col_launch(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, int kk,
                             const double* prev_ss2, float eps, double* ss2_out, bool zero, void* stream) {
    const int kpg = kp_of(k);                                       // (the column kernel takes any rank: G's pitch is all it needs)
    REQUIRE(kpg > 0 && W && AH && G && ss2_out && m >= 1 && ldw >= k && ldah >= k && kk >= 0 && kk < k, "hals_w_col: bad arguments");
    hipStream_t st = S(stream);
    if (zero && batch_memset(ss2_out, 0, sizeof(double), st) != hipSuccess) return fail(DNMF_EHIP, "hals_w_col: memset failed");
    const unsigned grid = (unsigned)std::min<long>(cdiv(m, 256), 2048);
    DNMF_LAUNCH(hals_w_col_kernel, dim3(grid), dim3(256), 0, st, W, m, k, ldw, AH, ldah, G, kpg, kk, prev_ss2,
                       eps, ss2_out);
    return check_launch("hals_w_col");
}

int dnmf_hals_w_col(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, int kk,
                    const double* prev_ss2, float eps, double* ss2_out, void* stream) {
    return hals_w_col_launch(W, m, k, ldw, AH, ldah, G, kk, prev_ss2, eps, ss2_out, true, stream);
}

int dnmf_hals_w_scale(float* W, long m, long ldw, int col, const double* ss2, void* stream) {
    REQUIRE(W && ss2 && m >= 1 && col >= 0 && ldw > col, "hals_w_scale: bad arguments");
    const unsigned grid = (unsigned)std::min<long>(cdiv(m, 256), 2048);
    DNMF_LAUNCH(hals_w_scale_kernel, dim3(grid), dim3(256), 0, S(stream), W, m, ldw, col, ss2);
    return check_launch("hals_w_scale");
}

int dnmf_hals_update_w(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, float eps,
                       double* ss2, void* stream) {
    REQUIRE(ss2 != nullptr && k >= 1, "hals_update_w: ss2 scratch (k doubles) required");
    if (batch_memset(ss2, 0, (size_t)k * sizeof(double), S(stream)) != hipSuccess) return fail(DNMF_EHIP, "hals_update_w: memset failed");
    for (int kk = 0; kk < k; ++kk) {   // one launch per column: the column norm is a grid-wide dependency
        int rc = hals_w_col_launch(W, m, k, ldw, AH, ldah, G, kk, kk ? ss2 + kk - 1 : nullptr, eps, ss2 + kk, false, stream);
        if (rc) return rc;
    }
    return dnmf_hals_w_scale(W, m, ldw, k - 1, ss2 + k - 1, stream);
}

}  // extern "C"
namespace {
constexpr int DNMF_MAX_DEVICES = 64;
// co-residency of the persistent sweep: workgroups the device can hold at once
template <typename K>
long resident_workgroups(K kernel, int threads, size_t lds) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds) != hipSuccess) return 0;
    return (long)cus * per_cu;
}

// pe != nullptr: the cross-rank sweep (HalsPeers, dnmf_common.h) -- the slot slabs are the peers' exported ones (pe->slab), this
// rank's slab of the sweep's parity is reset AFTER the kernel (see dnmf_comm.hip).  check_only: no launch, 0 = the sweep applies
template <int KT>
int launch_hals_sweep(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, float eps,
                      unsigned long long* slab, double* ss2, float* T, hipStream_t st, const HalsPeers* pe = nullptr,
                      bool check_only = false) {
    constexpr int KP = 32 * KT;
    const long ldt = KP;
    const bool vecw = aligned16(W) && aligned16(AH) && ldw % 4 == 0 && ldah % 4 == 0 && k % 4 == 0;   // pass 1 reads W, AH
    // pass 2: T rows are aligned.  Only KP = 64 has a 16-byte variant of the row load / store (once per sweep): hipcc
    // register-allocates the KP = 32 one pathologically (the whole row in scratch, 20000 spills), and every variant of the
    // fully expanded KP = 128 sweep costs a minute of build time.
    constexpr bool HASVEC = KT == 2;
    const bool vec = HASVEC && aligned16(W) && ldw % 4 == 0 && k % 4 == 0;
    const long grid = cdiv(m, HALS_WG);
    constexpr size_t lds = (size_t)KP * KP * sizeof(float);          // G staged per workgroup
    // co-residency capacity per DEVICE (a process may drive several GPUs, or a CU-masked one), filled once per device under a
    // lock (ctypes callers release the GIL)
    static std::mutex mu;
    static long cap_v[DNMF_MAX_DEVICES], cap_s[DNMF_MAX_DEVICES];
    static bool have[DNMF_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DNMF_MAX_DEVICES) return 1;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!have[dev]) {
            allow_lds(hals_w_sweep_kernel<KP, HASVEC>, lds); allow_lds(hals_w_sweep_kernel<KP, false>, lds);
            cap_v[dev] = resident_workgroups(hals_w_sweep_kernel<KP, HASVEC>, HALS_WG, lds);
            cap_s[dev] = resident_workgroups(hals_w_sweep_kernel<KP, false>, HALS_WG, lds);
            have[dev] = true;
        }
    }
    // a batch whose workgroups do not all fit the device runs the sweep kernel on as many problems at a time as do (round 5: it took the
    // column launches instead -- 20 problems of 65536 rows, BASELINE config 5: 41 us x k per step -- and a batched fit then differed
    // from single fits, which do fit)
    BatchCtx* bc = dnmf_batch_();
    const int B = bc->B;
    const long cap = vec ? cap_v[dev] : cap_s[dev];
    if (grid > HALS_MAX_WG || grid > cap || (pe && B > 1) || !dnmf_persistent_on_()) return 1;   // not applicable (or switched off): the caller takes the column path
    const int per = (int)std::min<long>(B, cap / grid);
    if (ldw >= (1L << 23) || ldah >= (1L << 23)) return 1;                        // beyond the 32-bit tile offsets of pass 1: column path
    if (check_only) return 0;
    if (!pe && batch_memset(slab, 0xff, (size_t)k * HALS_MAX_WG * sizeof(unsigned long long), st) != hipSuccess)
        return fail(DNMF_EHIP, "hals_sweep_w: memset failed");
    HalsPeers peers{};                       // P = 0: the local sweep
    if (pe) peers = *pe;
    {   // pass 1: T = AH - W G' (G' = G masked to l > j), the W-update kernel in its HALS mode
        constexpr size_t lds1 = (size_t)KP * (KP + 4) * sizeof(float);
        constexpr int OCC = KT == 4 ? 4 : 5;
        static bool once = false;
        if (!once) {
            allow_lds(update_w_seq_kernel<KT, 4, OCC, false, UW_HALS_T>, lds1); allow_lds(update_w_seq_kernel<KT, 4, OCC, true, UW_HALS_T>, lds1);
            allow_lds(update_w_seq_kernel<KT, 1, OCC, true, UW_HALS_T>, lds1);
            once = true;
        }
        const unsigned g1 = upd_grid(cdiv(m, 32), KT);
        constexpr unsigned T1 = 64 * upd_waves(KT);
        if (vecw && k == KP && m % 32 == 0)
            DNMF_LAUNCH((update_w_seq_kernel<KT, 4, OCC, false, UW_HALS_T>), dim3(g1), dim3(T1), lds1, st, W, m, k, ldw, AH, ldah, G, eps, T, ldt);
        else if (vecw)
            DNMF_LAUNCH((update_w_seq_kernel<KT, 4, OCC, true, UW_HALS_T>), dim3(g1), dim3(T1), lds1, st, W, m, k, ldw, AH, ldah, G, eps, T, ldt);
        else
            DNMF_LAUNCH((update_w_seq_kernel<KT, 1, OCC, true, UW_HALS_T>), dim3(g1), dim3(T1), lds1, st, W, m, k, ldw, AH, ldah, G, eps, T, ldt);
        int rc = check_launch("hals_sweep_w(transform)");
        if (rc) return rc;
    }
    static const int dbg = (int)tune("DNMF_HALS_DBG", 0);     // tuning build: 1 = no grid exchange (timing experiment, wrong norms)
    int rc_sweep = DNMF_OK;
    for (int z0 = 0; z0 < B && !rc_sweep; z0 += per) {
        bc->B = std::min(per, B - z0);
        bc->tab.z0 = z0;
        if (vec) DNMF_LAUNCH((hals_w_sweep_kernel<KP, HASVEC>), dim3((unsigned)grid), dim3(HALS_WG), lds, st, W, m, k, ldw, (const float*)T, ldt, G, eps, slab, ss2, dbg, peers);
        else DNMF_LAUNCH((hals_w_sweep_kernel<KP, false>), dim3((unsigned)grid), dim3(HALS_WG), lds, st, W, m, k, ldw, (const float*)T, ldt, G, eps, slab, ss2, dbg, peers);
        rc_sweep = check_launch("hals_sweep_w");
    }
    bc->B = B;
    bc->tab.z0 = 0;
    if (rc_sweep) return rc_sweep;
    // cross-rank: this rank's slab of this parity goes back to "empty" behind the sweep that used it -- a peer writes the sweep
    // after next into it only after finishing the next one, which needs this rank's next sweep, which is enqueued behind this fill
    if (pe && hipMemsetAsync(pe->slab[pe->rank], 0xff, (size_t)k * HALS_MAX_WG * sizeof(unsigned long long), st) != hipSuccess)
        return fail(DNMF_EHIP, "hals_sweep_w: reset of the peer slab failed");
    return DNMF_OK;
}
}  // namespace
extern "C" {

int dnmf_hals_sweep_w(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, float eps, void* ws,
                      size_t ws_bytes, void* stream) {
    const int kt = kt_of(k);                                        // (-1 for a wide rank: the column launches below)
    const int kp = kp_of(k);
    REQUIRE(kp > 0 && W && AH && G && ws && m >= 1 && ldw >= k && ldah >= k, "hals_sweep_w: bad arguments");
    const size_t slab_bytes = (size_t)kp * HALS_MAX_WG * sizeof(unsigned long long);
    const size_t t_off = align256(slab_bytes + (size_t)kp * sizeof(double));
    if (ws_bytes < slab_bytes + (size_t)kp * sizeof(double)) return fail(DNMF_EWS, "hals_sweep_w: workspace too small");
    unsigned long long* slab = (unsigned long long*)ws;
    double* ss2 = (double*)((char*)ws + slab_bytes);
    float* T = (float*)((char*)ws + t_off);
    static const int mode = (int)tune("DNMF_HALS_SWEEP", 1);     // 0: always the column-per-launch path (A/B runs)
    int rc = 1;
    if (mode && kt > 0 && ws_bytes >= t_off + (size_t)m * kp * sizeof(float)) {
        hipStream_t st = S(stream);
        if (kt == 1) rc = launch_hals_sweep<1>(W, m, k, ldw, AH, ldah, G, eps, slab, ss2, T, st);
        else if (kt == 2) rc = launch_hals_sweep<2>(W, m, k, ldw, AH, ldah, G, eps, slab, ss2, T, st);
        else rc = launch_hals_sweep<4>(W, m, k, ldw, AH, ldah, G, eps, slab, ss2, T, st);
    }
    if (rc != 1) return rc;
    return dnmf_hals_update_w(W, m, k, ldw, AH, ldah, G, eps, ss2, stream);   // too many rows to keep resident: one launch per column
}

}  // extern "C"

// library-internal (csrc/dnmf_comm.hip): the persistent sweep with the column norms summed over the ranks of `pe`.
// pe == nullptr: only tell whether the sweep applies to this shape on this device (0) or not (1); returns the workgroup count in *nwg
__attribute__((visibility("hidden"))) int dnmf_hals_sweep_w_peers_(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G,
                                                                   float eps, void* ws, size_t ws_bytes, const HalsPeers* pe, int* nwg,
                                                                   void* stream);
int dnmf_hals_sweep_w_peers_(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, float eps, void* ws, size_t ws_bytes,
                             const HalsPeers* pe, int* nwg, void* stream) {
    const int kt = kt_of(k);
    if (kt < 0 || !ws) return 1;
    const int kp = 32 * kt;
    const size_t slab_bytes = (size_t)kp * HALS_MAX_WG * sizeof(unsigned long long);
    const size_t t_off = align256(slab_bytes + (size_t)kp * sizeof(double));
    if (ws_bytes < t_off + (size_t)m * kp * sizeof(float)) return 1;
    if (nwg) *nwg = (int)cdiv(m, HALS_WG);
    unsigned long long* slab = (unsigned long long*)ws;                 // (unused by the cross-rank sweep: the peers' slabs serve)
    double* ss2 = (double*)((char*)ws + slab_bytes);
    float* T = (float*)((char*)ws + t_off);
    hipStream_t st = S(stream);
    const bool chk = pe == nullptr;
    if (kt == 1) return launch_hals_sweep<1>(W, m, k, ldw, AH, ldah, G, eps, slab, ss2, T, st, pe, chk);
    if (kt == 2) return launch_hals_sweep<2>(W, m, k, ldw, AH, ldah, G, eps, slab, ss2, T, st, pe, chk);
    return launch_hals_sweep<4>(W, m, k, ldw, AH, ldah, G, eps, slab, ss2, T, st, pe, chk);
}

__attribute__((visibility("hidden"))) int dnmf_small_timeout_take_(unsigned* out);      // csrc/dnmf_fit.hip
__attribute__((visibility("hidden"))) int dnmf_team_timeout_take_(unsigned* out);       // csrc/dnmf_team.hip

extern "C" {

int dnmf_hals_sweep_status(int* timed_out, void* stream) {
    REQUIRE(timed_out, "hals_sweep_status: null pointer");
    hipStream_t st = S(stream);
    unsigned int v = 0;
    const unsigned int zero = 0;
    if (hipStreamSynchronize(st) != hipSuccess) return fail(DNMF_EHIP, "hals_sweep_status: stream synchronize failed");
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_hals_timeout), sizeof(v), 0, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(DNMF_EHIP, "hals_sweep_status: read failed");
    if (v && hipMemcpyToSymbol(HIP_SYMBOL(g_hals_timeout), &zero, sizeof(zero), 0, hipMemcpyHostToDevice) != hipSuccess)
        return fail(DNMF_EHIP, "hals_sweep_status: clear failed");
    unsigned int v2 = 0;                                   // the persistent small-problem fit (csrc/dnmf_small.h) reports through the same call
    if (dnmf_small_timeout_take_(&v2) != DNMF_OK) return fail(DNMF_EHIP, "hals_sweep_status: read of the small-fit word failed");
    unsigned int v3 = 0;                                   // ... and so does the one-pass MU/Frobenius team kernel (csrc/dnmf_team.h)
    if (dnmf_team_timeout_take_(&v3) != DNMF_OK) return fail(DNMF_EHIP, "hals_sweep_status: read of the team kernel's word failed");
    *timed_out = (v != 0) || (v2 != 0) || (v3 != 0);
    return DNMF_OK;
}

int dnmf_hals_update_h(float* H, int k, long n, long ldh, const float* AtW, long ldatw, const float* G, float eps,
                       void* stream) {
    if (wide_k(k)) {
        REQUIRE(H && AtW && G && n >= 1 && ldh >= n && ldatw >= n, "hals_update_h: bad arguments");
        return dnmf_wide_hals_update_h_(H, k, n, ldh, AtW, ldatw, G, 256, eps, stream);
    }
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && H && AtW && G && n >= 1 && ldh >= n && ldatw >= n, "hals_update_h: bad arguments");
    const dim3 grid((unsigned)cdiv(n, 256)), block(256);
    hipStream_t st = S(stream);
#define HH_CASE(KT_)                                                                                              \
    if (kt == KT_) {                                                                                              \
        const size_t lds = (size_t)(32 * KT_) * (32 * KT_) * sizeof(float);                                        \
        DNMF_LAUNCH((hals_h_kernel<32 * KT_>), grid, block, lds, st, H, k, n, ldh, AtW, ldatw, G, eps);     \
    }
    HH_CASE(1) HH_CASE(2)
#undef HH_CASE
    if (kt == 4) {
        const int kp = 128;
        const size_t lds = (size_t)(kp * kp + kp * 64) * sizeof(float);   // 96 KiB
        static bool once = false;
        if (!once) { allow_lds(hals_h_kernel_lds, lds); once = true; }
        DNMF_LAUNCH(hals_h_kernel_lds, dim3((unsigned)cdiv(n, 64)), dim3(64), lds, st, H, k, n, ldh, AtW, ldatw, G, kp, eps);
    }
    return check_launch("hals_update_h");
}

}  // extern "C"
