// dnmf.hip -- MI355X (gfx950 / CDNA4) kernels + C ABI for the distributed-NMF multiplicative-update path.
//
// Written for gfx950 only: 64-wide wavefronts, fp32-input MFMA (v_mfma_f32_32x32x2_f32, exact fp32 = an
// fmaf chain), 160 KiB LDS per CU, 8 XCDs.  See DESIGN.md for the data layout and per-kernel rooflines and
// include/dnmf.h for the reference lines each entry point replaces.
//
// Two GEMM forms carry the whole path (k = NMF rank, padded to KP = 32*KT):
//   NT form  C[i][j] = sum_c X[i][c] * Y[j][c]   contraction index contiguous in both operands
//            -> A H^T (K2), H H^T (K1), W (H H^T) (K3).  X is streamed through LDS (the MFMA operand
//               layout puts the 32 rows of a tile across lanes, so a transpose is unavoidable).
//   TN form  C[j][c] = sum_i X[i][j] * Y[i][c]   contraction index is the row of both operands
//            -> W^T A (K6), W^T W (K5), (W^T W) H (K7).  Operands go global -> VGPR directly in MFMA
//               layout with 16-byte coalesced loads; no LDS, no barriers.
// MFMA 32x32x2 f32 operand maps (lane l, li = l & 31, h = l >> 5):
//   A-operand: A[i = li][kk = h]   B-operand: B[kk = h][j = li]
//   C/D: col = li, row = (reg & 3) + 8 * (reg >> 2) + 4 * h   (reg in [0,16))
// The contraction order inside a tile and the output row/column order inside a tile are permuted freely
// (sums are order-agnostic up to fp32 rounding; outputs are written to their true addresses).
#include "dnmf_common.h"
#include "dnmf_host.h"
#include "dnmf_nt.h"
#include "dnmf_tn.h"
#include "dnmf_stream.h"
#include "dnmf_update.h"
#include "dnmf_nn.h"
#include "dnmf_k16.h"

// csrc/dnmf_wide.hip: the k x k products of ranks 128 < k <= 256
#define HID __attribute__((visibility("hidden")))
HID int dnmf_wide_mu_update_w_(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, long ldg, float eps, void* stream);
HID int dnmf_wide_mu_update_h_(float* H, int k, long n, long ldh, const float* AtW, long ldatw, const float* G, long ldg, float eps, int clamp,
                               void* stream);
#undef HID

char* dnmf_errbuf_() {
    static thread_local char buf[DNMF_ERRBUF] = "";
    return buf;
}

BatchCtx* dnmf_batch_() {
    static thread_local BatchCtx ctx = {1, {}};
    return &ctx;
}

namespace {

template <int KT, int MT, int NW, int KS, bool FAST, int MODE, int PF, typename TX>
int launch_nt_pf(const NtArgs& a, int nsplit, hipStream_t st) {
    constexpr int BM = 32 * MT * (NW / KS);
    constexpr size_t lds32 = 2ul * (BM + 32 * KT) * BK * sizeof(float);                     // fp32 tiles (also the W.G loop)
    constexpr size_t lds16 = 2ul * (BM * BK + 32 * KT * BKH) * sizeof(float);               // bf16 X: [X raw | Y 64 wide]
    constexpr size_t lds = std::is_same<TX, bf16_t>::value ? (lds16 > lds32 ? lds16 : lds32) : lds32;
    static bool once = false;
    if (!once) { allow_lds(nt_kernel<KT, MT, NW, KS, FAST, MODE, PF, TX>, lds); once = true; }
    DNMF_LAUNCH((nt_kernel<KT, MT, NW, KS, FAST, MODE, PF, TX>), dim3((unsigned)cdiv(a.nrows, BM), (unsigned)nsplit),
                       dim3(64 * NW), lds, st, a);
    return check_launch("nt_kernel");
}

template <int KT, int MT, int NW, int KS, bool FAST, int MODE, typename TX>
int launch_nt_inst(const NtArgs& a, int nsplit, hipStream_t st) {
    // PF code (switch DNMF_NT_PF for A/B runs) -- what the interior tiles of the streamed operand do:
    //   1  loads of k-tile t+1 issued at the top of tile t, tiles in order (also the generic / edge path)
    //   5  1 + every workgroup starts at a different k-tile (rotated order) + nontemporal loads of A
    //   7  5 with LDS-DMA staging (global_load_lds: no staging VGPRs, no ds_write)
    //  10  5 with TWO k-tiles in flight and a branch-free loop (nt_mainloop_p2) -- the default; grids of at most one
    //      workgroup per CU take three tiles in flight (nt_mainloop_p3t, code 13)
    // Measured on MI355X (tools/kbench.py, k = 64, n = 8192; ms at 262144 / 65536 / 32768 rows):
    //   1: 2.54 / 0.82 / -     5: 2.42-2.49 / 0.76 / 0.44     7: +-2 % of 5     10: 2.43-2.47 / 0.76 / 0.40
    // HBM reads per launch at 262144 rows (PMC): 8.67 GB for 1, 10.99 GB with the rotation alone (the streamed A evicts
    // H from L2), 8.47 GB with the nontemporal hint on top.  What did NOT help (measured, not kept): a prefetch
    // distance of 2 with conditional loads (hipcc then drains vmcnt(0) at every join: 0-18 % slower), fragment reads
    // one group ahead in a second register set (-0..4 %), 8-wave workgroups with two contraction slices (+-3 %),
    // 64-row tiles, capping workgroups per CU through the LDS request (+-5 %), loads issued for PAIRS of adjacent tiles
    // (256 B per row per visit, two register super-sets: the held clock rose 1.92 -> 1.99 GHz -- fewer DRAM row
    // activations -- but 208 registers leave two workgroups per CU, MFMA busy 85 -> 81 %, net +-2 %), and an NT form
    // WITHOUT LDS in the style of the TN kernel (lane (li, h) loads 16 B of its own row of X and of the factor, four
    // consecutive steps consume each row's 128-B line from L1): correct, but a load instruction then touches 32 rows x
    // 32 B and the texture-address unit, not the matrix pipe, paces the kernel -- 2.6x SLOWER at k = 32 and k = 64 (MFMA
    // busy 21-31 %, HBM reads 1.8x algorithmic).  The transpose of X has to go through LDS.  With A cache-resident the same
    // kernel reaches 103 / 118 / 128 TFLOP/s at 32768 / 65536 / 262144 rows against 82-89 / 91 / 113 from HBM -- with
    // IDENTICAL cycle counts (PMC): the difference is the core clock the chip holds (1.8-2.0 GHz with the HBM stream,
    // 2.2-2.4 GHz without), i.e. power, not the instruction schedule (measured ceilings).
    static const int pf = (int)tune("DNMF_NT_PF", 10);
    if constexpr (FAST && KS == 1) if (MODE == NT_FUSED_W || !a.store_all) {
#ifdef DNMF_TUNING      // the A/B variants are only instantiated in the tuning build
        if (pf == 5) return launch_nt_pf<KT, MT, NW, KS, FAST, MODE, 5, TX>(a, nsplit, st);
        if constexpr (std::is_same<TX, float>::value)
            if (pf == 7) return launch_nt_pf<KT, MT, NW, KS, FAST, MODE, 7, TX>(a, nsplit, st);
#endif
        if constexpr (std::is_same<TX, float>::value) {
            // 13 = 10 with three tiles in flight: chosen when the grid has at most one workgroup per CU (a 32768-row
            // shard = the per-GPU work of the 8-GPU configuration), where a single wave per SIMD has to cover the HBM latency
            constexpr int BM = 32 * MT * (NW / KS);
            if (pf == 13 || (pf == 10 && cdiv(a.nrows, BM) * nsplit <= 256))
                return launch_nt_pf<KT, MT, NW, KS, FAST, MODE, 13, TX>(a, nsplit, st);
            if (pf == 10) return launch_nt_pf<KT, MT, NW, KS, FAST, MODE, 10, TX>(a, nsplit, st);
        } else {
            if (pf == 10) return launch_nt_pf<KT, MT, NW, KS, FAST, MODE, 5, TX>(a, nsplit, st);   // bf16 X: one 64-wide tile in flight
        }
    }
    return launch_nt_pf<KT, MT, NW, KS, FAST, MODE, 1, TX>(a, nsplit, st);
}

// Tile configuration per padded rank (KT = KP/32).  The kernel template also supports 8-wave workgroups with two contraction slices per row group (KS = 2: same
// tile, twice the waves per SIMD); measured within +-3 % of the 4-wave form at every shard size and not instantiated.
template <int MODE, typename TX = float>
int launch_nt(int kt, bool fast, const NtArgs& a, int nsplit, hipStream_t st) {
#define NT_CASE(KT_, MT_, NW_, KS_)                                                               \
    return fast ? launch_nt_inst<KT_, MT_, NW_, KS_, true, MODE, TX>(a, nsplit, st)                \
                : launch_nt_inst<KT_, MT_, NW_, KS_, false, MODE, TX>(a, nsplit, st);
    // 128-row tiles for every rank.  (k <= 32 used 256-row tiles, MT = 2, in an earlier version: equal at 262144 rows,
    // 1.8x slower at 32768 rows where it left half the CUs without a workgroup.)
#ifdef DNMF_TUNING
    if (kt == 2 && tune("DNMF_NT_MT", 1) == 2) { NT_CASE(2, 2, 4, 1) }     // 256-row tiles at k = 64 (A/B runs)
    // 32-row workgroups (4 waves = 4 contraction slices of one row group): fills the GPU on an 8192-row slab, for the
    // slab-wise one-pass experiment of tools/onepass.py
    if constexpr (std::is_same<TX, float>::value)
        if (kt == 1 && tune("DNMF_NT_KS4", 0) != 0) { NT_CASE(1, 1, 4, 4) }
#endif
    if (kt == 1) { NT_CASE(1, 1, 4, 1) }
    if (kt == 2) { NT_CASE(2, 1, 4, 1) }
    if (kt == 4) { NT_CASE(4, 1, 4, 1) }
#undef NT_CASE
    return fail(DNMF_EINVAL, "unsupported k tile %d", kt);
}

inline int nt_rows_per_tile(int) { return 128; }

template <int MODE, typename TY = float>
int launch_tn(int kt, bool fast, const TnArgs& a, hipStream_t st) {
    const long waves = (long)a.nchunks * a.ncolblk;
    const dim3 grid((unsigned)cdiv(waves, 4)), block(256);
    // nontemporal loads of the streamed operand (A): +2 % at 262144 rows, +7 % at 65536 rows, slightly less HBM
    // traffic (the reused W rows stay in L2).  DNMF_TN_NT=0 switches them off for A/B runs.
    static const bool nty = tune("DNMF_TN_NT", 1) != 0;
#define TN_CASE(KT_, NT_)                                                                                \
    if (kt == KT_) {                                                                                     \
        if (fast && nty) DNMF_LAUNCH((tn_kernel<KT_, NT_, true, MODE, true, TY>), grid, block, 0, st, a); \
        else if (fast) DNMF_LAUNCH((tn_kernel<KT_, NT_, true, MODE, false, TY>), grid, block, 0, st, a);  \
        else DNMF_LAUNCH((tn_kernel<KT_, NT_, false, MODE, false, TY>), grid, block, 0, st, a);          \
        return check_launch("tn_kernel");                                                                \
    }
    TN_CASE(1, 4)
    TN_CASE(2, 4)
    TN_CASE(4, 2)
#undef TN_CASE
    return fail(DNMF_EINVAL, "unsupported k tile %d", kt);
}

// W^T W streams only W (m x k): with 256-row chunks a 32768-row shard gives 128 waves on 32 CUs and the launch is one long
// latency chain; 128-row chunks spread it over the chip (28.8 -> 21.5 us at 32768 rows, 30.2 -> 22.2 us at 65536, k = 64)
// but cost a second reduction stage on tall matrices (34 -> 42 us at 262144 rows), which keep 256.
inline long gram_min_rows(long m) { return m <= 131072 ? 128 : 256; }

struct SplitPlan { int nsplit; long cols_per_split; };

SplitPlan plan_gram_nt(long n) {
    SplitPlan s;
    // each workgroup walks its k-tiles as one latency chain: 128 columns (4 tiles) while that still gives at most 64
    // partial tiles (one reduction stage) -- 14.0 -> 8.8 us at n = 8192 -- else 256
    s.cols_per_split = n <= 64 * 128 ? 128 : 256;
    if (n > 256 * 512) s.cols_per_split = round_up(cdiv(n, 512), BK);
    s.nsplit = (int)std::max<long>(1, cdiv(n, s.cols_per_split));
    return s;
}


// element-wise pass (dnmf_stream.h): 16-byte vectors when X (and S) allow it; a long-row or a patch launch
template <int OP>
int launch_ew(float* X, long rows, long cols, long ldx, const float* Sm, long lds_, const float* x, float eps, int clamp,
              const char* what, hipStream_t st) {
    const bool vec = aligned16(X) && ldx % 4 == 0 && cols % 4 == 0 && (!Sm || (aligned16(Sm) && lds_ % 4 == 0)) &&
                     (!x || OP == EW_ROWS_MUL || OP == EW_KL_BYROW || aligned16(x));
    const long cvecs = vec ? cols / 4 : cols;
    constexpr int U = 4;
    if (cvecs > 256 && rows <= 65535) {                       // long rows
        const dim3 grid((unsigned)cdiv(cvecs, 256 * U), (unsigned)rows);
        // a matrix that cannot stay in the caches until it is touched again (>= 256 MiB) is streamed with nontemporal
        // loads and stores: 5.5 -> 6.1 TB/s on the 1 GiB operands of the isolation pass; smaller factors (the W and H of a
        // step are re-read by the next kernel) keep the default policy.  DNMF_EW_NT = 0 / 1 forces it in the tuning build.
        const long ntp_dflt = (double)rows * cols * sizeof(float) >= 256.0 * (1 << 20);
        if (vec && tune("DNMF_EW_NT", ntp_dflt)) {
            DNMF_LAUNCH((ew_kernel<OP, 4, true, true>), grid, dim3(256), 0, st, X, rows, cols, ldx, Sm, lds_, x, eps, clamp, 0);
            return check_launch(what);
        }
        if (vec) DNMF_LAUNCH((ew_kernel<OP, 4, true>), grid, dim3(256), 0, st, X, rows, cols, ldx, Sm, lds_, x, eps, clamp, 0);
        else DNMF_LAUNCH((ew_kernel<OP, 1, true>), grid, dim3(256), 0, st, X, rows, cols, ldx, Sm, lds_, x, eps, clamp, 0);
        return check_launch(what);
    }
    if (cvecs > 256) return fail(DNMF_EINVAL, "%s: %ld rows x %ld columns: neither a long-row nor a patch shape", what, rows, cols);
    int txs = 0;
    while (txs < 8 && (1L << txs) < cvecs) ++txs;
    const long TY = 256 >> txs;
    const dim3 grid((unsigned)cdiv(rows, TY * U));
    if (vec && tune("DNMF_EW_NT", (double)rows * cols * sizeof(float) >= 256.0 * (1 << 20))) {
        DNMF_LAUNCH((ew_kernel<OP, 4, false, true>), grid, dim3(256), 0, st, X, rows, cols, ldx, Sm, lds_, x, eps, clamp, txs);
        return check_launch(what);
    }
    if (vec) DNMF_LAUNCH((ew_kernel<OP, 4, false>), grid, dim3(256), 0, st, X, rows, cols, ldx, Sm, lds_, x, eps, clamp, txs);
    else DNMF_LAUNCH((ew_kernel<OP, 1, false>), grid, dim3(256), 0, st, X, rows, cols, ldx, Sm, lds_, x, eps, clamp, txs);
    return check_launch(what);
}

}  // namespace
// csrc/dnmf_team.hip: the one-pass MU/Frobenius step (16 < k <= 32, n a multiple of 512 up to 4096)
__attribute__((visibility("hidden"))) size_t dnmf_team_ws_bytes_(long m, long n, int k);
__attribute__((visibility("hidden"))) int dnmf_team_fro_(const float* A, long m, long n, long lda, float* W, long ldw, const float* H, long ldh,
                                                          const float* G, int k, float eps, void* part, size_t part_bytes, void* stream,
                                                          const float** P_out, int* nparts, const float** Pg_out, int* kp_out);
namespace {

struct WsLayout {
    size_t g_off, s_off, x_off, part_off, total;  // G [KP*KP] | S = AtW / AH / UHT / WTU | x [KP] | partials
};

size_t partial_bytes(long m, long n, int k) {
    if (wide_k(k))      // the panel calls of the contractions and of the Gram blocks + the m x n quotient image of the KL products
        return align256(std::max(std::max(partial_bytes(m, n, WIDE_PANEL), (size_t)256 * HALS_MAX_WG * sizeof(unsigned long long) + 256 * sizeof(double)),
                                 std::max(partial_bytes(m, WIDE_PANEL, WIDE_PANEL), partial_bytes(WIDE_PANEL, n, WIDE_PANEL)))) +
               align256((size_t)m * round_up(n, 4) * sizeof(float));
    const int kt = kt_of(k), kp = 32 * kt;
    size_t b = 0;
    if (k <= 16 && n % 64 == 0) {   // tn16 partial slabs (fp32: V = 4; bf16: V = 8 has fewer column blocks, more chunks)
        for (int v = 4; v <= 8; v += 4) {
            if (n % (16 * v)) continue;
            Tn16Plan q = plan_tn16(m, n, v);
            b = std::max(b, (size_t)q.nchunks * 16 * n * sizeof(float) + reduce_scratch_bytes(q.nchunks, k, n) + GRAM_RIDE_BYTES);
        }
        const Tn16Plan w = plan_wtu16(m, n);                        // kl_wtu16 partial slabs
        b = std::max(b, (size_t)w.nchunks * 16 * n * sizeof(float) + reduce_scratch_bytes(w.nchunks, k, n));
    }
    {   // wta / kl_wtu: A [m x n]
        TnPlan p = plan_tn(m, n, kt, tn_nt(kt));
        b = std::max(b, (size_t)p.nchunks * p.chunk_stride * sizeof(float) + reduce_scratch_bytes(p.nchunks, k, n));
        const KlWtuPlan q = plan_kl_wtu(m, n, kt);
        b = std::max(b, (size_t)q.nchunks * q.tn.ldp * kp * sizeof(float) + reduce_scratch_bytes((int)q.nchunks, k, n));
    }
    {   // gram W^T W: Y = W [m x k]
        TnPlan p = plan_tn(m, kp, kt, kt == 4 ? 2 : kt, gram_min_rows(m));
        b = std::max(b, (size_t)p.nchunks * p.chunk_stride * sizeof(float) + reduce_scratch_bytes(p.nchunks, kp, kp));
    }
    {   // gram H H^T
        SplitPlan s = plan_gram_nt(n);
        b = std::max(b, (size_t)s.nsplit * nt_rows_per_tile(kt) * kp * sizeof(float) + reduce_scratch_bytes(s.nsplit, kp, kp));
    }
    {   // kl_uht column-split slabs
        const long rowtiles = cdiv(m, 128);
        const int nsp = plan_uht(m, n, kt).nsplit;
        if (nsp > 1) b = std::max(b, (size_t)nsp * m * kp * sizeof(float) + reduce_scratch_bytes(nsp, (int)m, k));
        if (k <= 16) {   // kl_uht16: 16-wide slabs, also for a single split (ranks below 16 go through the slab)
            const long ns16 = std::min<long>(std::max<long>(1, 1024 / rowtiles), std::max<long>(1, n / 256));
            const int nsp16 = (int)cdiv(n, round_up(cdiv(n, ns16), BK));
            b = std::max(b, (size_t)nsp16 * m * 16 * sizeof(float) + reduce_scratch_bytes(nsp16, (int)m, k));
        }
    }
    b = std::max(b, (size_t)1024 * kp * sizeof(float));  // colsum partials (at most 1024 slabs)
    b = std::max(b, dnmf_team_ws_bytes_(m, n, k));       // one-pass MU/FRO step: the teams' partials + the granule ring (csrc/dnmf_team.hip)
    // HALS W sweep: slots + norms + the m x KP block T of its first pass (only when the sweep is asked for a factor of
    // this shape, i.e. n == k: dnmf_hals_sweep_w is called with dnmf_ws_bytes(m, k, k))
    b = std::max(b, align256((size_t)kp * HALS_MAX_WG * sizeof(unsigned long long) + (size_t)kp * sizeof(double)) +
                        (n == k ? (size_t)m * kp * sizeof(float) : 0));
    return b;
}

WsLayout ws_layout(long m, long n, int k) {
    const int kp = kp_of(k);
    WsLayout w;
    w.g_off = 0;
    w.s_off = align256((size_t)kp * kp * sizeof(float));
    const size_t s_elems = std::max((size_t)k * round_up(n, 4), (size_t)m * round_up(k, 4));
    w.x_off = w.s_off + align256(s_elems * sizeof(float));
    w.part_off = w.x_off + align256((size_t)kp * sizeof(float));
    // + room for the zero-padded factor images of the KL products (pad_factors) at the end of the partial area
    w.total = w.part_off + align256(partial_bytes(m, n, k)) + pad_bytes(m, n, kp);
    return w;
}

}  // namespace

// =============================================================================================== C ABI
// Process-wide: may kernels whose workgroups WAIT FOR EACH OTHER run (the whole fits of small problems, the persistent HALS W sweep
// -- local and across ranks --, the one-pass MU/FRO team kernel)?  They need every workgroup of a launch resident at once, i.e. the GPU
// to themselves; off, every path takes its launch-chain form (same update rules, no residency needed).  dnmf_set_persistent.
static int g_persistent_on = 1;
__attribute__((visibility("hidden"))) int dnmf_persistent_on_() { return g_persistent_on; }
__attribute__((visibility("hidden"))) void dnmf_persistent_set_(int on) { g_persistent_on = on; }
__attribute__((visibility("hidden"))) void dnmf_ws_offsets_(long m, long n, int k, size_t out[5]);
void dnmf_ws_offsets_(long m, long n, int k, size_t out[5]) {   // library-internal (csrc/dnmf_split.hip)
    const WsLayout L = ws_layout(m, n, k);
    out[0] = L.g_off; out[1] = L.s_off; out[2] = L.x_off; out[3] = L.part_off; out[4] = L.total;
}

extern "C" {

const char* dnmf_last_error(void) { return dnmf_errbuf_(); }
int dnmf_set_persistent(int on) {
    const int was = dnmf_persistent_on_();
    dnmf_persistent_set_(on != 0);
    return was;
}
int dnmf_version(void) { return 100; }
int dnmf_kp(int k) { return kp_of(k); }

size_t dnmf_ws_bytes(long m, long n, int k) {
    if (kp_of(k) < 0 || m < 1 || n < 1) return 0;
    return ws_layout(m, n, k).total;
}


}  // extern "C"
namespace {
template <typename TA>
int aht_impl(const TA* A, long m, long n, long lda, const float* H, int k, long ldh, float* AH, long ldah, void* stream, long hblk = 0);
template <typename TA>
int wta_impl(const TA* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw, void* ws, size_t ws_bytes, void* stream,
             float* G = nullptr);
// Gram matrices of a wide rank: the zero-padded 256 x 256 buffer, filled block by block with the tuned contractions
// (H_p H_q^T = "A H^T" with A = H_p; W_p^T W_q = "W^T A" with A = W_q)
int wide_gram(bool hht, const float* F, long len, int k, long ld, float* G, void* ws, size_t ws_bytes, void* stream) {
    if (batch_memset(G, 0, (size_t)256 * 256 * sizeof(float), S(stream)) != hipSuccess) return fail(DNMF_EHIP, "gram: memset failed");
    for (int p = 0; p < 2; ++p)
        for (int q = 0; q < 2; ++q) {
            const int kx = p ? k - WIDE_PANEL : WIDE_PANEL, ky = q ? k - WIDE_PANEL : WIDE_PANEL;
            float* blk = G + (long)p * WIDE_PANEL * 256 + q * WIDE_PANEL;
            int rc;
            if (hht) rc = aht_impl<float>(F + (long)p * WIDE_PANEL * ld, kx, len, ld, F + (long)q * WIDE_PANEL * ld, ky, ld, blk, 256, stream);
            else rc = wta_impl<float>(F + q * WIDE_PANEL, len, ky, ld, F + p * WIDE_PANEL, kx, ld, blk, 256, ws, ws_bytes, stream);
            if (rc) return rc;
        }
    return DNMF_OK;
}
}  // namespace
extern "C" {

int dnmf_gram_hht(const float* H, int k, long n, long ldh, float* G, void* ws, size_t ws_bytes, void* stream) {
    if (wide_k(k)) {
        REQUIRE(H && G && n >= 1 && ldh >= n, "gram_hht: bad arguments (k=%d n=%ld)", k, n);
        return wide_gram(true, H, n, k, ldh, G, ws, ws_bytes, stream);
    }
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && H && G && ws && n >= 1 && ldh >= n, "gram_hht: bad arguments (k=%d n=%ld)", k, n);
    const int kp = 32 * kt;
    const SplitPlan sp = plan_gram_nt(n);
    const long tile_rows = nt_rows_per_tile(kt);
    const size_t pbytes = (size_t)sp.nsplit * tile_rows * kp * sizeof(float);
    const size_t need = pbytes + reduce_scratch_bytes(sp.nsplit, kp, kp);
    if (ws_bytes < need) return fail(DNMF_EWS, "gram_hht: workspace %zu < %zu", ws_bytes, need);
    NtArgs a{};
    a.X = H; a.ldx = ldh; a.nrows = k; a.ncols = n;
    a.Y = H; a.ldy = ldh; a.yrows = k;
    a.cols_per_split = sp.cols_per_split;
    a.out = (float*)ws; a.ldo = kp; a.split_stride = tile_rows * kp; a.store_all = 1;
    const bool fast = aligned16(H) && ldh % 4 == 0 && n % 4 == 0;
    int rc = launch_nt<NT_STORE>(kt, fast, a, sp.nsplit, S(stream));
    if (rc) return rc;
    return launch_reduce((const float*)ws, tile_rows * kp, kp, sp.nsplit, G, kp, k, k, kp, kp,
                         (float*)((char*)ws + pbytes), S(stream));
}

int dnmf_gram_wtw(const float* W, long m, int k, long ldw, float* G, void* ws, size_t ws_bytes, void* stream) {
    if (wide_k(k)) {
        REQUIRE(W && G && ws && m >= 1 && ldw >= k, "gram_wtw: bad arguments (k=%d m=%ld)", k, m);
        return wide_gram(false, W, m, k, ldw, G, ws, ws_bytes, stream);
    }
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && W && G && ws && m >= 1 && ldw >= k, "gram_wtw: bad arguments (k=%d m=%ld)", k, m);
    const int kp = 32 * kt;
    // TN form with X = Y = W; NT column sets per wave (KT = 4 uses 2 column blocks of 64 to bound registers)
    const int nt = kt == 4 ? 2 : kt;
    TnPlan p = plan_tn(m, kp, kt, nt, gram_min_rows(m));
    const size_t pbytes = (size_t)p.nchunks * p.chunk_stride * sizeof(float);
    const size_t need = pbytes + reduce_scratch_bytes(p.nchunks, kp, kp);
    if (ws_bytes < need) return fail(DNMF_EWS, "gram_wtw: workspace %zu < %zu", ws_bytes, need);
    TnArgs a{};
    a.X = W; a.ldx = ldw; a.xcols = k; a.Y = W; a.ldy = ldw; a.ycols = k;
    a.nrows = m; a.rows_per_chunk = p.rows_per_chunk; a.nchunks = p.nchunks; a.ncolblk = p.ncolblk;
    a.P = (float*)ws; a.chunk_stride = p.chunk_stride; a.ldp = p.ldp;
    const bool fast = aligned16(W) && ldw % 4 == 0 && k % 4 == 0;
    const long waves = (long)a.nchunks * a.ncolblk;
    const dim3 grid((unsigned)cdiv(waves, 4)), block(256);
    hipStream_t st = S(stream);
#define GRAM_CASE(KT_, NT_)                                                                          \
    if (kt == KT_) {                                                                                 \
        if (fast) DNMF_LAUNCH((tn_kernel<KT_, NT_, true, TN_PARTIAL>), grid, block, 0, st, a); \
        else DNMF_LAUNCH((tn_kernel<KT_, NT_, false, TN_PARTIAL>), grid, block, 0, st, a);    \
    }
    GRAM_CASE(1, 1) GRAM_CASE(2, 2) GRAM_CASE(4, 2)
#undef GRAM_CASE
    int rc = check_launch("gram_wtw");
    if (rc) return rc;
    return launch_reduce((const float*)ws, p.chunk_stride, p.ldp, p.nchunks, G, kp, k, k, kp, kp,
                         (float*)((char*)ws + pbytes), st);
}

}  // extern "C"  (typed implementations shared by the fp32 and the bf16-A entry points)
namespace {
// alignment the vector path needs from A: 16 B for fp32 rows, 8 B for bf16 rows (4 elements per lane either way)

// rank k <= 16 with aligned operands and a column count that is a whole number of k-tiles: the 16-wide kernel.
// Returns 1 when not applicable (caller falls through to the 32-wide kernels), else the launch status.
template <int MODE, typename TA>
int try_nt16(const NtArgs& a, bool fast, long n, int k, hipStream_t st) {
    constexpr bool b16 = std::is_same<TA, bf16_t>::value;
    if (!(k <= 16 && fast && k16_on() && n % (b16 ? BKH : BK) == 0)) return 1;
    DNMF_LAUNCH((nt16_kernel<TA, MODE>), dim3((unsigned)cdiv(a.nrows, 128)), dim3(256), nt16_lds_bytes(b16), st, a);
    return check_launch("nt16_kernel");
}

// hblk = 0: H is one k x n matrix (ldh).  hblk > 0 (fp32 A): H is the stack of n / hblk column blocks [q][k][hblk], ldh = hblk.
template <typename TA>
int aht_impl(const TA* A, long m, long n, long lda, const float* H, int k, long ldh, float* AH, long ldah,
             void* stream, long hblk) {
    if (wide_k(k)) {                                               // A H^T splits exactly along the rows of H: two passes over A
        REQUIRE(!hblk && H && AH && ldah >= k && ldh >= n, "aht: bad arguments (k = %d; H as column blocks: k <= %d)", k, DNMF_TUNED_MAX_K);
        if (int rc = aht_impl<TA>(A, m, n, lda, H, WIDE_PANEL, ldh, AH, ldah, stream, 0)) return rc;
        return aht_impl<TA>(A, m, n, lda, H + (long)WIDE_PANEL * ldh, k - WIDE_PANEL, ldh, AH + WIDE_PANEL, ldah, stream, 0);
    }
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && H && AH && m >= 1 && n >= 1 && lda >= n && ldah >= k, "aht: bad arguments");
    REQUIRE(hblk ? (ldh == hblk && n % hblk == 0 && hblk % BK == 0 && (n / 16) * (hblk / 16) < (1L << 32)) : ldh >= n,
            "aht: bad H layout (ldh %ld, block %ld, n %ld)", ldh, hblk, n);
    NtArgs a{};
    a.X = A; a.ldx = lda; a.nrows = m; a.ncols = n;
    a.Y = H; a.ldy = ldh; a.yrows = k;
    if (hblk) {
        const unsigned long units = (unsigned long)(hblk / 16);
        a.yb = YBlk{(unsigned)(((1UL << 32) + units - 1) / units), (long)(k - 1) * hblk};
    }
    a.cols_per_split = round_up(n, BK);
    a.out = AH; a.ldo = ldah; a.split_stride = 0; a.store_all = 0;
    const bool fast = a_rows16(A, lda) && aligned16(H) && ldh % 4 == 0 && n % 4 == 0;
    if (int rc16 = try_nt16<NT_STORE, TA>(a, fast, n, k, S(stream)); rc16 != 1) return rc16;
    return launch_nt<NT_STORE, TA>(kt, fast, a, 1, S(stream));
}

template <typename TA>
int aht_update_w_impl(const TA* A, long m, long n, long lda, const float* H, int k, long ldh, const float* G,
                      float* W, long ldw, float eps, void* stream) {
    REQUIRE(!wide_k(k), "aht_update_w: the fused form takes k <= %d (dnmf_aht followed by dnmf_mu_update_w beyond it)", DNMF_TUNED_MAX_K);
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && H && G && W && m >= 1 && n >= 1 && (lda >= n || alias_ok(lda)) && ldh >= n && ldw >= k, "aht_update_w: bad arguments");
    NtArgs a{};
    a.X = A; a.ldx = lda; a.nrows = m; a.ncols = n;
    a.Y = H; a.ldy = ldh; a.yrows = k;
    a.cols_per_split = round_up(n, BK);
    a.W = W; a.ldw = ldw; a.G = G; a.eps = eps; a.k = k;
    const bool fast = a_rows16(A, lda) && aligned16(H) && ldh % 4 == 0 && n % 4 == 0;
    a.wfast = aligned16(W) && ldw % 4 == 0 && k % 4 == 0;
    if (int rc16 = try_nt16<NT_FUSED_W, TA>(a, fast, n, k, S(stream)); rc16 != 1) return rc16;
    return launch_nt<NT_FUSED_W, TA>(kt, fast, a, 1, S(stream));
}
}  // namespace
extern "C" {

int dnmf_aht(const float* A, long m, long n, long lda, const float* H, int k, long ldh, float* AH, long ldah,
             void* stream) {
    return aht_impl<float>(A, m, n, lda, H, k, ldh, AH, ldah, stream);
}
int dnmf_aht_hblocks(const float* A, long m, long n, long lda, const float* Hs, long nh, int k, float* AH, long ldah,
                     void* stream) {
    REQUIRE(nh >= 1, "aht_hblocks: bad block width");
    return aht_impl<float>(A, m, n, lda, Hs, k, nh, AH, ldah, stream, nh);
}
int dnmf_aht_bf16a(const void* A, long m, long n, long lda, const float* H, int k, long ldh, float* AH, long ldah,
                   void* stream) {
    return aht_impl<bf16_t>((const bf16_t*)A, m, n, lda, H, k, ldh, AH, ldah, stream);
}
int dnmf_aht_update_w(const float* A, long m, long n, long lda, const float* H, int k, long ldh, const float* G,
                      float* W, long ldw, float eps, void* stream) {
    return aht_update_w_impl<float>(A, m, n, lda, H, k, ldh, G, W, ldw, eps, stream);
}
int dnmf_aht_update_w_bf16a(const void* A, long m, long n, long lda, const float* H, int k, long ldh, const float* G,
                            float* W, long ldw, float eps, void* stream) {
    return aht_update_w_impl<bf16_t>((const bf16_t*)A, m, n, lda, H, k, ldh, G, W, ldw, eps, stream);
}

}  // extern "C"
namespace {
template <int KT, int V, int OCC>
int launch_update_w_seq(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, float eps, hipStream_t st) {
    constexpr size_t lds = (size_t)(32 * KT) * (32 * KT + 4) * sizeof(float);
    static bool once = false;
    if (!once) { allow_lds(update_w_seq_kernel<KT, V, OCC, false>, lds); allow_lds(update_w_seq_kernel<KT, V, OCC, true>, lds); once = true; }
    // one tile per wave (the kernel's tile loop only matters beyond 2^31 workgroups): measured on the 3.2 GB pass at
    // k = 64, workgroups that loop over tiles (256 x OCC of them, G staged once each) 4.5 TB/s, one tile per wave 4.75
    const unsigned grid = upd_grid(cdiv(m, 32), KT);
    constexpr unsigned T = 64 * upd_waves(KT);
    if (k == 32 * KT && m % 32 == 0)
        DNMF_LAUNCH((update_w_seq_kernel<KT, V, OCC, false>), dim3(grid), dim3(T), lds, st, W, m, k, ldw, AH, ldah, G, eps, (float*)nullptr, 0L);
    else
        DNMF_LAUNCH((update_w_seq_kernel<KT, V, OCC, true>), dim3(grid), dim3(T), lds, st, W, m, k, ldw, AH, ldah, G, eps, (float*)nullptr, 0L);
    return check_launch("mu_update_w");
}
}  // namespace
extern "C" {

int dnmf_mu_update_w(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, float eps,
                     void* stream) {
    if (wide_k(k)) {
        REQUIRE(W && AH && G && m >= 1 && ldw >= k && ldah >= k, "mu_update_w: bad arguments");
        return dnmf_wide_mu_update_w_(W, m, k, ldw, AH, ldah, G, 256, eps, stream);
    }
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && W && AH && G && m >= 1 && ldw >= k && ldah >= k, "mu_update_w: bad arguments");
    REQUIRE(ldw < (1L << 23) && ldah < (1L << 23), "mu_update_w: leading dimension beyond the 32-bit tile offsets");
    const bool fast = aligned16(W) && aligned16(AH) && ldw % 4 == 0 && ldah % 4 == 0 && k % 4 == 0;
    hipStream_t st = S(stream);
    {   // 16-row wave tiles (update_w16_kernel; round 5): full ranks, whole tiles, aligned rows, 2 GiB descriptor windows.  Measured on
        // the isolation pass (k x 2^22, tools/dbg/w16_ab.sh): k = 32: 0.346 -> 0.310 ms (0.58 -> 0.65 of HBM), k = 64: 0.759 -> 0.632
        // (0.53 -> 0.64), k = 128: 1.49 -> 1.40 (0.54 -> 0.58: the matrix work of 2 n k^2 flops bounds that one).  Workgroups walk
        // their tiles grid-stride; capped at 1024 workgroups from k = 64 on (G staged once per several tiles: 0.648 -> 0.632 at
        // k = 64), one tile per wave at k = 32.  DNMF_UPD_W16 = 0 (tuning build): the 32-row kernel; > 1: that cap.
        static const long w16 = tune("DNMF_UPD_W16", 1);
        if (w16 && fast && k == 32 * kt && m % 16 == 0 && 16 * ldw * 4 + k * 4 < 0x7fffffffL && 16 * ldah * 4 + k * 4 < 0x7fffffffL) {
            const long tiles = m / 16;
            auto run = [&](auto kt_c, auto nwv_c, auto occ_c) {
                constexpr int KT_ = decltype(kt_c)::value, NWV_ = decltype(nwv_c)::value, OCC_ = decltype(occ_c)::value;
                constexpr size_t lds = (size_t)(32 * KT_) * (32 * KT_ + 4) * sizeof(float);
                static bool once = false;
                if (!once) { allow_lds(update_w16_kernel<KT_, NWV_, OCC_>, lds); once = true; }
                const long cap = w16 > 1 ? w16 : (KT_ == 1 ? (1L << 30) : 1024L);
                const unsigned grid = (unsigned)std::min<long>(cdiv(tiles, NWV_), cap);
                DNMF_LAUNCH((update_w16_kernel<KT_, NWV_, OCC_>), dim3(grid), dim3(64 * NWV_), lds, st, W, m, ldw, AH, ldah, G, eps);
                return check_launch("mu_update_w(16)");
            };
            using std::integral_constant;
            if (kt == 1) return run(integral_constant<int, 1>{}, integral_constant<int, 4>{}, integral_constant<int, 8>{});
            if (kt == 2) return run(integral_constant<int, 2>{}, integral_constant<int, 4>{}, integral_constant<int, 8>{});
            return run(integral_constant<int, 4>{}, integral_constant<int, 8>{}, integral_constant<int, 4>{});
        }
    }
    // waves per SIMD requested from the compiler (3.2 GB pass, k = 64: 3 / 4 / 5 / 6 -> 4.68 / 4.72 / 4.76 / 4.67 TB/s;
    // k = 128 holds 120 registers: 4)
    static const int var0 = (int)tune("DNMF_UPD_W", 0);
    const int var = var0 ? var0 : (kt == 4 ? 4 : 5);
#define UWS(KT_, OCC_)                                                                                            \
    if (kt == KT_ && var == OCC_)                                                                                 \
        return fast ? launch_update_w_seq<KT_, 4, OCC_>(W, m, k, ldw, AH, ldah, G, eps, st)                       \
                    : launch_update_w_seq<KT_, 1, OCC_>(W, m, k, ldw, AH, ldah, G, eps, st);
    UWS(1, 5) UWS(2, 5) UWS(4, 4)
#ifdef DNMF_TUNING
    if (var == 35 && kt == 2 && fast && k == 64 && m % 32 == 0) {      // nontemporal loads and stores (A/B)
        constexpr size_t lds = 64 * 68 * sizeof(float);
        static bool once = false;
        if (!once) { allow_lds(update_w_seq_kernel<2, 4, 5, false, UW_MU, 2>, lds); once = true; }
        DNMF_LAUNCH((update_w_seq_kernel<2, 4, 5, false, UW_MU, 2>), dim3((unsigned)cdiv(cdiv(m, 32), 4)), dim3(256), lds, st, W, m, k, ldw, AH, ldah, G, eps, (float*)nullptr, 0L);
        return check_launch("mu_update_w(nt)");
    }
    UWS(1, 3) UWS(2, 3) UWS(4, 3) UWS(1, 4) UWS(2, 4) UWS(4, 5) UWS(1, 6) UWS(2, 6) UWS(1, 8) UWS(2, 2) UWS(4, 2)
#endif
#undef UWS
    return fail(DNMF_EINVAL, "mu_update_w: no kernel for k tile %d / variant %d", kt, var);
}

}  // extern "C"
namespace {
// G != nullptr: also G = W^T W (KP x KP, zero padded; dist_nmf.py:705) -- for k <= 16 inside the same two launches (the wave of
// column block 0 of every row chunk accumulates it, the reduction launch sums the partial tiles), otherwise dnmf_gram_wtw first.
template <typename TA>
int wta_impl(const TA* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw,
             void* ws, size_t ws_bytes, void* stream, float* G) {
    if (wide_k(k)) {                                               // W^T A splits exactly along the columns of W: two passes over A
        REQUIRE(W && AtW && ws && ldw >= k && ldatw >= n, "wta: bad arguments (k = %d)", k);
        if (G) if (int rc = dnmf_gram_wtw(W, m, k, ldw, G, ws, ws_bytes, stream)) return rc;
        if (int rc = wta_impl<TA>(A, m, n, lda, W, WIDE_PANEL, ldw, AtW, ldatw, ws, ws_bytes, stream, nullptr)) return rc;
        return wta_impl<TA>(A, m, n, lda, W + WIDE_PANEL, k - WIDE_PANEL, ldw, AtW + (long)WIDE_PANEL * ldatw, ldatw, ws, ws_bytes, stream, nullptr);
    }
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && AtW && ws && m >= 1 && n >= 1 && (lda >= n || alias_ok(lda)) && ldw >= k && ldatw >= n, "wta: bad arguments");
    const int kp = 32 * kt;
    static const bool gram_ride = tune("DNMF_WTA_GRAM", 1) != 0;
    auto gram_first = [&]() { return G ? dnmf_gram_wtw(W, m, k, ldw, G, ws, ws_bytes, stream) : DNMF_OK; };
    {   // rank k <= 16: 16-wide kernel (16-byte aligned rows of A, whole column blocks, workspace permitting)
        constexpr int V = std::is_same<TA, bf16_t>::value ? 8 : 4;
        if (k <= 16 && k16_on() && a_rows16(A, lda) && n % (16 * V) == 0 && lda != 0) {
            const Tn16Plan q = plan_tn16(m, n, V);
            const size_t pb = (size_t)q.nchunks * 16 * n * sizeof(float);
            if (pb + reduce_scratch_bytes(q.nchunks, k, n) <= ws_bytes) {
                TnArgs a{};
                a.X = W; a.ldx = ldw; a.xcols = k; a.Y = A; a.ldy = lda; a.ycols = n;
                a.nrows = m; a.rows_per_chunk = q.rows_per_chunk; a.nchunks = q.nchunks; a.ncolblk = q.ncolblk;
                a.P = (float*)ws; a.chunk_stride = 16 * n; a.ldp = n;
                const size_t gb = (size_t)q.nchunks * 256 * sizeof(float);            // partial Gram tiles behind the slabs
                const bool ride = G && gram_ride && reduce_slices(q.nchunks) == 1 && pb + gb <= ws_bytes;
                int rc;
                if (!ride && (rc = gram_first())) return rc;
                hipStream_t st = S(stream);
                const dim3 grid16((unsigned)cdiv((long)q.nchunks * q.ncolblk, 4));
                if (ride) {
                    a.Pg = (float*)((char*)ws + pb);
                    DNMF_LAUNCH((tn16_kernel<TA, true>), grid16, dim3(256), 0, st, a);
                } else {
                    DNMF_LAUNCH((tn16_kernel<TA>), grid16, dim3(256), 0, st, a);
                }
                if ((rc = check_launch("tn16_kernel"))) return rc;
                const GramTail gt{a.Pg, G, 16, k, kp, q.nchunks};
                return launch_reduce((const float*)ws, 16 * n, n, q.nchunks, AtW, ldatw, k, n, k, n, (float*)((char*)ws + pb), st,
                                     ride ? &gt : nullptr);
            }
        }
    }
    const int nt = tn_nt(kt);
    TnPlan p = plan_tn(m, n, kt, nt, 256, wta_waves(m, n, kt, sizeof(TA)));
    const size_t pbytes = (size_t)p.nchunks * p.chunk_stride * sizeof(float);
    const size_t need = pbytes + reduce_scratch_bytes(p.nchunks, k, n);
    if (ws_bytes < need) return fail(DNMF_EWS, "wta: workspace %zu < %zu", ws_bytes, need);
    TnArgs a{};
    a.X = W; a.ldx = ldw; a.xcols = k; a.Y = A; a.ldy = lda; a.ycols = n;
    a.nrows = m; a.rows_per_chunk = p.rows_per_chunk; a.nchunks = p.nchunks; a.ncolblk = p.ncolblk;
    a.P = (float*)ws; a.chunk_stride = p.chunk_stride; a.ldp = p.ldp;
    // the streamed A is read 4 elements per lane, W only KT per lane (1 float for k <= 32): W's alignment need is KT-wide
    const bool fast = a_aligned(A) && lda % 4 == 0 && n % 4 == 0 &&
                      ((uintptr_t)W % (4 * kt)) == 0 && ldw % kt == 0 && k % kt == 0;
    // (the riding Gram was also built for the 32-wide kernel at 16 < k <= 32: no gain -- 0.3707 vs 0.3706 ms per iteration at
    // 65536 x 4096, k = 32, where the two launches it saves cost 6 us -- so these ranks keep dnmf_gram_wtw)
    int rc;
    if ((rc = gram_first())) return rc;
    if ((rc = launch_tn<TN_PARTIAL, TA>(kt, fast, a, S(stream)))) return rc;
    return launch_reduce((const float*)ws, p.chunk_stride, p.ldp, p.nchunks, AtW, ldatw, k, n, k, n,
                         (float*)((char*)ws + pbytes), S(stream));
}
}  // namespace
extern "C" {

int dnmf_wta(const float* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw,
             void* ws, size_t ws_bytes, void* stream) {
    return wta_impl<float>(A, m, n, lda, W, k, ldw, AtW, ldatw, ws, ws_bytes, stream);
}
int dnmf_wta_bf16a(const void* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw,
                   void* ws, size_t ws_bytes, void* stream) {
    return wta_impl<bf16_t>((const bf16_t*)A, m, n, lda, W, k, ldw, AtW, ldatw, ws, ws_bytes, stream);
}

int dnmf_wta_gram(const float* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw, float* G,
                  void* ws, size_t ws_bytes, void* stream) {
    REQUIRE(G, "wta_gram: null Gram buffer");
    return wta_impl<float>(A, m, n, lda, W, k, ldw, AtW, ldatw, ws, ws_bytes, stream, G);
}
int dnmf_wta_gram_bf16a(const void* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw,
                        float* G, void* ws, size_t ws_bytes, void* stream) {
    REQUIRE(G, "wta_gram: null Gram buffer");
    return wta_impl<bf16_t>((const bf16_t*)A, m, n, lda, W, k, ldw, AtW, ldatw, ws, ws_bytes, stream, G);
}

}  // extern "C"
namespace {
template <int KT, int NT, int OCC>
int launch_update_h_seq(float* H, int k, long n, long ldh, const float* AtW, long ldatw, const float* G, float eps, int clamp,
                        hipStream_t st) {
    constexpr size_t lds = (size_t)(32 * KT) * (32 * KT + 4) * sizeof(float);
    static bool once = false;
    if (!once) { allow_lds(update_h_seq_kernel<KT, NT, OCC, false>, lds); allow_lds(update_h_seq_kernel<KT, NT, OCC, true>, lds); once = true; }
    const unsigned grid = upd_grid(cdiv(n, 32 * NT), KT);
    constexpr unsigned T = 64 * upd_waves(KT);
    if constexpr (NT == 2) {
        // a long H that cannot stay cached (>= 64 MiB) streams with nontemporal loads and stores: 4.87 -> 5.07 TB/s at
        // 64 x 2^22 (each line is touched by exactly one instruction here; the W-side kernel touches a line four times
        // and loses a factor of two with the same hint, so it keeps the default policy)
        if (k == 32 * KT && n % (32 * NT) == 0 && (double)k * n * sizeof(float) >= 64.0 * (1 << 20)) {
            static bool once2 = false;
            if (!once2) { allow_lds(update_h_seq_kernel<KT, NT, OCC, false, true, 2, 2>, lds); once2 = true; }
            DNMF_LAUNCH((update_h_seq_kernel<KT, NT, OCC, false, true, 2, 2>), dim3(grid), dim3(T), lds, st, H, k, n, ldh, AtW, ldatw, G, eps, clamp);
            return check_launch("mu_update_h");
        }
    }
    if (k == 32 * KT && n % (32 * NT) == 0)
        DNMF_LAUNCH((update_h_seq_kernel<KT, NT, OCC, false>), dim3(grid), dim3(T), lds, st, H, k, n, ldh, AtW, ldatw, G, eps, clamp);
    else
        DNMF_LAUNCH((update_h_seq_kernel<KT, NT, OCC, true>), dim3(grid), dim3(T), lds, st, H, k, n, ldh, AtW, ldatw, G, eps, clamp);
    return check_launch("mu_update_h");
}
}  // namespace
extern "C" {

int dnmf_mu_update_h(float* H, int k, long n, long ldh, const float* AtW, long ldatw, const float* G, float eps,
                     int clamp, void* stream) {
    if (wide_k(k)) {
        REQUIRE(H && AtW && G && n >= 1 && ldh >= n && ldatw >= n, "mu_update_h: bad arguments");
        return dnmf_wide_mu_update_h_(H, k, n, ldh, AtW, ldatw, G, 256, eps, clamp, stream);
    }
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && H && AtW && G && n >= 1 && ldh >= n && ldatw >= n, "mu_update_h: bad arguments");
    // 32-bit tile offsets of the buffer accesses: (27 + 4) rows of a 32-row block plus the column part stay below 2 GiB
    REQUIRE(ldh < (1L << 24) && ldatw < (1L << 24), "mu_update_h: leading dimension beyond the 32-bit tile offsets");
    const bool even = ((uintptr_t)H % 8 == 0) && ((uintptr_t)AtW % 8 == 0) && ldh % 2 == 0 && ldatw % 2 == 0 && n % 2 == 0;
    hipStream_t st = S(stream);
    // variant code = 10 * NT (columns per lane) + waves per SIMD requested from the compiler.  Measured on the 3.2 GB pass
    // (k x 2^22): k = 64: 14 / 16 -> 4.76 / 4.78 TB/s, 23 -> 4.97, 24 (spills) -> 4.0; k = 32: 14 -> 4.93, 23 -> 5.26;
    // k = 128: 14 -> 3.44, 23 -> 2.58.  Two columns per lane (256 B per row and wave) stream better but need 8-byte
    // aligned rows, and a short H (fewer than 1024 such tiles: a latency chain on a few CUs) keeps 32-column tiles.
    static const int var0 = (int)tune("DNMF_UPD_H", 0);
    // k = 128 (round 4): LDS holds two workgroups per CU there whatever the registers allow (66 KiB of G each), so the two-column
    // tile at TWO waves per SIMD (200 registers, no spills) costs no occupancy and halves the accesses per byte: 4.25 -> 4.5 TB/s on
    // the 6.4 GB pass (variant 23 spills there: 2.4 TB/s)
    int var = var0 ? var0 : ((even && k == 32 * kt && n % 64 == 0 && n / 64 >= 1024) ? (kt == 4 ? 22 : 23) : 14);
    if (var >= 20 && !even) var = 14;
#define UHS(KT_, NT_, OCC_)                                                                                       \
    if (kt == KT_ && var == 10 * NT_ + OCC_)                                                                      \
        return launch_update_h_seq<KT_, NT_, OCC_>(H, k, n, ldh, AtW, ldatw, G, eps, clamp, st);
    UHS(1, 1, 4) UHS(2, 1, 4) UHS(4, 1, 4) UHS(1, 2, 3) UHS(2, 2, 3) UHS(4, 2, 2)
#ifdef DNMF_TUNING
    if (var >= 91 && var <= 93 && kt == 2 && k == 64 && n % 64 == 0 && even) {   // cache-policy variants of the k = 64, NT = 2 kernel
        constexpr size_t lds = 64 * 68 * sizeof(float);
        const unsigned grid = (unsigned)cdiv(cdiv(n, 64), 4);
        if (var == 91) DNMF_LAUNCH((update_h_seq_kernel<2, 2, 3, false, true, 2, 0>), dim3(grid), dim3(256), lds, st, H, k, n, ldh, AtW, ldatw, G, eps, clamp);
        if (var == 92) DNMF_LAUNCH((update_h_seq_kernel<2, 2, 3, false, true, 0, 2>), dim3(grid), dim3(256), lds, st, H, k, n, ldh, AtW, ldatw, G, eps, clamp);
        if (var == 93) DNMF_LAUNCH((update_h_seq_kernel<2, 2, 3, false, true, 2, 2>), dim3(grid), dim3(256), lds, st, H, k, n, ldh, AtW, ldatw, G, eps, clamp);
        return check_launch("mu_update_h(aux)");
    }
    if (var == 99 && kt == 2 && k == 64 && n % 32 == 0) {   // memory pattern of the k = 64 kernel without its matrix work
        constexpr size_t lds = 64 * 68 * sizeof(float);
        const unsigned grid = (unsigned)std::min<long>(cdiv(cdiv(n, 32), 4), tune("DNMF_UPD_GRID", 1024L));
        DNMF_LAUNCH((update_h_seq_kernel<2, 1, 4, false, false>), dim3(grid), dim3(256), lds, st, H, k, n, ldh, AtW, ldatw, G, eps, clamp);
        return check_launch("mu_update_h(nomma)");
    }
    UHS(1, 1, 5) UHS(2, 1, 5) UHS(1, 1, 6) UHS(2, 1, 6) UHS(1, 1, 8) UHS(4, 1, 3) UHS(4, 1, 5)
    UHS(1, 2, 4) UHS(2, 2, 4) UHS(4, 2, 3) UHS(1, 2, 5) UHS(1, 2, 6)
#endif
#undef UHS
    return fail(DNMF_EINVAL, "mu_update_h: no kernel for k tile %d / variant %d", kt, var);
}

int dnmf_clamp_min(float* X, long rows, long cols, long ldx, float eps, void* stream) {
    REQUIRE(X && rows >= 1 && cols >= 1 && ldx >= cols, "clamp_min: bad arguments");
    return launch_ew<EW_CLAMP>(X, rows, cols, ldx, nullptr, 0, nullptr, eps, 0, "clamp", S(stream));
}

int dnmf_scale_cols_div(float* W, long m, int k, long ldw, const float* s, float eps, void* stream) {
    REQUIRE(W && s && m >= 1 && k >= 1 && ldw >= k, "scale_cols_div: bad arguments");
    return launch_ew<EW_COLS_DIV>(W, m, k, ldw, nullptr, 0, s, eps, 0, "scale_cols_div", S(stream));
}

int dnmf_scale_rows_mul(float* H, int k, long n, long ldh, const float* s, void* stream) {
    REQUIRE(H && s && n >= 1 && k >= 1 && ldh >= n, "scale_rows_mul: bad arguments");
    return launch_ew<EW_ROWS_MUL>(H, k, n, ldh, nullptr, 0, s, 0.f, 0, "scale_rows_mul", S(stream));
}

}  // extern "C"
namespace {
// one wave that samples the shader clock while other kernels run (include/dnmf.h: dnmf_clock_probe)
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* buf, int n, int naps) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        const unsigned long long c = __builtin_amdgcn_s_memtime(), w = wall_clock64();
        __builtin_nontemporal_store(c, &buf[2 * i]);
        __builtin_nontemporal_store(w, &buf[2 * i + 1]);
        for (int s = 0; s < naps; ++s) __builtin_amdgcn_s_sleep(127);       // 127 x 64 cycles
    }
}

template <typename TA>
int sqnorm_impl(const TA* A, long m, long n, long lda, double* out, void* stream) {
    REQUIRE(A && out && m >= 1 && n >= 1 && lda >= n, "sqnorm: bad arguments");
    hipStream_t st = S(stream);
    if (batch_memset(out, 0, sizeof(double), st) != hipSuccess) return fail(DNMF_EHIP, "sqnorm: memset failed");
    const bool fast = a_aligned(A) && lda % 4 == 0 && n % 4 == 0;
    const long work = fast ? m * (n / 4) : m * n;
    const unsigned grid = (unsigned)std::min<long>(cdiv(work, 256), 2048);
    if (fast) DNMF_LAUNCH((sqnorm_kernel<true, TA>), dim3(grid), dim3(256), 0, st, A, m, n, lda, out);
    else DNMF_LAUNCH((sqnorm_kernel<false, TA>), dim3(grid), dim3(256), 0, st, A, m, n, lda, out);
    return check_launch("sqnorm");
}
}  // namespace
extern "C" {

int dnmf_sqnorm(const float* A, long m, long n, long lda, double* out, void* stream) {
    return sqnorm_impl<float>(A, m, n, lda, out, stream);
}
int dnmf_sqnorm_bf16a(const void* A, long m, long n, long lda, double* out, void* stream) {
    return sqnorm_impl<bf16_t>((const bf16_t*)A, m, n, lda, out, stream);
}

// X_per = X * (1 + nv + 2 nv U[0,1)) in one pass (pyDNMFk.py:42-44); `bf16` != 0: X and X_per are bfloat16 (config 5's storage)
int dnmf_perturb_uniform(const void* X, void* X_per, long rows, long cols, long ldx, long ldo, float noise_var,
                         unsigned long long seed, int bf16, void* stream) {
    REQUIRE(X && X_per && rows >= 1 && cols >= 1 && ldx >= cols && ldo >= cols, "perturb_uniform: bad arguments (cols %ld, ld %ld / %ld)", cols, ldx, ldo);
    hipStream_t st = S(stream);
    // rows of 16-byte aligned vectors of 8 elements: the vector kernel; anything else: one element per thread, SAME values
    const bool vec = cols % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && aligned16(X) && aligned16(X_per);
    const long total = vec ? rows * (cols / 8) : rows * cols;
    const unsigned grid = (unsigned)std::min<long>(cdiv(total, 256), 256L * 32);
    if (vec) {
        if (bf16) hipLaunchKernelGGL(perturb_uniform_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)X, (bf16_t*)X_per, rows, cols, ldx, ldo, noise_var, seed);
        else hipLaunchKernelGGL(perturb_uniform_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)X, (float*)X_per, rows, cols, ldx, ldo, noise_var, seed);
    } else {
        if (bf16) hipLaunchKernelGGL(perturb_uniform_any_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)X, (bf16_t*)X_per, rows, cols, ldx, ldo, noise_var, seed);
        else hipLaunchKernelGGL(perturb_uniform_any_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)X, (float*)X_per, rows, cols, ldx, ldo, noise_var, seed);
    }
    return check_launch("perturb_uniform");
}

int dnmf_clock_probe(unsigned long long* samples, int n, int naps, void* stream) {
    REQUIRE(samples && n >= 1 && naps >= 0 && naps <= 1000, "clock_probe: bad arguments");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, S(stream), samples, n, naps);
    return check_launch("clock_probe");
}

int dnmf_rowsum(const float* H, int k, long n, long ldh, float* x, void* stream) {
    REQUIRE(H && x && k >= 1 && n >= 1 && ldh >= n, "rowsum: bad arguments");
    DNMF_LAUNCH(rowsum_kernel, dim3(k), dim3(1024), 0, S(stream), H, n, ldh, x);
    return check_launch("rowsum");
}

int dnmf_colsum(const float* W, long m, int k, long ldw, float* x, void* ws, size_t ws_bytes, void* stream) {
    const int kp = kp_of(k);
    REQUIRE(kp > 0 && W && x && ws && m >= 1 && ldw >= k, "colsum: bad arguments");
    // slabs of 128 rows (a 32768-row shard gives 256 workgroups; 1024-row slabs left it on 32 CUs: 116 us for 16 MiB),
    // at most 1024 slabs; the partials are summed in slab order by one small launch
    const long rows_per_blk = std::max<long>(128, round_up(cdiv(m, 1024), 8));
    const int nblk = (int)cdiv(m, rows_per_blk);
    if (ws_bytes < (size_t)nblk * kp * sizeof(float)) return fail(DNMF_EWS, "colsum: workspace too small");
    DNMF_LAUNCH(colsum_partial_kernel, dim3(nblk), dim3(256), 0, S(stream), W, m, k, ldw, rows_per_blk, (float*)ws, kp);
    DNMF_LAUNCH(colsum_final_kernel, dim3(1), dim3(1024), 0, S(stream), (const float*)ws, nblk, kp, k, x);
    return check_launch("colsum");
}

int dnmf_kl_update_w(float* W, long m, int k, long ldw, const float* Sm, long lds_, const float* x, float eps,
                     void* stream) {
    REQUIRE(W && Sm && x && m >= 1 && k >= 1 && ldw >= k && lds_ >= k, "kl_update_w: bad arguments");
    return launch_ew<EW_KL_BYCOL>(W, m, k, ldw, Sm, lds_, x, eps, 0, "kl_update_w", S(stream));
}

int dnmf_kl_update_h(float* H, int k, long n, long ldh, const float* Sm, long lds_, const float* x, float eps,
                     int clamp, void* stream) {
    REQUIRE(H && Sm && x && n >= 1 && k >= 1 && ldh >= n && lds_ >= n, "kl_update_h: bad arguments");
    return launch_ew<EW_KL_BYROW>(H, k, n, ldh, Sm, lds_, x, eps, clamp, "kl_update_h", S(stream));
}

}  // extern "C"
namespace {
template <typename TA>
int mu_fro_step_impl(const TA* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k,
                     float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream) {
    REQUIRE(kp_of(k) > 0 && A && W && H && ws && m >= 1 && n >= 1, "mu_fro_step: bad arguments");
    const WsLayout L = ws_layout(m, n, k);
    if (ws_bytes < L.total) return fail(DNMF_EWS, "mu_fro_step: workspace %zu < %zu", ws_bytes, L.total);
    char* base = (char*)ws;
    float* G = (float*)(base + L.g_off);
    float* Sb = (float*)(base + L.s_off);
    void* part = base + L.part_off;
    const size_t part_bytes = L.total - L.part_off;
    int rc;
    if constexpr (std::is_same<TA, float>::value) {
        // 16 < k <= 32 on whole 512-column pieces: ONE pass over A (csrc/dnmf_team.h) -- the W update of a row block and its share of
        // W^T A while the block is on chip; H H^T before, W^T W after and the H update are the launches of the two-pass sequence
        if (w_update && dnmf_team_ws_bytes_(m, n, k)) {
            if ((rc = dnmf_gram_hht(H, k, n, ldh, G, part, part_bytes, stream))) return rc;
            const float *P = nullptr, *Pg = nullptr;
            int nparts = 0, tkp = 32;
            rc = dnmf_team_fro_(A, m, n, lda, W, ldw, H, ldh, G, k, eps, part, part_bytes, stream, &P, &nparts, &Pg, &tkp);
            if (rc < 0) return rc;
            if (rc == DNMF_OK) {
                // the teams' partials of W^T A -> S, and in the same launch the teams' partial Gram tiles -> G = W_new^T W_new (:748; the
                // kernel read H H^T from G: it has finished -- same stream)
                const long ldatw = round_up(n, 4);
                const GramTail gt{Pg, G, tkp, k, 32, nparts};
                if ((rc = launch_reduce(P, (long)tkp * n, n, nparts, Sb, ldatw, k, n, k, n, nullptr, S(stream), &gt))) return rc;
                if ((rc = dnmf_mu_update_h(H, k, n, ldh, Sb, ldatw, G, eps, clamp, stream))) return rc;
                if (clamp) return dnmf_clamp_min(W, m, k, ldw, eps, stream);
                return DNMF_OK;
            }
            w_update = -1;                                                            // H H^T is in G already
        }
    }
    if (w_update) {                                                                   // dist_nmf.py:716-732
        if (w_update > 0 && (rc = dnmf_gram_hht(H, k, n, ldh, G, part, part_bytes, stream))) return rc;
        if (wide_k(k)) {                                             // beyond the fused kernel's rank: the product, then the update
            const long ldah = round_up(k, 4);
            if ((rc = aht_impl<TA>(A, m, n, lda, H, k, ldh, Sb, ldah, stream, 0))) return rc;
            if ((rc = dnmf_mu_update_w(W, m, k, ldw, Sb, ldah, G, eps, stream))) return rc;
        } else if ((rc = aht_update_w_impl<TA>(A, m, n, lda, H, k, ldh, G, W, ldw, eps, stream))) return rc;
    }
    const long ldatw = round_up(n, 4);                                                // dist_nmf.py:736-751
    if ((rc = wta_impl<TA>(A, m, n, lda, W, k, ldw, Sb, ldatw, part, part_bytes, stream, G))) return rc;   // + W^T W (:705)
    if ((rc = dnmf_mu_update_h(H, k, n, ldh, Sb, ldatw, G, eps, clamp, stream))) return rc;
    if (clamp) return dnmf_clamp_min(W, m, k, ldw, eps, stream);                      // pyDNMF.py:155-157
    return DNMF_OK;
}
}  // namespace
extern "C" {

int dnmf_mu_fro_step(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k,
                     float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream) {
    return mu_fro_step_impl<float>(A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, clamp, ws, ws_bytes, stream);
}
int dnmf_mu_fro_step_bf16a(const void* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k,
                           float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream) {
    return mu_fro_step_impl<bf16_t>((const bf16_t*)A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, clamp, ws, ws_bytes, stream);
}

int dnmf_mu_kl_step(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps,
                    int w_update, int clamp, void* ws, size_t ws_bytes, void* stream) {
    REQUIRE(kp_of(k) > 0 && A && W && H && ws && m >= 1 && n >= 1, "mu_kl_step: bad arguments");
    const WsLayout L = ws_layout(m, n, k);
    if (ws_bytes < L.total) return fail(DNMF_EWS, "mu_kl_step: workspace %zu < %zu", ws_bytes, L.total);
    char* base = (char*)ws;
    float* Sb = (float*)(base + L.s_off);
    float* x = (float*)(base + L.x_off);
    void* part = base + L.part_off;
    const size_t part_bytes = L.total - L.part_off;
    int rc;
    if (w_update) {                                                                   // dist_nmf.py:813-830
        const long ldu = round_up(k, 4);
        if ((rc = dnmf_rowsum(H, k, n, ldh, x, stream))) return rc;
        if ((rc = dnmf_kl_uht(A, m, n, lda, W, ldw, H, ldh, k, eps, Sb, ldu, part, part_bytes, stream))) return rc;
        if ((rc = dnmf_kl_update_w(W, m, k, ldw, Sb, ldu, x, eps, stream))) return rc;
    }
    const long ldo = round_up(n, 4);                                                  // dist_nmf.py:832-849
    if ((rc = dnmf_colsum(W, m, k, ldw, x, part, part_bytes, stream))) return rc;
    if ((rc = dnmf_kl_wtu(A, m, n, lda, W, ldw, H, ldh, k, eps, Sb, ldo, part, part_bytes, stream))) return rc;
    if ((rc = dnmf_kl_update_h(H, k, n, ldh, Sb, ldo, x, eps, clamp, stream))) return rc;
    if (clamp) return dnmf_clamp_min(W, m, k, ldw, eps, stream);
    return DNMF_OK;
}

}  // extern "C"
