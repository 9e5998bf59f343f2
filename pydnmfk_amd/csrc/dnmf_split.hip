// dnmf_split.hip -- C ABI of the bf16x6 contractions (kernels and arithmetic: dnmf_split.h).  Second translation unit of
// libdnmf_hip.so; everything else of a step (Gram matrices, the H update, clamps) is the fp32 code of dnmf.hip.
#include "dnmf_common.h"
#include "dnmf_host.h"
#include "dnmf_split.h"
#include "dnmf_split_kl.h"

// workspace layout of the fused steps, owned by dnmf.hip: {g_off, s_off, x_off, part_off, total}
__attribute__((visibility("hidden"))) void dnmf_ws_offsets_(long m, long n, int k, size_t out[5]);

namespace {

// The split kernels exist for 32 < k <= 128 (below that the fp32 kernels are already bound by the HBM, not by the matrix
// cores) and take A in whole 128-byte lines: 16-byte aligned rows, n a multiple of 128.  Every other shape runs the fp32
// kernels of dnmf.hip -- the caller gets the exact products instead.
template <typename TA>
bool split_shape(const TA* A, long m, long n, long lda, int k) {
    constexpr long V = 16 / sizeof(TA);                              // elements per 16 bytes
    // fp32 A at k <= 32 streams at the HBM rate on the fp32 kernels already.  bf16-stored A is matrix-pipe bound on them from
    // k = 17 on (the 32-wide kernels; iteration at 262144 x 8192, k = 32: 2.39 ms, three-product kernels 1.48 ms); at k <= 16 the
    // 16-wide fp32 kernels are as fast (1.39-1.43 ms vs 1.43-1.46 ms) and stay.
    const int kmin = std::is_same<TA, bf16_t>::value ? 17 : 33;
    return k >= kmin && k <= DNMF_TUNED_MAX_K && n % 128 == 0 && lda % V == 0 && aligned16(A) && lda >= n && lda <= 4 * n && m >= 1 &&
           n < (1L << 19) && tune("DNMF_SPLIT", 1) != 0;
}

struct TnxPlan { int nt, ncolblk, nchunks; long rows_per_chunk, ldp, chunk_stride; };

TnxPlan plan_tnx(long m, long n, int kt) {
    TnxPlan p;
    p.nt = kt == 4 ? 2 : 4;
    p.ncolblk = (int)(n / (32 * p.nt));
    const long cb4 = cdiv(p.ncolblk, 4);
    static const long target = tune("DNMF_TNX_WGS", 512);           // one resident round: 256 CUs x 2 workgroups
    long nchunks = std::max<long>(1, target / cb4);
    nchunks = std::min<long>(nchunks, cdiv(m, 256));
    // a chunk is addressed with 32-bit offsets from its first row: rows_per_chunk * lda * 4 < 2^31 for every lda <= 4 n that
    // split_shape admits (column slices of a wider matrix: the overlapped H phase of dist_nmf.py works on halves of A)
    nchunks = std::max<long>(nchunks, cdiv(m * n * 16, 1L << 30));        // + the 64-row rounding: 1024 n < 2^29
    p.rows_per_chunk = round_up(cdiv(m, nchunks), XK);
    p.nchunks = (int)cdiv(m, p.rows_per_chunk);
    p.ldp = n;
    p.chunk_stride = 32L * kt * n;
    return p;
}

size_t h_image_bytes(long n, int kp) { return align256((size_t)3 * kp * round_up(n, XK) * sizeof(bf16_t)); }
size_t w_image_bytes(long m, int kp) { return align256((size_t)3 * kp * round_up(m, XK) * sizeof(bf16_t)); }

size_t wta_need(long m, long n, int k) {
    const int kt = kt_of(k), kp = 32 * kt;
    const TnxPlan p = plan_tnx(m, n, kt);
    return w_image_bytes(m, kp) + align256((size_t)p.nchunks * p.chunk_stride * sizeof(float)) + reduce_scratch_bytes(p.nchunks, k, n);
}

int cut_h(const float* H, int k, int kp, long n, long ldh, bf16_t* img, SplitOperand& o, hipStream_t st) {
    o.ld = round_up(n, XK);
    o.split_stride = kp * o.ld;
    o.S = img;
    const long threads = kp * (o.ld / 8);
    hipLaunchKernelGGL(split3_rows_kernel, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, st, H, ldh, k, n, img, o.ld,
                       o.split_stride, kp);
    return check_launch("split3_rows");
}

int cut_wt(const float* W, long m, int k, int kp, long ldw, bf16_t* img, SplitOperand& o, hipStream_t st) {
    o.ld = round_up(m, XK);
    o.split_stride = kp * o.ld;
    o.S = img;
    hipLaunchKernelGGL(split3_cols_kernel, dim3((unsigned)(o.ld / 64), (unsigned)(kp / 32)), dim3(256), 0, st, W, ldw, m, k, img, o.ld, o.split_stride);
    return check_launch("split3_cols");
}

template <int KT, int MODE, typename TX, int NW = 4>
int launch_ntx_kt(const NtArgs& a, const SplitOperand& ys, hipStream_t st) {
    // two stages [A tile, 32 NW rows x 128 bytes | H tile bf16 pieces (32 indices per row for fp32 A, 64 for bf16 A)]; >= the W.G loop's LDS
    constexpr size_t lds = 2 * (32 * NW * 128 + 3 * 32 * KT * (std::is_same<TX, bf16_t>::value ? 128 : 64));
    static bool once = false;
    // A is touched once: stream it past the caches when it cannot stay in them anyway
    const bool nt = (double)a.nrows * a.ncols * sizeof(TX) >= 256.0 * (1 << 20) && tune("DNMF_SPLIT_NT", 1) != 0;
    if (!once) { allow_lds(ntx_kernel<KT, MODE, 0, TX, NW>, lds); allow_lds(ntx_kernel<KT, MODE, 2, TX, NW>, lds); once = true; }
    const dim3 grid((unsigned)cdiv(a.nrows, 32 * NW), 1);
    if (nt) hipLaunchKernelGGL((ntx_kernel<KT, MODE, 2, TX, NW>), grid, dim3(64 * NW), lds, st, a, ys);
    else hipLaunchKernelGGL((ntx_kernel<KT, MODE, 0, TX, NW>), grid, dim3(64 * NW), lds, st, a, ys);
    return check_launch("ntx_kernel");
}

template <int MODE, typename TX>
int launch_ntx(const NtArgs& a, const SplitOperand& ys, int kt, hipStream_t st) {
    if constexpr (std::is_same<TX, bf16_t>::value) {
        if (kt == 1) return launch_ntx_kt<1, MODE, TX>(a, ys, st);
    }
#ifdef DNMF_TUNING
    // 192-row workgroups = three waves per SIMD (166 registers, two register sets in flight): measured 2.16 ms against 1.91 ms
    // for the 128-row form on 262144 x 8192 -- more resident waves do not fill the idle issue slots.  A/B runs only.
    if constexpr (std::is_same<TX, float>::value)
        if (kt == 2 && tune("DNMF_SPLIT_NW6", 0) != 0) return launch_ntx_kt<2, MODE, TX, 6>(a, ys, st);
#endif
    return kt == 2 ? launch_ntx_kt<2, MODE, TX>(a, ys, st) : launch_ntx_kt<4, MODE, TX>(a, ys, st);
}

template <int KT, int NT, int XKT, typename TX>
int launch_tnx(const TnArgs& a, const SplitOperand& wsplit, const TnxPlan& p, long m, long n, hipStream_t st) {
    constexpr size_t lds = 2 * 3 * 32 * KT * (XKT / 8) * 16;
    static bool once = false;
    if (!once) { allow_lds(tnx_kernel<KT, NT, XKT, 0, TX>, lds); allow_lds(tnx_kernel<KT, NT, XKT, 2, TX>, lds); once = true; }
    const bool nt = (double)m * n * sizeof(TX) >= 256.0 * (1 << 20) && tune("DNMF_SPLIT_NT", 1) != 0;
    const dim3 grid((unsigned)(cdiv(p.ncolblk, 4) * p.nchunks));
    if (nt) hipLaunchKernelGGL((tnx_kernel<KT, NT, XKT, 2, TX>), grid, dim3(256), lds, st, a, wsplit);
    else hipLaunchKernelGGL((tnx_kernel<KT, NT, XKT, 0, TX>), grid, dim3(256), lds, st, a, wsplit);
    return check_launch("tnx_kernel");
}

// ---- the four Frobenius entry points, for fp32 and for bf16-stored A (fallbacks: the fp32-MFMA entry points of dnmf.hip)
template <typename TA> struct Fp32Twin;
template <> struct Fp32Twin<float> {
    static int aht(const float* A, long m, long n, long lda, const float* H, int k, long ldh, float* AH, long ldah, void* s) { return dnmf_aht(A, m, n, lda, H, k, ldh, AH, ldah, s); }
    static int wta(const float* A, long m, long n, long lda, const float* W, int k, long ldw, float* o, long ldo, void* ws, size_t wb, void* s) { return dnmf_wta(A, m, n, lda, W, k, ldw, o, ldo, ws, wb, s); }
    static int ahtw(const float* A, long m, long n, long lda, const float* H, int k, long ldh, const float* G, float* W, long ldw, float eps, void* s) { return dnmf_aht_update_w(A, m, n, lda, H, k, ldh, G, W, ldw, eps, s); }
    static int step(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps, int wu, int cl, void* ws, size_t wb, void* s) { return dnmf_mu_fro_step(A, m, n, lda, W, ldw, H, ldh, k, eps, wu, cl, ws, wb, s); }
};
template <> struct Fp32Twin<bf16_t> {
    static int aht(const bf16_t* A, long m, long n, long lda, const float* H, int k, long ldh, float* AH, long ldah, void* s) { return dnmf_aht_bf16a(A, m, n, lda, H, k, ldh, AH, ldah, s); }
    static int wta(const bf16_t* A, long m, long n, long lda, const float* W, int k, long ldw, float* o, long ldo, void* ws, size_t wb, void* s) { return dnmf_wta_bf16a(A, m, n, lda, W, k, ldw, o, ldo, ws, wb, s); }
    static int ahtw(const bf16_t* A, long m, long n, long lda, const float* H, int k, long ldh, const float* G, float* W, long ldw, float eps, void* s) { return dnmf_aht_update_w_bf16a(A, m, n, lda, H, k, ldh, G, W, ldw, eps, s); }
    static int step(const bf16_t* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps, int wu, int cl, void* ws, size_t wb, void* s) { return dnmf_mu_fro_step_bf16a(A, m, n, lda, W, ldw, H, ldh, k, eps, wu, cl, ws, wb, s); }
};

template <typename TA>
int ahtw_x6(const TA* A, long m, long n, long lda, const float* H, int k, long ldh, const float* G, float* W, long ldw, float eps,
            void* ws, size_t ws_bytes, void* stream) {
    if (!split_shape(A, m, n, lda, k)) return Fp32Twin<TA>::ahtw(A, m, n, lda, H, k, ldh, G, W, ldw, eps, stream);
    REQUIRE(H && G && W && ws && ldh >= n && ldw >= k, "aht_update_w_bf16x6: bad arguments");
    const int kt = kt_of(k), kp = 32 * kt;
    if (ws_bytes < h_image_bytes(n, kp)) return fail(DNMF_EWS, "aht_update_w_bf16x6: workspace %zu < %zu", ws_bytes, h_image_bytes(n, kp));
    hipStream_t st = S(stream);
    SplitOperand ys;
    int rc = cut_h(H, k, kp, n, ldh, (bf16_t*)ws, ys, st);
    if (rc) return rc;
    NtArgs a{};
    a.X = A; a.ldx = lda; a.nrows = m; a.ncols = n; a.Y = H; a.ldy = ldh; a.yrows = k; a.cols_per_split = n;
    a.W = W; a.ldw = ldw; a.G = G; a.eps = eps; a.k = k;
    a.wfast = aligned16(W) && ldw % 4 == 0 && k % 4 == 0;
    return launch_ntx<NT_FUSED_W, TA>(a, ys, kt, st);
}

template <typename TA>
int aht_x6(const TA* A, long m, long n, long lda, const float* H, int k, long ldh, float* AH, long ldah, void* ws, size_t ws_bytes,
           void* stream) {
    if (!split_shape(A, m, n, lda, k)) return Fp32Twin<TA>::aht(A, m, n, lda, H, k, ldh, AH, ldah, stream);
    REQUIRE(H && AH && ws && ldh >= n && ldah >= k, "aht_bf16x6: bad arguments");
    const int kt = kt_of(k), kp = 32 * kt;
    if (ws_bytes < h_image_bytes(n, kp)) return fail(DNMF_EWS, "aht_bf16x6: workspace %zu < %zu", ws_bytes, h_image_bytes(n, kp));
    hipStream_t st = S(stream);
    SplitOperand ys;
    int rc = cut_h(H, k, kp, n, ldh, (bf16_t*)ws, ys, st);
    if (rc) return rc;
    NtArgs a{};
    a.X = A; a.ldx = lda; a.nrows = m; a.ncols = n; a.Y = H; a.ldy = ldh; a.yrows = k; a.cols_per_split = n;
    a.out = AH; a.ldo = ldah; a.split_stride = 0; a.store_all = 0;
    return launch_ntx<NT_STORE, TA>(a, ys, kt, st);
}

template <typename TA>
int wta_x6(const TA* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw, void* ws, size_t ws_bytes,
           void* stream) {
    if (!split_shape(A, m, n, lda, k)) return Fp32Twin<TA>::wta(A, m, n, lda, W, k, ldw, AtW, ldatw, ws, ws_bytes, stream);
    REQUIRE(W && AtW && ws && ldw >= k && ldatw >= n, "wta_bf16x6: bad arguments");
    const size_t need = wta_need(m, n, k);
    if (ws_bytes < need) return fail(DNMF_EWS, "wta_bf16x6: workspace %zu < %zu", ws_bytes, need);
    const int kt = kt_of(k), kp = 32 * kt;
    hipStream_t st = S(stream);
    SplitOperand wsplit;
    int rc = cut_wt(W, m, k, kp, ldw, (bf16_t*)ws, wsplit, st);
    if (rc) return rc;
    const TnxPlan p = plan_tnx(m, n, kt);
    float* P = (float*)((char*)ws + w_image_bytes(m, kp));
    const size_t pbytes = align256((size_t)p.nchunks * p.chunk_stride * sizeof(float));
    TnArgs a{};
    a.X = W; a.ldx = ldw; a.xcols = k; a.Y = A; a.ldy = lda; a.ycols = n;
    a.nrows = m; a.rows_per_chunk = p.rows_per_chunk; a.nchunks = p.nchunks; a.ncolblk = p.ncolblk;
    a.P = P; a.chunk_stride = p.chunk_stride; a.ldp = p.ldp;
    if constexpr (std::is_same<TA, bf16_t>::value) {
        if (kt == 1) rc = launch_tnx<1, 4, 64, TA>(a, wsplit, p, m, n, st);
    }
    if (kt != 1) rc = kt == 2 ? launch_tnx<2, 4, 64, TA>(a, wsplit, p, m, n, st) : launch_tnx<4, 2, 32, TA>(a, wsplit, p, m, n, st);
    if (rc) return rc;
    return launch_reduce(P, p.chunk_stride, p.ldp, p.nchunks, AtW, ldatw, k, n, k, n, (float*)((char*)P + pbytes), st);
}

// ---------------------------------------------------------------------------------------------- KL products
// whole 128-column blocks, 16-byte aligned rows of A; other shapes run the fp32 kernels.
bool klx_shape(const float* A, long m, long n, long lda, int k) {
    return k >= 1 && k <= DNMF_TUNED_MAX_K && n % 128 == 0 && lda % 4 == 0 && aligned16(A) && lda >= n && lda <= 4 * n && m >= 1 &&
           n < (1L << 19) && tune("DNMF_SPLIT_KL", 1) != 0;
}

struct KlxImages { SplitOperand wp, ht; };
long klx_mpad(long m) { return round_up(m, 128); }
size_t klx_w_bytes(long m, int kp) { return align256((size_t)3 * klx_mpad(m) * kp * sizeof(bf16_t)); }
size_t klx_h_bytes(long n, int kp) { return align256((size_t)3 * n * kp * sizeof(bf16_t)); }

struct WtuxPlan { int nt, ncolblk; long rowblks_per_chunk, nchunks, ldp; };
WtuxPlan plan_wtux(long m, long n, int kt) {
    WtuxPlan p;
    p.nt = 2;
    p.ncolblk = (int)(n / (32 * p.nt));
    const long nrowblk = cdiv(m, 32);
    const long target = kt == 1 ? 3072 : (kt == 2 ? 2048 : 1024);      // waves: one resident round
    long nchunks = std::max<long>(1, target / p.ncolblk);
    nchunks = std::min<long>(nchunks, std::max<long>(1, nrowblk / 4));
    nchunks = std::max<long>(nchunks, cdiv(m * n * 16, 1L << 30));   // 32-bit offsets inside a chunk, lda <= 4 n
    p.rowblks_per_chunk = cdiv(nrowblk, nchunks);
    p.nchunks = cdiv(nrowblk, p.rowblks_per_chunk);
    p.ldp = n;
    return p;
}
size_t wtux_need(long m, long n, int k) {
    const int kt = kt_of(k), kp = 32 * kt;
    const WtuxPlan p = plan_wtux(m, n, kt);
    return align256((size_t)p.nchunks * kp * p.ldp * sizeof(float)) + reduce_scratch_bytes((int)p.nchunks, k, n);
}

struct UhtxPlan { int nsplit; long cols_per_split; };
UhtxPlan plan_uhtx(long m, long n) {
    UhtxPlan u;
    const long rowtiles = cdiv(m, 128);
    long ns = std::max<long>(1, cdiv(1024, rowtiles));
    ns = std::min<long>(ns, std::max<long>(1, n / 256));
    u.cols_per_split = round_up(cdiv(n, ns), 64);                    // two 32-column tiles per trip of klx_uht_kernel
    u.nsplit = (int)cdiv(n, u.cols_per_split);
    return u;
}
size_t uhtx_need(long m, long n, int k) {
    const int kp = 32 * kt_of(k);
    const UhtxPlan u = plan_uhtx(m, n);
    return u.nsplit > 1 ? align256((size_t)u.nsplit * m * kp * sizeof(float)) + reduce_scratch_bytes(u.nsplit, (int)m, k) : 0;
}
size_t klx_need(long m, long n, int k) {                              // images + the larger of the two partial areas
    const int kp = 32 * kt_of(k);
    return klx_w_bytes(m, kp) + klx_h_bytes(n, kp) + std::max(wtux_need(m, n, k), uhtx_need(m, n, k));
}

// wp [piece][m_pad][kp] = pieces of W as it lies in memory; ht [piece][n][kp] = pieces of H transposed
int klx_images(const float* W, long m, long ldw, const float* H, long n, long ldh, int k, int kp, char* base, KlxImages& im,
               hipStream_t st) {
    const long mp = klx_mpad(m);
    im.wp.S = (bf16_t*)base; im.wp.ld = kp; im.wp.split_stride = mp * kp;
    im.ht.S = (bf16_t*)(base + klx_w_bytes(m, kp)); im.ht.ld = kp; im.ht.split_stride = n * kp;
    hipLaunchKernelGGL(split3_rows_kernel, dim3((unsigned)cdiv(mp * (kp / 8), 256)), dim3(256), 0, st, W, ldw,
                       (int)std::min<long>(m, INT32_MAX), (long)k, const_cast<bf16_t*>(im.wp.S), (long)kp, im.wp.split_stride, (int)mp);
    hipLaunchKernelGGL(split3_cols_kernel, dim3((unsigned)cdiv(kp, 64), (unsigned)(n / 32)), dim3(256), 0, st, H, ldh, (long)k, (int)n,
                       const_cast<bf16_t*>(im.ht.S), (long)kp, im.ht.split_stride);
    return check_launch("split3 (KL images)");
}

}  // namespace

extern "C" {

size_t dnmf_ws_bytes_bf16x6(long m, long n, int k) {
    const size_t base = dnmf_ws_bytes(m, n, k);
    if (!base) return 0;
    if (k > DNMF_TUNED_MAX_K) return align256(base);              // (wide ranks: every entry point forwards to the fp32 path)
    size_t extra = 0;
    if (n % 128 == 0) extra = h_image_bytes(n, 32 * kt_of(k)) + wta_need(m, n, k);      // (k <= 32: bf16-stored A only)
    if (n % 128 == 0) extra = std::max(extra, klx_need(m, n, k));
    return align256(base) + extra;
}

int dnmf_aht_update_w_bf16x6(const float* A, long m, long n, long lda, const float* H, int k, long ldh, const float* G,
                             float* W, long ldw, float eps, void* ws, size_t ws_bytes, void* stream) {
    return ahtw_x6<float>(A, m, n, lda, H, k, ldh, G, W, ldw, eps, ws, ws_bytes, stream);
}
int dnmf_aht_update_w_bf16a_bf16x6(const void* A, long m, long n, long lda, const float* H, int k, long ldh, const float* G,
                                   float* W, long ldw, float eps, void* ws, size_t ws_bytes, void* stream) {
    return ahtw_x6<bf16_t>((const bf16_t*)A, m, n, lda, H, k, ldh, G, W, ldw, eps, ws, ws_bytes, stream);
}

int dnmf_aht_bf16x6(const float* A, long m, long n, long lda, const float* H, int k, long ldh, float* AH, long ldah,
                    void* ws, size_t ws_bytes, void* stream) {
    return aht_x6<float>(A, m, n, lda, H, k, ldh, AH, ldah, ws, ws_bytes, stream);
}
int dnmf_aht_bf16a_bf16x6(const void* A, long m, long n, long lda, const float* H, int k, long ldh, float* AH, long ldah,
                          void* ws, size_t ws_bytes, void* stream) {
    return aht_x6<bf16_t>((const bf16_t*)A, m, n, lda, H, k, ldh, AH, ldah, ws, ws_bytes, stream);
}

int dnmf_wta_bf16x6(const float* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw,
                    void* ws, size_t ws_bytes, void* stream) {
    return wta_x6<float>(A, m, n, lda, W, k, ldw, AtW, ldatw, ws, ws_bytes, stream);
}
int dnmf_wta_bf16a_bf16x6(const void* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw,
                          void* ws, size_t ws_bytes, void* stream) {
    return wta_x6<bf16_t>((const bf16_t*)A, m, n, lda, W, k, ldw, AtW, ldatw, ws, ws_bytes, stream);
}

}  // extern "C"
namespace {
template <typename TA>
int step_x6(const TA* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k,
            float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream) {
    if (!split_shape(A, m, n, lda, k)) return Fp32Twin<TA>::step(A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, clamp, ws, ws_bytes, stream);
    REQUIRE(W && H && ws, "mu_fro_step_bf16x6: bad arguments");
    const size_t total = dnmf_ws_bytes_bf16x6(m, n, k);
    if (ws_bytes < total) return fail(DNMF_EWS, "mu_fro_step_bf16x6: workspace %zu < %zu", ws_bytes, total);
    size_t off[5];
    dnmf_ws_offsets_(m, n, k, off);
    char* base = (char*)ws;
    float* G = (float*)(base + off[0]);
    float* Sb = (float*)(base + off[1]);
    void* part = base + off[3];
    const size_t part_bytes = off[4] - off[3];
    char* img = base + align256(off[4]);
    const size_t img_bytes = total - align256(off[4]);
    int rc;
    if (w_update) {                                                                   // dist_nmf.py:716-732
        if ((rc = dnmf_gram_hht(H, k, n, ldh, G, part, part_bytes, stream))) return rc;
        if ((rc = ahtw_x6<TA>(A, m, n, lda, H, k, ldh, G, W, ldw, eps, img, img_bytes, stream))) return rc;
    }
    const long ldatw = round_up(n, 4);                                                // dist_nmf.py:736-751
    if ((rc = dnmf_gram_wtw(W, m, k, ldw, G, part, part_bytes, stream))) return rc;
    if ((rc = wta_x6<TA>(A, m, n, lda, W, k, ldw, Sb, ldatw, img + h_image_bytes(n, 32 * kt_of(k)), img_bytes - h_image_bytes(n, 32 * kt_of(k)), stream))) return rc;
    if ((rc = dnmf_mu_update_h(H, k, n, ldh, Sb, ldatw, G, eps, clamp, stream))) return rc;
    if (clamp) return dnmf_clamp_min(W, m, k, ldw, eps, stream);                      // pyDNMF.py:155-157
    return DNMF_OK;
}

}  // namespace
extern "C" {

int dnmf_mu_fro_step_bf16x6(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k,
                            float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream) {
    return step_x6<float>(A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, clamp, ws, ws_bytes, stream);
}
int dnmf_mu_fro_step_bf16a_bf16x6(const void* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k,
                                  float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream) {
    return step_x6<bf16_t>((const bf16_t*)A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, clamp, ws, ws_bytes, stream);
}

int dnmf_kl_wtu_bf16x6(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k,
                       float eps, float* WTU, long ldo, void* ws, size_t ws_bytes, void* stream) {
    if (!klx_shape(A, m, n, lda, k)) return dnmf_kl_wtu(A, m, n, lda, W, ldw, H, ldh, k, eps, WTU, ldo, ws, ws_bytes, stream);
    REQUIRE(W && H && WTU && ws && ldw >= k && ldh >= n && ldo >= n, "kl_wtu_bf16x6: bad arguments");
    const int kt = kt_of(k), kp = 32 * kt;
    const size_t img = klx_w_bytes(m, kp) + klx_h_bytes(n, kp), need = img + wtux_need(m, n, k);
    if (ws_bytes < need) return fail(DNMF_EWS, "kl_wtu_bf16x6: workspace %zu < %zu", ws_bytes, need);
    hipStream_t st = S(stream);
    KlxArgs a{};
    KlxImages im;
    int rc = klx_images(W, m, ldw, H, n, ldh, k, kp, (char*)ws, im, st);
    if (rc) return rc;
    const WtuxPlan p = plan_wtux(m, n, kt);
    a.A = A; a.lda = lda; a.m = m; a.n = n; a.eps = eps; a.k = k;
    a.wp = im.wp; a.ht = im.ht;
    a.P = (float*)((char*)ws + img); a.ldp = p.ldp; a.chunk_stride = p.ldp * kp;
    a.nrowblk = cdiv(m, 32); a.ncolblk = p.ncolblk; a.rowblks_per_chunk = p.rowblks_per_chunk; a.nchunks = p.nchunks;
    const dim3 grid((unsigned)(cdiv(p.nchunks, 4) * p.ncolblk)), block(256);
    const size_t lds = (size_t)3 * 32 * p.nt * kp * sizeof(bf16_t) + 4ul * 3 * 32 * kp * sizeof(bf16_t);   // H^T tile + 4 W block images
    static bool once = false;
    if (!once) {
        allow_lds(klx_wtu_kernel<1, 2>, 3 * 64 * 32 * 2 + 12 * 32 * 32 * 2); allow_lds(klx_wtu_kernel<2, 2>, 3 * 64 * 64 * 2 + 12 * 32 * 64 * 2);
        allow_lds(klx_wtu_kernel<4, 2>, 3 * 64 * 128 * 2 + 12 * 32 * 128 * 2);
        once = true;
    }
    if (kt == 1) hipLaunchKernelGGL((klx_wtu_kernel<1, 2>), grid, block, lds, st, a);
    else if (kt == 2) hipLaunchKernelGGL((klx_wtu_kernel<2, 2>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((klx_wtu_kernel<4, 2>), grid, block, lds, st, a);
    if ((rc = check_launch("klx_wtu_kernel"))) return rc;
    const size_t pbytes = align256((size_t)p.nchunks * kp * p.ldp * sizeof(float));
    return launch_reduce(a.P, a.chunk_stride, a.ldp, (int)p.nchunks, WTU, ldo, k, n, k, n, (float*)((char*)a.P + pbytes), st);
}

int dnmf_kl_uht_bf16x6(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k,
                       float eps, float* UHT, long ldo, void* ws, size_t ws_bytes, void* stream) {
    if (!klx_shape(A, m, n, lda, k)) return dnmf_kl_uht(A, m, n, lda, W, ldw, H, ldh, k, eps, UHT, ldo, ws, ws_bytes, stream);
    REQUIRE(W && H && UHT && ws && ldw >= k && ldh >= n && ldo >= k, "kl_uht_bf16x6: bad arguments");
    const int kt = kt_of(k), kp = 32 * kt;
    const size_t img = klx_w_bytes(m, kp) + klx_h_bytes(n, kp), need = img + uhtx_need(m, n, k);
    if (ws_bytes < need) return fail(DNMF_EWS, "kl_uht_bf16x6: workspace %zu < %zu", ws_bytes, need);
    hipStream_t st = S(stream);
    KlxArgs a{};
    KlxImages im;
    int rc = klx_images(W, m, ldw, H, n, ldh, k, kp, (char*)ws, im, st);
    if (rc) return rc;
    const UhtxPlan u = plan_uhtx(m, n);
    const bool split = u.nsplit > 1;
    a.A = A; a.lda = lda; a.m = m; a.n = n; a.eps = eps; a.k = k;
    a.wp = im.wp; a.ht = im.ht;
    a.out = split ? (float*)((char*)ws + img) : UHT;
    a.ldo = split ? kp : ldo; a.split_stride = (long)m * kp; a.cols_per_split = u.cols_per_split; a.out_cols = split ? kp : k;
    const dim3 grid((unsigned)cdiv(m, 128), (unsigned)u.nsplit), block(256);
    static bool once = false;
    if (!once) {
        allow_lds(klx_uht_kernel<1>, 2 * UhtStage<1>::BYTES); allow_lds(klx_uht_kernel<2>, 2 * UhtStage<2>::BYTES);
        allow_lds(klx_uht_kernel<4>, 2 * UhtStage<4>::BYTES);
        once = true;
    }
    if (kt == 1) hipLaunchKernelGGL((klx_uht_kernel<1>), grid, block, 2 * UhtStage<1>::BYTES, st, a);
    else if (kt == 2) hipLaunchKernelGGL((klx_uht_kernel<2>), grid, block, 2 * UhtStage<2>::BYTES, st, a);
    else hipLaunchKernelGGL((klx_uht_kernel<4>), grid, block, 2 * UhtStage<4>::BYTES, st, a);
    if ((rc = check_launch("klx_uht_kernel")) || !split) return rc;
    const size_t pbytes = align256((size_t)u.nsplit * m * kp * sizeof(float));
    return launch_reduce(a.out, a.split_stride, kp, u.nsplit, UHT, ldo, (int)m, k, (int)m, k, (float*)((char*)a.out + pbytes), st);
}

int dnmf_mu_kl_step_bf16x6(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps,
                           int w_update, int clamp, void* ws, size_t ws_bytes, void* stream) {
    if (!klx_shape(A, m, n, lda, k)) return dnmf_mu_kl_step(A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, clamp, ws, ws_bytes, stream);
    REQUIRE(W && H && ws, "mu_kl_step_bf16x6: bad arguments");
    const size_t total = dnmf_ws_bytes_bf16x6(m, n, k);
    if (ws_bytes < total) return fail(DNMF_EWS, "mu_kl_step_bf16x6: workspace %zu < %zu", ws_bytes, total);
    size_t off[5];
    dnmf_ws_offsets_(m, n, k, off);
    char* base = (char*)ws;
    float* Sb = (float*)(base + off[1]);
    float* x = (float*)(base + off[2]);
    void* part = base + off[3];
    const size_t part_bytes = off[4] - off[3];
    char* img = base + align256(off[4]);
    const size_t img_bytes = total - align256(off[4]);
    int rc;
    if (w_update) {                                                                   // dist_nmf.py:813-830
        const long ldu = round_up(k, 4);
        if ((rc = dnmf_rowsum(H, k, n, ldh, x, stream))) return rc;
        if ((rc = dnmf_kl_uht_bf16x6(A, m, n, lda, W, ldw, H, ldh, k, eps, Sb, ldu, img, img_bytes, stream))) return rc;
        if ((rc = dnmf_kl_update_w(W, m, k, ldw, Sb, ldu, x, eps, stream))) return rc;
    }
    const long ldo = round_up(n, 4);                                                  // dist_nmf.py:832-849
    if ((rc = dnmf_colsum(W, m, k, ldw, x, part, part_bytes, stream))) return rc;
    if ((rc = dnmf_kl_wtu_bf16x6(A, m, n, lda, W, ldw, H, ldh, k, eps, Sb, ldo, img, img_bytes, stream))) return rc;
    if ((rc = dnmf_kl_update_h(H, k, n, ldh, Sb, ldo, x, eps, clamp, stream))) return rc;
    if (clamp) return dnmf_clamp_min(W, m, k, ldw, eps, stream);
    return DNMF_OK;
}

}  // extern "C"
