// dnmf_split.hip -- C ABI of the bf16x6 contractions (kernels and arithmetic: dnmf_split.h).  Second translation unit of
// libdnmf_hip.so; everything else of a step (Gram matrices, the H update, clamps) is the fp32 code of dnmf.hip.
#include "dnmf_common.h"
#include "dnmf_host.h"
#include "dnmf_split.h"

// workspace layout of the fused steps, owned by dnmf.hip: {g_off, s_off, x_off, part_off, total}
__attribute__((visibility("hidden"))) void dnmf_ws_offsets_(long m, long n, int k, size_t out[5]);

namespace {

// The split kernels exist for 32 < k <= 64 (below that the fp32 kernels are already bound by the HBM, not by the matrix
// cores) and take A in whole 128-byte lines: 16-byte aligned rows, n a multiple of 128.  Every other shape runs the fp32
// kernels of dnmf.hip -- the caller gets the exact products instead.
bool split_shape(const float* A, long m, long n, long lda, int k) {
    return k > 32 && k <= 64 && n % 128 == 0 && lda % 4 == 0 && aligned16(A) && lda >= n && lda <= 4 * n && m >= 1 &&
           n < (1L << 19) && tune("DNMF_SPLIT", 1) != 0;
}

struct TnxPlan { int ncolblk, nchunks; long rows_per_chunk, ldp, chunk_stride; };

TnxPlan plan_tnx(long m, long n) {
    TnxPlan p;
    p.ncolblk = (int)(n / 128);
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
    p.chunk_stride = 64 * n;
    return p;
}

size_t h_image_bytes(long n) { return align256((size_t)3 * 64 * round_up(n, XK) * sizeof(bf16_t)); }
size_t w_image_bytes(long m) { return align256((size_t)3 * 64 * round_up(m, XK) * sizeof(bf16_t)); }

size_t wta_need(long m, long n, int k) {
    const TnxPlan p = plan_tnx(m, n);
    return w_image_bytes(m) + align256((size_t)p.nchunks * p.chunk_stride * sizeof(float)) + reduce_scratch_bytes(p.nchunks, k, n);
}

int cut_h(const float* H, int k, long n, long ldh, bf16_t* img, SplitOperand& o, hipStream_t st) {
    o.ld = round_up(n, XK);
    o.split_stride = 64 * o.ld;
    o.S = img;
    const long threads = 64 * (o.ld / 8);
    hipLaunchKernelGGL(split3_rows_kernel, dim3((unsigned)cdiv(threads, 256)), dim3(256), 0, st, H, ldh, k, n, img, o.ld,
                       o.split_stride, 64);
    return check_launch("split3_rows");
}

int cut_wt(const float* W, long m, int k, long ldw, bf16_t* img, SplitOperand& o, hipStream_t st) {
    o.ld = round_up(m, XK);
    o.split_stride = 64 * o.ld;
    o.S = img;
    hipLaunchKernelGGL(split3_cols_kernel, dim3((unsigned)(o.ld / 64), 2), dim3(256), 0, st, W, ldw, m, k, img, o.ld, o.split_stride);
    return check_launch("split3_cols");
}

template <int MODE>
int launch_ntx(const NtArgs& a, const SplitOperand& ys, hipStream_t st) {
    constexpr size_t lds = 2 * (128 * 32 * 4 + 3 * 64 * 64);      // two stages [A tile fp32 | H tile bf16 pieces]; >= the W.G loop's 49152 B
    static bool once = false;
    // A is touched once: stream it past the caches when it cannot stay in them anyway
    const bool nt = (double)a.nrows * a.ncols * 4 >= 256.0 * (1 << 20) && tune("DNMF_SPLIT_NT", 1) != 0;
    static const int nset = (int)tune("DNMF_SPLIT_NSET", 4);
    if (!once) {
        allow_lds(ntx_kernel<2, MODE, 0, 4>, lds); allow_lds(ntx_kernel<2, MODE, 2, 4>, lds);
        allow_lds(ntx_kernel<2, MODE, 0, 2>, lds); allow_lds(ntx_kernel<2, MODE, 2, 2>, lds);
        once = true;
    }
    const dim3 grid((unsigned)cdiv(a.nrows, 128), 1);
    if (nset == 2) {
        if (nt) hipLaunchKernelGGL((ntx_kernel<2, MODE, 2, 2>), grid, dim3(256), lds, st, a, ys);
        else hipLaunchKernelGGL((ntx_kernel<2, MODE, 0, 2>), grid, dim3(256), lds, st, a, ys);
    } else {
        if (nt) hipLaunchKernelGGL((ntx_kernel<2, MODE, 2, 4>), grid, dim3(256), lds, st, a, ys);
        else hipLaunchKernelGGL((ntx_kernel<2, MODE, 0, 4>), grid, dim3(256), lds, st, a, ys);
    }
    return check_launch("ntx_kernel");
}

}  // namespace

extern "C" {

size_t dnmf_ws_bytes_bf16x6(long m, long n, int k) {
    const size_t base = dnmf_ws_bytes(m, n, k);
    if (!base) return 0;
    if (!(k > 32 && k <= 64 && n % 128 == 0)) return base;
    return align256(base) + h_image_bytes(n) + wta_need(m, n, k);
}

int dnmf_aht_update_w_bf16x6(const float* A, long m, long n, long lda, const float* H, int k, long ldh, const float* G,
                             float* W, long ldw, float eps, void* ws, size_t ws_bytes, void* stream) {
    if (!split_shape(A, m, n, lda, k)) return dnmf_aht_update_w(A, m, n, lda, H, k, ldh, G, W, ldw, eps, stream);
    REQUIRE(H && G && W && ws && ldh >= n && ldw >= k, "aht_update_w_bf16x6: bad arguments");
    if (ws_bytes < h_image_bytes(n)) return fail(DNMF_EWS, "aht_update_w_bf16x6: workspace %zu < %zu", ws_bytes, h_image_bytes(n));
    hipStream_t st = S(stream);
    SplitOperand ys;
    int rc = cut_h(H, k, n, ldh, (bf16_t*)ws, ys, st);
    if (rc) return rc;
    NtArgs a{};
    a.X = A; a.ldx = lda; a.nrows = m; a.ncols = n; a.Y = H; a.ldy = ldh; a.yrows = k; a.cols_per_split = n;
    a.W = W; a.ldw = ldw; a.G = G; a.eps = eps; a.k = k;
    a.wfast = aligned16(W) && ldw % 4 == 0 && k % 4 == 0;
    return launch_ntx<NT_FUSED_W>(a, ys, st);
}

int dnmf_aht_bf16x6(const float* A, long m, long n, long lda, const float* H, int k, long ldh, float* AH, long ldah,
                    void* ws, size_t ws_bytes, void* stream) {
    if (!split_shape(A, m, n, lda, k)) return dnmf_aht(A, m, n, lda, H, k, ldh, AH, ldah, stream);
    REQUIRE(H && AH && ws && ldh >= n && ldah >= k, "aht_bf16x6: bad arguments");
    if (ws_bytes < h_image_bytes(n)) return fail(DNMF_EWS, "aht_bf16x6: workspace %zu < %zu", ws_bytes, h_image_bytes(n));
    hipStream_t st = S(stream);
    SplitOperand ys;
    int rc = cut_h(H, k, n, ldh, (bf16_t*)ws, ys, st);
    if (rc) return rc;
    NtArgs a{};
    a.X = A; a.ldx = lda; a.nrows = m; a.ncols = n; a.Y = H; a.ldy = ldh; a.yrows = k; a.cols_per_split = n;
    a.out = AH; a.ldo = ldah; a.split_stride = 0; a.store_all = 0;
    return launch_ntx<NT_STORE>(a, ys, st);
}

int dnmf_wta_bf16x6(const float* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw,
                    void* ws, size_t ws_bytes, void* stream) {
    if (!split_shape(A, m, n, lda, k)) return dnmf_wta(A, m, n, lda, W, k, ldw, AtW, ldatw, ws, ws_bytes, stream);
    REQUIRE(W && AtW && ws && ldw >= k && ldatw >= n, "wta_bf16x6: bad arguments");
    const size_t need = wta_need(m, n, k);
    if (ws_bytes < need) return fail(DNMF_EWS, "wta_bf16x6: workspace %zu < %zu", ws_bytes, need);
    hipStream_t st = S(stream);
    SplitOperand wsplit;
    int rc = cut_wt(W, m, k, ldw, (bf16_t*)ws, wsplit, st);
    if (rc) return rc;
    const TnxPlan p = plan_tnx(m, n);
    float* P = (float*)((char*)ws + w_image_bytes(m));
    const size_t pbytes = align256((size_t)p.nchunks * p.chunk_stride * sizeof(float));
    TnArgs a{};
    a.X = W; a.ldx = ldw; a.xcols = k; a.Y = A; a.ldy = lda; a.ycols = n;
    a.nrows = m; a.rows_per_chunk = p.rows_per_chunk; a.nchunks = p.nchunks; a.ncolblk = p.ncolblk;
    a.P = P; a.chunk_stride = p.chunk_stride; a.ldp = p.ldp;
    constexpr size_t lds = 2 * 3 * 64 * 128;
    static bool once = false;
    if (!once) { allow_lds(tnx_kernel<2, 0>, lds); allow_lds(tnx_kernel<2, 2>, lds); once = true; }
    const bool nt = (double)m * n * 4 >= 256.0 * (1 << 20) && tune("DNMF_SPLIT_NT", 1) != 0;
    const dim3 grid((unsigned)(cdiv(p.ncolblk, 4) * p.nchunks));
    if (nt) hipLaunchKernelGGL((tnx_kernel<2, 2>), grid, dim3(256), lds, st, a, wsplit);
    else hipLaunchKernelGGL((tnx_kernel<2, 0>), grid, dim3(256), lds, st, a, wsplit);
    if ((rc = check_launch("tnx_kernel"))) return rc;
    return launch_reduce(P, p.chunk_stride, p.ldp, p.nchunks, AtW, ldatw, k, n, k, n, (float*)((char*)P + pbytes), st);
}

int dnmf_mu_fro_step_bf16x6(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k,
                            float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream) {
    if (!split_shape(A, m, n, lda, k)) return dnmf_mu_fro_step(A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, clamp, ws, ws_bytes, stream);
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
        if ((rc = dnmf_aht_update_w_bf16x6(A, m, n, lda, H, k, ldh, G, W, ldw, eps, img, img_bytes, stream))) return rc;
    }
    const long ldatw = round_up(n, 4);                                                // dist_nmf.py:736-751
    if ((rc = dnmf_gram_wtw(W, m, k, ldw, G, part, part_bytes, stream))) return rc;
    if ((rc = dnmf_wta_bf16x6(A, m, n, lda, W, k, ldw, Sb, ldatw, img + h_image_bytes(n), img_bytes - h_image_bytes(n), stream))) return rc;
    if ((rc = dnmf_mu_update_h(H, k, n, ldh, Sb, ldatw, G, eps, clamp, stream))) return rc;
    if (clamp) return dnmf_clamp_min(W, m, k, ldw, eps, stream);                      // pyDNMF.py:155-157
    return DNMF_OK;
}

}  // extern "C"
