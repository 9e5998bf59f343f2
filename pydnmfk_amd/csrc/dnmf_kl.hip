// dnmf_kl.hip -- C ABI of the NN-form kernels (residual / per-column error, the two KL products: csrc/dnmf_nn.h).  A translation
// unit of its own: this kernel family compiles as long as the rest together, so it builds side by side with csrc/dnmf.hip.
#include "dnmf_common.h"
#include "dnmf_host.h"
#include "dnmf_nt.h"
#include "dnmf_stream.h"
#include "dnmf_nn.h"

// the 16-wide kernels for k <= 16 live in csrc/dnmf_kl16.hip (own compiler flags); 1 = not applicable
__attribute__((visibility("hidden"))) int dnmf_kl16_uht_(const float* A, long m, long n, long lda, const float* W, long ldw,
                                                         const float* H, long ldh, long hblk, int k, float eps, float* UHT,
                                                         long ldo, void* ws, size_t ws_bytes, void* stream);
__attribute__((visibility("hidden"))) int dnmf_kl16_wtu_(const float* A, long m, long n, long lda, const float* W, long ldw,
                                                         const float* H, long ldh, int k, float eps, float* WTU, long ldo,
                                                         void* ws, size_t ws_bytes, void* stream);

// the software-pipelined U H^T for whole 128-row tiles (csrc/dnmf_kluht.hip)
__attribute__((visibility("hidden"))) int dnmf_kl_uht_pipe_(const float* A, long rowtiles, long n, long lda, const float* W, long ldw,
                                                            const float* H, long ldh, long hblk, long hextra, int kt, float eps,
                                                            float* out, long ldo, long split_stride, long cols_per_split,
                                                            int nsplit, void* stream);

// csrc/dnmf_wide.hip: ranks 128 < k <= 256
#define HID __attribute__((visibility("hidden")))
HID int dnmf_wide_quot_(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k, float eps, float* U,
                        long ldu, void* stream);
HID int dnmf_wide_resid_(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k, double* out,
                         void* stream);
HID int dnmf_wide_column_err_(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k, double* num,
                              double* den, void* stream);
#undef HID

namespace {
// The KL products of a wide rank: U = A / (W H + eps) materialised at the END of the caller's workspace (dnmf_ws_bytes reserves the
// image for k > 128), then the tuned contraction on U.  uht: U H^T; else W^T U.
int wide_kl_product(bool uht, const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k, float eps,
                    float* out, long ldo, void* ws, size_t ws_bytes, void* stream) {
    const long ldu = round_up(n, 4);
    const size_t ub = align256((size_t)m * ldu * sizeof(float));
    if (!ws || ws_bytes < ub) return fail(DNMF_EWS, "kl product (k = %d): workspace %zu < %zu (the quotient image)", k, ws_bytes, ub);
    const size_t inner = (ws_bytes - ub) & ~size_t(255);
    float* U = (float*)((char*)ws + inner);
    if (int rc = dnmf_wide_quot_(A, m, n, lda, W, ldw, H, ldh, k, eps, U, ldu, stream)) return rc;
    return uht ? dnmf_aht(U, m, n, ldu, H, k, ldh, out, ldo, stream) : dnmf_wta(U, m, n, ldu, W, k, ldw, out, ldo, ws, inner, stream);
}
}  // namespace

extern "C" {

static NnArgs nn_args(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                      int k, float eps) {
    NnArgs a{};
    a.A = A; a.lda = lda; a.m = m; a.n = n; a.W = W; a.ldw = ldw; a.H = H; a.ldh = ldh; a.k = k; a.eps = eps;
    a.nrowblk = cdiv(m, 32); a.ncolblk = (int)cdiv(n, 128);
    a.kreal = k;
    static const int kl_pipe = (int)tune("DNMF_KL_PIPE", 1);
    a.pipe = kl_pipe;
    return a;
}

static bool nn_fast(const float* A, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k) {
    return aligned16(A) && aligned16(W) && aligned16(H) && lda % 4 == 0 && n % 4 == 0 && ldw % 4 == 0 && k % 4 == 0 &&
           ldh % 4 == 0;
}

}  // extern "C"
namespace {
// ws (optional): with room for the zero-padded factor images (pad_bytes: every dnmf_ws_bytes workspace has it) a rank that is
// not a whole number of 32-wide tiles -- every k an NMFk sweep visits -- runs the LDS-staged kernel on [m x KP] / [KP x n]
// images instead of the predicated one-tile-per-wave kernel (round 4: 0.645 -> ~0.39 ms at 32768 x 16384, k = 16; zero
// columns of W / rows of H add nothing to W H).
template <typename TA>
int resid_sqnorm_impl(const TA* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                      int k, double* out, void* stream, void* ws = nullptr, size_t ws_bytes = 0) {
    if (wide_k(k)) {
        REQUIRE((std::is_same<TA, float>::value) && A && W && H && out && m >= 1 && n >= 1 && lda >= n && ldw >= k && ldh >= n,
                "resid_sqnorm: bad arguments (k = %d: float32 data)", k);
        return dnmf_wide_resid_(reinterpret_cast<const float*>(A), m, n, lda, W, ldw, H, ldh, k, out, stream);
    }
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && out && m >= 1 && n >= 1 && lda >= n && ldw >= k && ldh >= n, "resid_sqnorm: bad arguments");
    hipStream_t st = S(stream);
    if (batch_memset(out, 0, sizeof(double), st) != hipSuccess) return fail(DNMF_EHIP, "resid_sqnorm: memset failed");
    if (ws && a_aligned(A) && lda % 4 == 0 && n % 4 == 0 && n >= 128 && m >= 4096)
        pad_factors(W, ldw, H, ldh, k, m, n, 32 * kt, ws, ws_bytes, 0, st);      // (no-op for friendly factors)
    NnArgs a = nn_args(reinterpret_cast<const float*>(A), m, n, lda, W, ldw, H, ldh, k, 0.f);
    a.out = out;
    const bool fast = a_aligned(A) && nn_fast(W, n, lda, W, ldw, H, ldh, k);
    hipStream_t stq = st;
    // H staged through LDS, waves walking row chunks (resid_lds_kernel) when the block is big enough to give every wave a
    // chunk of several row blocks; small blocks keep one tile per wave (resid_kernel)
    static const int lds_on = (int)tune("DNMF_RESID_LDS", 1);
    if (lds_on && fast && k % 32 == 0 && n >= 128 && m >= 4096) {
        const int nt = 4;
        TnPlan pl = plan_tn(m, n, kt, nt);
        {                                                          // one wave per SIMD (1024 waves): half of plan_tn's round
            const long nch = std::max<long>(1, pl.nchunks / 2);
            pl.rows_per_chunk = round_up(cdiv(m, nch), 32);
        }
        const long rpc = std::max<long>(1, round_up(pl.rows_per_chunk, 32) / 32);
        const long nchunks = cdiv(a.nrowblk, rpc);
        a.ncolblk = (int)cdiv(n, 128);
        const dim3 grid2((unsigned)(cdiv(nchunks, 4) * a.ncolblk)), block2(256);
        const size_t lds = (size_t)(32 * kt) * 128 * sizeof(float);
#define RL_CASE(KT_)                                                                                 \
        if (kt == KT_) {                                                                             \
            static bool once = false;                                                                \
            if (!once) { allow_lds(resid_lds_kernel<KT_, true, TA>, lds); once = true; }              \
            DNMF_LAUNCH((resid_lds_kernel<KT_, true, TA>), grid2, block2, lds, stq, a, rpc);  \
        }
        RL_CASE(1) RL_CASE(2) RL_CASE(4)
#undef RL_CASE
        return check_launch("resid_sqnorm(lds)");
    }
    const dim3 grid((unsigned)cdiv(a.nrowblk * a.ncolblk, 4)), block(256);
#define RS_CASE(KT_)                                                                   \
    if (kt == KT_) {                                                                   \
        if (fast) DNMF_LAUNCH((resid_kernel<KT_, true, TA>), grid, block, 0, st, a); \
        else DNMF_LAUNCH((resid_kernel<KT_, false, TA>), grid, block, 0, st, a);    \
    }
    RS_CASE(1) RS_CASE(2) RS_CASE(4)
#undef RS_CASE
    return check_launch("resid_sqnorm");
}
}  // namespace
extern "C" {

int dnmf_resid_sqnorm(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                      int k, double* out, void* stream) {
    return resid_sqnorm_impl<float>(A, m, n, lda, W, ldw, H, ldh, k, out, stream);
}
int dnmf_resid_sqnorm_bf16a(const void* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                            int k, double* out, void* stream) {
    return resid_sqnorm_impl<bf16_t>((const bf16_t*)A, m, n, lda, W, ldw, H, ldh, k, out, stream);
}
int dnmf_resid_sqnorm_ws(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                         int k, double* out, void* ws, size_t ws_bytes, void* stream) {
    return resid_sqnorm_impl<float>(A, m, n, lda, W, ldw, H, ldh, k, out, stream, ws, ws_bytes);
}
int dnmf_resid_sqnorm_ws_bf16a(const void* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                               int k, double* out, void* ws, size_t ws_bytes, void* stream) {
    return resid_sqnorm_impl<bf16_t>((const bf16_t*)A, m, n, lda, W, ldw, H, ldh, k, out, stream, ws, ws_bytes);
}

}  // extern "C"
namespace {
template <typename TA>
int column_err_impl(const TA* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k,
                    double* num, double* den, void* stream) {
    if (wide_k(k)) {
        REQUIRE((std::is_same<TA, float>::value) && A && W && H && num && den && m >= 1 && n >= 1 && lda >= n && ldw >= k && ldh >= n,
                "column_err: bad arguments (k = %d: float32 data)", k);
        return dnmf_wide_column_err_(reinterpret_cast<const float*>(A), m, n, lda, W, ldw, H, ldh, k, num, den, stream);
    }
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && num && den && m >= 1 && n >= 1 && lda >= n && ldw >= k && ldh >= n, "column_err: bad arguments");
    hipStream_t st = S(stream);
    NnArgs a = nn_args(reinterpret_cast<const float*>(A), m, n, lda, W, ldw, H, ldh, k, 0.f);
    const bool fast = a_aligned(A) && nn_fast(W, n, lda, W, ldw, H, ldh, k);
    // about 8 waves per SIMD in flight, every wave walking a chunk of row blocks of its 128-column block
    const long rpc = std::max<long>(1, cdiv(a.nrowblk * a.ncolblk, 8192));
    const long waves = cdiv(a.nrowblk, rpc) * a.ncolblk;
    const dim3 grid((unsigned)cdiv(waves, 4)), block(256);
#define CE_CASE(KT_)                                                                             \
    if (kt == KT_) {                                                                             \
        if (fast) DNMF_LAUNCH((colerr_kernel<KT_, true, TA>), grid, block, 0, st, a, num, den, rpc); \
        else DNMF_LAUNCH((colerr_kernel<KT_, false, TA>), grid, block, 0, st, a, num, den, rpc);    \
    }
    CE_CASE(1) CE_CASE(2) CE_CASE(4)
#undef CE_CASE
    return check_launch("column_err");
}
}  // namespace
extern "C" {

int dnmf_column_err(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k,
                    double* num, double* den, void* stream) {
    return column_err_impl<float>(A, m, n, lda, W, ldw, H, ldh, k, num, den, stream);
}
int dnmf_column_err_bf16a(const void* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k,
                          double* num, double* den, void* stream) {
    return column_err_impl<bf16_t>((const bf16_t*)A, m, n, lda, W, ldw, H, ldh, k, num, den, stream);
}

}  // extern "C"
namespace {

// H as column blocks [n / hblk][k][hblk] -> one zero-padded contiguous image [kp x round_up(n, 4)] (and W -> [m x kp]) at the
// end of the workspace: the block-aware twin of pad_factors for ranks / alignments the interior paths do not take
bool pad_factors_hblocks(const float*& W, long& ldw, const float*& H, long& ldh, long hblk, int& k, long m, long n, int kp,
                         void* ws, size_t ws_bytes, size_t own_need, hipStream_t st) {
    const size_t pb = pad_bytes(m, n, kp);
    if (!ws || ws_bytes < align256(own_need) + pb) return false;
    char* base = (char*)ws + align256(own_need);
    float* Wp = (float*)base;
    const long ldhp = round_up(n, 4);
    float* Hp = (float*)(base + align256((size_t)m * kp * sizeof(float)));
    if (hipMemsetAsync(Wp, 0, (size_t)m * kp * sizeof(float), st) != hipSuccess) return false;
    if (hipMemcpy2DAsync(Wp, (size_t)kp * sizeof(float), W, (size_t)ldw * sizeof(float), (size_t)k * sizeof(float), (size_t)m,
                         hipMemcpyDeviceToDevice, st) != hipSuccess) return false;
    if (hipMemsetAsync(Hp, 0, (size_t)kp * ldhp * sizeof(float), st) != hipSuccess) return false;
    for (long q = 0; q * hblk < n; ++q)
        if (hipMemcpy2DAsync(Hp + q * hblk, (size_t)ldhp * sizeof(float), H + q * k * hblk, (size_t)hblk * sizeof(float),
                             (size_t)hblk * sizeof(float), (size_t)k, hipMemcpyDeviceToDevice, st) != hipSuccess) return false;
    W = Wp; ldw = kp; H = Hp; ldh = ldhp; k = kp;
    return true;
}

// hblk = 0: H is one k x n matrix (ldh).  hblk > 0: H is the stack of n / hblk column blocks [q][k][hblk] (ldh = hblk).
int kl_uht_impl(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, long hblk, int k,
                float eps, float* UHT, long ldo, void* ws, size_t ws_bytes, void* stream) {   // (W, ldw, H, ldh, k may be re-pointed at padded copies)
    if (wide_k(k)) {
        REQUIRE(!hblk && A && W && H && UHT && m >= 1 && n >= 1 && lda >= n && ldw >= k && ldh >= n && ldo >= k,
                "kl_uht: bad arguments (k = %d; H as column blocks: k <= %d)", k, DNMF_TUNED_MAX_K);
        return wide_kl_product(true, A, m, n, lda, W, ldw, H, ldh, k, eps, UHT, ldo, ws, ws_bytes, stream);
    }
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && UHT && m >= 1 && n >= 1 && lda >= n && ldw >= k && ldo >= k, "kl_uht: bad arguments");
    REQUIRE(hblk ? (ldh == hblk && n % hblk == 0 && hblk % BK == 0) : ldh >= n, "kl_uht: bad H layout (ldh %ld, block %ld, n %ld)", ldh, hblk, n);
    const int kp = 32 * kt;
    const int k_out = k;                                   // columns of UHT the caller gets
    if (int rc16 = dnmf_kl16_uht_(A, m, n, lda, W, ldw, H, ldh, hblk, k, eps, UHT, ldo, ws, ws_bytes, stream); rc16 != 1) return rc16;
    UhtPlan u = plan_uht(m, n, kt);
    if (hblk) {                                            // a column split must not straddle a block: cols_per_split divides hblk
        const long nb = n / hblk;
        long s = std::max<long>(1, (u.nsplit + nb / 2) / nb);
        auto fits = [&](long sp) { return hblk % sp == 0 && (hblk / sp) % BK == 0 &&
                                          (size_t)(nb * sp) * m * kp * sizeof(float) + reduce_scratch_bytes((int)(nb * sp), (int)m, k) <= ws_bytes; };
        while (s > 1 && !fits(s)) --s;
        if (!fits(s) && nb * s > 1) return fail(DNMF_EWS, "kl_uht: workspace %zu too small for %ld column blocks", ws_bytes, nb);
        u.cols_per_split = hblk / s;
        u.nsplit = (int)(nb * s);
    }
    const size_t pbytes = u.nsplit > 1 ? (size_t)u.nsplit * m * kp * sizeof(float) : 0;
    const size_t need = pbytes + reduce_scratch_bytes(u.nsplit, (int)m, k);
    if (u.nsplit > 1 && (!ws || ws_bytes < need)) return fail(DNMF_EWS, "kl_uht: workspace %zu < %zu", ws_bytes, need);
    if (aligned16(A) && lda % 4 == 0 && n % 4 == 0) {
        if (!hblk) pad_factors(W, ldw, H, ldh, k, m, n, kp, ws, ws_bytes, need, S(stream));
        else if (!(k == kp && aligned16(W) && ldw % 4 == 0 && aligned16(H)) &&
                 pad_factors_hblocks(W, ldw, H, ldh, hblk, k, m, n, kp, ws, ws_bytes, need, S(stream))) hblk = 0;
    }
    NnArgs a = nn_args(A, m, n, lda, W, ldw, H, ldh, k, eps);
    a.kreal = k_out;                                       // (k may be the padded rank by now)
    a.hblk = hblk; a.hextra = hblk ? (long)k * hblk - hblk : 0;
    const bool split = u.nsplit > 1;
    float* out = split ? (float*)ws : UHT;
    const long ldout = split ? kp : ldo;
    const int out_cols = split ? kp : k_out;
    const bool fast = nn_fast(A, n, lda, W, ldw, H, ldh, k) && aligned16(out) && ldout % 4 == 0;
    hipStream_t st = S(stream);
    // Whole 128-row tiles of a friendly problem (aligned rows, k = KP or padded to it, whole 32-column tiles, 2 GiB descriptor
    // windows) go to the software-pipelined kernel; a ragged last row tile -- or everything else -- to kl_uht_kernel.  Same
    // arithmetic in the same order: the two are bit identical.
    static const int pipe_on = (int)tune("DNMF_KLUHT_PIPE", 1);
    auto window = [](long rows, long ld, long cols) { return (rows * ld + cols) * 4 < 0x7fffffffL; };
    long rowtile0 = 0;
    if (pipe_on && fast && k == kp && m >= 128 && n % BK == 0 && u.cols_per_split % BK == 0 && out_cols >= kp &&
        window(128, lda, u.cols_per_split) && window(kp, ldh, u.cols_per_split)) {
        rowtile0 = m / 128;
        if (int rcp = dnmf_kl_uht_pipe_(A, rowtile0, n, lda, W, ldw, H, ldh, hblk, a.hextra, kt, eps, out, ldout, (long)m * kp,
                                        u.cols_per_split, u.nsplit, stream); rcp) return rcp;
    }
    a.rowtile0 = rowtile0;
    const dim3 grid((unsigned)(cdiv(m, 128) - rowtile0), (unsigned)u.nsplit), block(256);
    const size_t lds = 2ul * kp * BK * sizeof(float);
    if (grid.x > 0) {
#define UH_CASE(KT_)                                                                                                  \
    if (kt == KT_) {                                                                                                  \
        if (fast) DNMF_LAUNCH((kl_uht_kernel<KT_, true>), grid, block, lds, st, a, out, ldout, (long)m * kp,    \
                                     u.cols_per_split, out_cols);                                                     \
        else DNMF_LAUNCH((kl_uht_kernel<KT_, false>), grid, block, lds, st, a, out, ldout, (long)m * kp,        \
                                u.cols_per_split, out_cols);                                                          \
    }
    UH_CASE(1) UH_CASE(2) UH_CASE(4)
#undef UH_CASE
    }
    int rc = check_launch("kl_uht");
    if (rc || !split) return rc;
    return launch_reduce((const float*)ws, (long)m * kp, kp, u.nsplit, UHT, ldo, (int)m, k_out, (int)m, k_out,
                         (float*)((char*)ws + pbytes), st);
}

}  // namespace
extern "C" {

int dnmf_kl_uht(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k,
                float eps, float* UHT, long ldo, void* ws, size_t ws_bytes, void* stream) {
    return kl_uht_impl(A, m, n, lda, W, ldw, H, ldh, 0, k, eps, UHT, ldo, ws, ws_bytes, stream);
}

size_t dnmf_ws_bytes_hblocks(long m, long n, int k, long nh) {
    const int kt = kt_of(k);
    if (kt < 0 || m < 1 || n < 1 || nh < 1 || n % nh) return 0;
    const int kp = 32 * kt;
    // a column split never straddles a block: at least n / nh splits, at most twice the planner's count
    const UhtPlan u = plan_uht(m, n, kt);
    const long nb = n / nh, nsp = nb * std::max<long>(1, (u.nsplit + nb / 2) / nb);
    const size_t slabs = (size_t)nsp * m * kp * sizeof(float) + reduce_scratch_bytes((int)nsp, (int)m, k);
    return std::max(dnmf_ws_bytes(m, n, k), align256(slabs) + pad_bytes(m, n, kp));
}

int dnmf_kl_uht_hblocks(const float* A, long m, long n, long lda, const float* W, long ldw, const float* Hs, long nh, int k,
                        float eps, float* UHT, long ldo, void* ws, size_t ws_bytes, void* stream) {
    REQUIRE(nh >= 1, "kl_uht_hblocks: bad block width");
    return kl_uht_impl(A, m, n, lda, W, ldw, Hs, nh, nh, k, eps, UHT, ldo, ws, ws_bytes, stream);
}

int dnmf_kl_wtu(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k,
                float eps, float* WTU, long ldo, void* ws, size_t ws_bytes, void* stream) {
    if (wide_k(k)) {
        REQUIRE(A && W && H && WTU && ws && m >= 1 && n >= 1 && lda >= n && ldw >= k && ldh >= n && ldo >= n, "kl_wtu: bad arguments");
        return wide_kl_product(false, A, m, n, lda, W, ldw, H, ldh, k, eps, WTU, ldo, ws, ws_bytes, stream);
    }
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && WTU && ws && m >= 1 && n >= 1 && lda >= n && ldw >= k && ldh >= n && ldo >= n, "kl_wtu: bad arguments");
    const int kp = 32 * kt;
    const KlWtuPlan plan = plan_kl_wtu(m, n, kt);
    const int nt = plan.nt;
    const TnPlan& p = plan.tn;
    const long rowblks_per_chunk = plan.rowblks_per_chunk;
    const long nchunks = plan.nchunks;
    const size_t pbytes = (size_t)nchunks * p.ldp * kp * sizeof(float);
    const size_t need = pbytes + reduce_scratch_bytes((int)nchunks, k, n);
    if (ws_bytes < need) return fail(DNMF_EWS, "kl_wtu: workspace %zu < %zu", ws_bytes, need);
    const int k_out = k;                                   // rows of WTU the caller gets
    if (int rc16 = dnmf_kl16_wtu_(A, m, n, lda, W, ldw, H, ldh, k, eps, WTU, ldo, ws, ws_bytes, stream); rc16 != 1) return rc16;
    if (aligned16(A) && lda % 4 == 0 && n % 4 == 0) pad_factors(W, ldw, H, ldh, k, m, n, kp, ws, ws_bytes, need, S(stream));
    NnArgs a = nn_args(A, m, n, lda, W, ldw, H, ldh, k, eps);
    a.kreal = k_out;
    a.ncolblk = p.ncolblk;
    a.P = (float*)ws; a.ldp = p.ldp; a.chunk_stride = p.ldp * kp;
    const bool fast = nn_fast(A, n, lda, W, ldw, H, ldh, k);
    const dim3 grid((unsigned)(cdiv(nchunks, 4) * a.ncolblk)), block(256);   // 4 row chunks (waves) per workgroup
    const size_t lds = (size_t)kp * 32 * nt * sizeof(float);                   // the H block of the workgroup's columns
    hipStream_t st = S(stream);
#define WU_CASE(KT_, NT_)                                                                                         \
    if (kt == KT_) {                                                                                              \
        if (fast) DNMF_LAUNCH((kl_wtu_kernel<KT_, NT_, true>), grid, block, lds, st, a, rowblks_per_chunk); \
        else DNMF_LAUNCH((kl_wtu_kernel<KT_, NT_, false>), grid, block, lds, st, a, rowblks_per_chunk);    \
    }
#ifdef DNMF_TUNING
    static const long wvar = tune("DNMF_WTU_VAR", 0);          // A/B: 10 * nt + waves per SIMD for kt = 1; 100 + waves per SIMD for kt = 2
#define WV(KT_, NT_, OCC_, VAR_) if (kt == KT_ && nt == NT_ && wvar == VAR_ && fast) { \
        DNMF_LAUNCH((kl_wtu_kernel<KT_, NT_, true, OCC_>), grid, block, lds, st, a, rowblks_per_chunk); } else
    WV(1, 2, 2, 22) WV(1, 2, 3, 23) WV(1, 4, 1, 41) WV(2, 2, 1, 101) WV(2, 2, 2, 102)
#undef WV
#endif
    {
    WU_CASE(1, 4) WU_CASE(2, 2) WU_CASE(4, 2)
    }
#undef WU_CASE
    int rc = check_launch("kl_wtu");
    if (rc) return rc;
    return launch_reduce((const float*)ws, a.chunk_stride, a.ldp, (int)nchunks, WTU, ldo, k_out, n, k_out, n,
                         (float*)((char*)ws + pbytes), st);
}

}  // extern "C"
