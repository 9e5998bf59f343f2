// dnmf_kl16.hip -- launchers of the 16-wide KL kernels (csrc/dnmf_kl16.h).  A translation unit of its own because it is
// compiled with -mllvm -amdgpu-mfma-vgpr-form (pydnmfk_amd/build.py): MFMA accumulators in VGPRs, so that the division
// between the two products reads and writes them in place.  The other units keep hipcc's heuristic (their k = 128 kernels
// need the AGPR half of the register file).
#include "dnmf_common.h"
#include "dnmf_host.h"
#include "dnmf_kl16.h"

// Library-internal (called from csrc/dnmf_kl.hip).  Return 1 when the 16-wide kernel does not apply (the caller goes on to
// the 32-wide kernels with its arguments untouched), else the launch status.
__attribute__((visibility("hidden"))) int dnmf_kl16_uht_(const float* A, long m, long n, long lda, const float* W, long ldw,
                                                         const float* H, long ldh, long hblk, int k, float eps, float* UHT,
                                                         long ldo, void* ws, size_t ws_bytes, void* stream);
__attribute__((visibility("hidden"))) int dnmf_kl16_wtu_(const float* A, long m, long n, long lda, const float* W, long ldw,
                                                         const float* H, long ldh, int k, float eps, float* WTU, long ldo,
                                                         void* ws, size_t ws_bytes, void* stream);

static bool friendly16(const float* W, long ldw, const float* H, long ldh, int k) {
    return k == 16 && aligned16(W) && ldw % 4 == 0 && aligned16(H) && ldh % 4 == 0;
}
constexpr long WINDOW = 0x7fffffffL;          // the kernels address a tile / a chunk through one 2 GiB buffer descriptor

// hblk > 0: H is the stack of n / hblk column blocks [q][k][hblk] (ldh = hblk); taken only with k = 16 and aligned factors
// (other ranks go to the caller's block-aware padding and come back as one matrix)
int dnmf_kl16_uht_(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, long hblk, int k,
                   float eps, float* UHT, long ldo, void* ws, size_t ws_bytes, void* stream) {
    if (!(k <= 16 && k16_on() && aligned16(A) && lda % 4 == 0 && n % BK == 0 && m <= 0x7fffffffL)) return 1;
    if (hblk && !friendly16(W, ldw, H, ldh, k)) return 1;
    const long ldh_img = friendly16(W, ldw, H, ldh, k) ? ldh : round_up(n, 4);
    if (128 * lda * 4 + n * 4 >= WINDOW || 16 * ldh_img * 4 + n * 4 >= WINDOW) return 1;
    hipStream_t st = S(stream);
    // column splits: rowtiles x nsplit workgroups, at most ONE resident round (4 workgroups per CU by registers and LDS) --
    // a second, partly filled round would run at a fraction of the memory parallelism (see plan_wtu16)
    UhtPlan u;
    {
        const long rowtiles = cdiv(m, 128);
        long ns = std::max<long>(1, 1024 / rowtiles);
        ns = std::min<long>(ns, std::max<long>(1, n / 256));
        u.cols_per_split = round_up(cdiv(n, ns), BK);
        u.nsplit = (int)cdiv(n, u.cols_per_split);
        if (hblk) {                                        // a column split must not straddle a block of H
            const long nb = n / hblk;
            long s = std::max<long>(1, ns / nb);
            while (s > 1 && !(hblk % s == 0 && (hblk / s) % BK == 0)) --s;
            u.cols_per_split = hblk / s;
            u.nsplit = (int)(nb * s);
        }
    }
    const int k_out = k;
    const bool direct = u.nsplit == 1 && k == 16 && aligned16(UHT) && ldo % 4 == 0;
    const size_t pb16 = direct ? 0 : (size_t)u.nsplit * m * 16 * sizeof(float);
    const size_t need16 = pb16 + reduce_scratch_bytes(u.nsplit, (int)m, k);
    if (!direct && !(ws && ws_bytes >= need16)) return 1;
    if (!friendly16(W, ldw, H, ldh, k) && !pad_factors(W, ldw, H, ldh, k, m, n, 16, ws, ws_bytes, need16, st)) return 1;
    Kl16Args a{};
    a.A = A; a.lda = lda; a.m = m; a.n = n; a.W = W; a.ldw = ldw; a.H = H; a.ldh = ldh; a.eps = eps;
    a.P = direct ? UHT : (float*)ws; a.ldp = direct ? ldo : 16; a.chunk_stride = direct ? 0 : m * 16;
    a.cols_per_split = u.cols_per_split;
    a.hblk = hblk; a.hextra = hblk ? (long)16 * hblk - hblk : 0;
    DNMF_LAUNCH(kl_uht16_kernel, dim3((unsigned)cdiv(m, 128), (unsigned)u.nsplit), dim3(256), kl_uht16_lds_bytes(), st, a);
    int rc = check_launch("kl_uht16");
    if (rc || direct) return rc;
    return launch_reduce((const float*)ws, m * 16, 16, u.nsplit, UHT, ldo, (int)m, k_out, (int)m, k_out,
                         (float*)((char*)ws + pb16), st);
}

int dnmf_kl16_wtu_(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k,
                   float eps, float* WTU, long ldo, void* ws, size_t ws_bytes, void* stream) {
    if (!(k <= 16 && k16_on() && aligned16(A) && lda % 4 == 0 && n % 64 == 0)) return 1;
    hipStream_t st = S(stream);
    const Tn16Plan q = plan_wtu16(m, n);
    const long ldw_img = friendly16(W, ldw, H, ldh, k) ? ldw : 16;
    if ((q.rows_per_chunk + 16) * lda * 4 + 1024 >= WINDOW || (q.rows_per_chunk + 16) * ldw_img * 4 >= WINDOW) return 1;
    const int k_out = k;
    const size_t pb16 = (size_t)q.nchunks * 16 * n * sizeof(float);
    const size_t need16 = pb16 + reduce_scratch_bytes(q.nchunks, k, n);
    if (!ws || ws_bytes < need16) return 1;
    if (!friendly16(W, ldw, H, ldh, k) && !pad_factors(W, ldw, H, ldh, k, m, n, 16, ws, ws_bytes, need16, st)) return 1;
    Kl16Args a{};
    a.A = A; a.lda = lda; a.m = m; a.n = n; a.W = W; a.ldw = ldw; a.H = H; a.ldh = ldh; a.eps = eps;
    a.P = (float*)ws; a.ldp = n; a.chunk_stride = 16 * n;
    a.rows_per_chunk = q.rows_per_chunk; a.nchunks = q.nchunks; a.ncolblk = q.ncolblk;
    DNMF_LAUNCH(kl_wtu16_kernel, dim3((unsigned)cdiv((long)q.nchunks * q.ncolblk, 4)), dim3(256), 0, st, a);
    int rc = check_launch("kl_wtu16");
    if (rc) return rc;
    return launch_reduce((const float*)ws, 16 * n, n, q.nchunks, WTU, ldo, k_out, n, k_out, n, (float*)((char*)ws + pb16), st);
}
