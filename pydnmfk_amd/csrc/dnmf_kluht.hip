// dnmf_kluht.hip -- launcher of the software-pipelined KL W-side product (csrc/dnmf_kluht.h).  A translation unit of its own:
// the kernel is cut for a register budget (2 / 3 / 4 waves per SIMD at k = 128 / 64 / 32: hipcc then keeps the MFMA
// accumulators in VGPRs and the division between the two products reads them in place) and compiles side by side with
// csrc/dnmf_kl.hip.
#include "dnmf_common.h"
#include "dnmf_host.h"
#include "dnmf_kluht.h"

// Library-internal (called from csrc/dnmf_kl.hip with arguments it has already validated: 16-byte aligned rows everywhere,
// k == 32 kt, n and cols_per_split whole numbers of 32-column tiles, `rowtiles` full 128-row tiles, descriptor windows).
__attribute__((visibility("hidden"))) int dnmf_kl_uht_pipe_(const float* A, long rowtiles, long n, long lda, const float* W, long ldw,
                                                            const float* H, long ldh, long hblk, long hextra, int kt, float eps,
                                                            float* out, long ldo, long split_stride, long cols_per_split,
                                                            int nsplit, void* stream);

int dnmf_kl_uht_pipe_(const float* A, long rowtiles, long n, long lda, const float* W, long ldw, const float* H, long ldh, long hblk,
                      long hextra, int kt, float eps, float* out, long ldo, long split_stride, long cols_per_split, int nsplit,
                      void* stream) {
    KlUhtArgs a{};
    a.A = A; a.lda = lda; a.n = n; a.W = W; a.ldw = ldw; a.H = H; a.ldh = ldh; a.eps = eps; a.hblk = hblk; a.hextra = hextra;
    a.out = out; a.ldo = ldo; a.split_stride = split_stride; a.cols_per_split = cols_per_split;
    const dim3 grid((unsigned)rowtiles, (unsigned)nsplit), block(256);
    const size_t lds = 2ul * 32 * kt * BK * sizeof(float);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#ifdef DNMF_TUNING
    {   // A/B variants: DNMF_KLUHT_VAR = 100 * waves per SIMD + 10 * A2 + ... , DNMF_KLUHT_ABL = ablation bits (wrong results)
        static const long var = tune("DNMF_KLUHT_VAR", 0), abl = tune("DNMF_KLUHT_ABL", 0);
#define KV(KT_, A2_, OCC_, ABL_) if (kt == KT_ && var == 100 * OCC_ + 10 * A2_ && abl == ABL_) { \
            DNMF_LAUNCH((kl_uht_pipe_kernel<KT_, A2_, OCC_, ABL_>), grid, block, lds, st, a); return check_launch("kl_uht(pipe var)"); }
        KV(1, true, 3, 0) KV(1, false, 4, 0) KV(1, true, 2, 0) KV(2, true, 2, 0) KV(2, false, 3, 0) KV(4, false, 1, 0) KV(4, true, 2, 0)
#define KA(KT_, A2_, OCC_, AUX_) if (kt == KT_ && var == 100 * OCC_ + 10 * A2_ + 1 + AUX_ && abl == 0) { \
            DNMF_LAUNCH((kl_uht_pipe_kernel<KT_, A2_, OCC_, 0, AUX_>), grid, block, lds, st, a); return check_launch("kl_uht(pipe var)"); }
        KA(1, true, 4, 0) KA(2, true, 3, 0) KA(4, false, 2, 0) KA(1, true, 4, 1) KA(2, true, 3, 1) KA(4, false, 2, 1)
#undef KA
        KV(1, true, 4, 1) KV(1, true, 4, 2) KV(1, true, 4, 4) KV(1, true, 4, 8) KV(1, true, 4, 16) KV(1, true, 4, 32) KV(1, true, 4, 12) KV(1, true, 4, 15) KV(1, true, 4, 64) KV(2, true, 3, 64) KV(1, true, 4, 128) KV(2, true, 3, 128) KV(1, true, 4, 136) KV(1, true, 4, 129)
        KV(2, true, 3, 1) KV(2, true, 3, 2) KV(2, true, 3, 4) KV(2, true, 3, 8) KV(2, true, 3, 12) KV(2, true, 3, 15)
#undef KV
    }
#endif
#ifdef DNMF_TUNING
    {   // MR = 2: 64-row wave tiles (256-row workgroups) for k <= 64 -- DNMF_KLUHT_MR: 0 off, 2 = with the second A register set, 3 = without.
        // Round 5, measured and NOT shipped: bit identical, LDS instructions per MFMA 0.41 -> 0.20, LDS wait cycles
        // -70 %, and the same time at k = 32 (0.655 -> 0.645 ms, MFMA busy 82 -> 80 % at 2.05 -> 2.02 GHz) and +25..50 % at k = 64 (250
        // registers: two waves per SIMD) -- profiles/r05_kluht_mr_pmc.txt.  The LDS reads were not what holds these products back.
        static const long mr = tune("DNMF_KLUHT_MR", 0);
        if (mr >= 2 && kt <= 2 && rowtiles >= 2) {
            const long pairs = rowtiles / 2;
            const dim3 grid2((unsigned)pairs, (unsigned)nsplit);
            if (kt == 1 && mr == 2) DNMF_LAUNCH((kl_uht_pipe_kernel<1, true, 2, 0, 0, 2>), grid2, block, lds, st, a);
            else if (kt == 1) DNMF_LAUNCH((kl_uht_pipe_kernel<1, false, 2, 0, 0, 2>), grid2, block, lds, st, a);
            else if (mr == 2) DNMF_LAUNCH((kl_uht_pipe_kernel<2, true, 2, 0, 0, 2>), grid2, block, lds, st, a);
            else DNMF_LAUNCH((kl_uht_pipe_kernel<2, false, 2, 0, 0, 2>), grid2, block, lds, st, a);
            if (int rc = check_launch("kl_uht(pipe, 64-row)")) return rc;
            if (!(rowtiles & 1)) return DNMF_OK;
            const long r0 = pairs * 256;                                  // one 128-row tile left: the one-group kernel on it
            a.A += r0 * lda; a.W += r0 * ldw; a.out += r0 * ldo;
            const dim3 grid1(1u, (unsigned)nsplit);
            if (kt == 1) DNMF_LAUNCH((kl_uht_pipe_kernel<1, true>), grid1, block, lds, st, a);
            else DNMF_LAUNCH((kl_uht_pipe_kernel<2, true>), grid1, block, lds, st, a);
            return check_launch("kl_uht(pipe)");
        }
    }
#endif
    if (kt == 1) DNMF_LAUNCH((kl_uht_pipe_kernel<1, true>), grid, block, lds, st, a);
    else if (kt == 2) DNMF_LAUNCH((kl_uht_pipe_kernel<2, true>), grid, block, lds, st, a);
    else DNMF_LAUNCH((kl_uht_pipe_kernel<4, false>), grid, block, lds, st, a);
    return check_launch("kl_uht(pipe)");
}
