// dnmf_host.h -- host-side helpers shared by the translation units of libdnmf_hip.so (argument checks, stream handling,
// the partial-sum reduction launch).
#pragma once
#include "dnmf_common.h"
#include "dnmf_tn.h"

namespace {

// =============================================================================================== host side
int kt_of(int k) {
    if (k < 1 || k > DNMF_MAX_K) return -1;
    return k <= 32 ? 1 : (k <= 64 ? 2 : 4);
}

// lda == 0 (every row of A aliases one row: A becomes cache resident) is an experiment of the tuning build only
// (tools/kbench.py ALIAS=1); the shipped library requires lda >= n everywhere, as include/dnmf.h says
inline bool alias_ok(long lda) { return lda == 0 && tune("DNMF_ALLOW_ALIAS", 0) != 0; }

hipStream_t S(void* s) {
    clear_hip_error();
    return reinterpret_cast<hipStream_t>(s);
}

template <typename K>
void allow_lds(K kernel, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

#define REQUIRE(cond, ...) \
    do { if (!(cond)) return fail(DNMF_EINVAL, __VA_ARGS__); } while (0)

// two-stage when there are many partials per output (gram of a tall W): slices of 32 splits, then one more pass
constexpr int REDUCE_SLICE = 32;
inline int reduce_slices(int nsplit) { return nsplit > 2 * REDUCE_SLICE ? (int)cdiv(nsplit, REDUCE_SLICE) : 1; }
inline size_t reduce_scratch_bytes(int nsplit, int rows_out, long cols_out) {
    const int y = reduce_slices(nsplit);
    return y > 1 ? (size_t)y * rows_out * round_up(cols_out, 4) * sizeof(float) : 0;
}

// `scratch` must hold reduce_scratch_bytes(nsplit, rows_out, cols_out)
int launch_reduce(const float* P, long stride, long ldp, int nsplit, float* out, long ldo, int rows, long cols,
                  int rows_out, long cols_out, float* scratch, hipStream_t st) {
    const long total = (long)rows_out * cdiv(cols_out, 4);
    const unsigned gx = (unsigned)cdiv(total, 64);
    const int ny = reduce_slices(nsplit);
    static const bool wide = tune("DNMF_REDUCE_WIDE", 1) != 0;
    if (wide && ny == 1 && rows == rows_out && cols == cols_out && cols % 4 == 0 && cols >= 4096 && ldo % 4 == 0 &&
        aligned16(out) && aligned16(P) && ldp % 4 == 0 && stride % 4 == 0) {
        hipLaunchKernelGGL(reduce_partials_wide_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, st, P, stride, ldp,
                           nsplit, out, ldo, rows, cols);
        return check_launch("reduce_partials_wide");
    }
    if (ny == 1) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, 1), dim3(256), 0, st, P, stride, ldp, nsplit, nsplit, out, ldo,
                           0L, rows, cols, rows_out, cols_out);
    } else {
        const long ld2 = round_up(cols_out, 4), ys = (long)rows_out * ld2;
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, ny), dim3(256), 0, st, P, stride, ldp, nsplit, REDUCE_SLICE,
                           scratch, ld2, ys, rows, cols, rows_out, cols_out);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, 1), dim3(256), 0, st, (const float*)scratch, ys, ld2, ny, ny, out,
                           ldo, 0L, rows_out, cols_out, rows_out, cols_out);
    }
    return check_launch("reduce_partials");
}

// ---- plans and predicates shared by csrc/dnmf.hip and csrc/dnmf_kl.hip
// ---- chunking heuristics (shared by the ws-size query and the launches)
struct TnPlan { int ncolblk; int nchunks; long rows_per_chunk; long ldp; long chunk_stride; };

TnPlan plan_tn(long nrows, long ycols, int kt, int nt, long min_rows = 256) {
    TnPlan p;
    p.ncolblk = (int)cdiv(ycols, 32 * nt);
    static const long target_waves = tune("DNMF_TN_WAVES", 2048);  // one resident round: 256 CUs x 2 waves/SIMD (tunable for experiments)
    long nchunks = std::max<long>(1, target_waves / p.ncolblk);
    nchunks = std::min<long>(nchunks, std::max<long>(1, cdiv(nrows, min_rows)));
    p.rows_per_chunk = round_up(cdiv(nrows, nchunks), 16);
    p.nchunks = (int)cdiv(nrows, p.rows_per_chunk);
    p.ldp = (long)p.ncolblk * 32 * nt;
    p.chunk_stride = p.ldp * 32 * kt;
    return p;
}

inline int tn_nt(int kt) { return kt == 4 ? 2 : 4; }  // column sets per wave in TN form

inline int kl_nt(int kt) { return kt == 1 ? 4 : 2; }  // column sets per wave in kl_wtu (three live tiles: out, S/U, A)

// zero-padded factor images of the KL products (see pad_factors)
size_t pad_bytes(long m, long n, int kp) {
    return align256((size_t)m * kp * sizeof(float)) + align256((size_t)kp * round_up(n, 4) * sizeof(float));
}

template <typename TA> bool a_aligned(const TA* A) { return ((uintptr_t)A % (4 * sizeof(TA))) == 0; }

// the NT form reads 16 B per lane from A whatever its type: bf16 rows need lda % 8 == 0 and a 16-byte aligned base
template <typename TA> bool a_rows16(const TA* A, long lda) { return aligned16(A) && (lda * sizeof(TA)) % 16 == 0; }

}  // namespace
