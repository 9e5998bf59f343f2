// dnmf_host.h -- host-side helpers shared by the translation units of libdnmf_hip.so (argument checks, stream handling,
// the partial-sum reduction launch).
#pragma once
#include "dnmf_common.h"
#include "dnmf_tn.h"

namespace {

// =============================================================================================== host side
int kt_of(int k) {      // 32-wide tiles of the TUNED kernels' padded rank; -1 beyond their limit (wide ranks: wide_k)
    if (k < 1 || k > DNMF_TUNED_MAX_K) return -1;
    return k <= 32 ? 1 : (k <= 64 ? 2 : 4);
}
// 128 < k <= 256: served by composition of the tuned kernels and the plain kernels of csrc/dnmf_wide.hip
inline bool wide_k(int k) { return k > DNMF_TUNED_MAX_K && k <= DNMF_MAX_K; }
inline int kp_of(int k) { return wide_k(k) ? 256 : (kt_of(k) < 0 ? -1 : 32 * kt_of(k)); }
constexpr int WIDE_PANEL = 128;          // the panels a wide rank is cut into

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
// room behind the partial slabs of W^T A for the partial Gram tiles of dnmf_wta_gram (<= 64 chunks x 16 x 16 floats)
constexpr size_t GRAM_RIDE_BYTES = 2 * REDUCE_SLICE * 256 * sizeof(float);

// `gram` (optional, single-stage reductions only): the partial Gram tiles of a GRAM kernel are summed by one extra workgroup of
// the same launch (dnmf_tn.h: GramTail)
int launch_reduce(const float* P, long stride, long ldp, int nsplit, float* out, long ldo, int rows, long cols,
                  int rows_out, long cols_out, float* scratch, hipStream_t st, const GramTail* gram = nullptr) {
    const long total = (long)rows_out * cdiv(cols_out, 4);
    const unsigned gx = (unsigned)cdiv(total, 64);
    const int ny = reduce_slices(nsplit);
    const GramTail gt = gram ? *gram : GramTail{nullptr, nullptr, 0, 0, 0, 0};
    const unsigned extra = gram ? (unsigned)gram_tail_blocks(gram->k) : 0u;
    if (gram && ny != 1) return fail(DNMF_EINVAL, "reduce: a Gram tail needs a single-stage reduction (%d partials)", nsplit);
    static const bool wide = tune("DNMF_REDUCE_WIDE", 1) != 0;
    if (wide && ny == 1 && rows == rows_out && cols == cols_out && cols % 4 == 0 && cols >= 4096 && ldo % 4 == 0 &&
        aligned16(out) && aligned16(P) && ldp % 4 == 0 && stride % 4 == 0) {
        DNMF_LAUNCH(reduce_partials_wide_kernel, dim3((unsigned)cdiv(total, 256) + extra), dim3(256), 0, st, P, stride, ldp,
                           nsplit, out, ldo, rows, cols, gt);
        return check_launch("reduce_partials_wide");
    }
    if (ny == 1) {
        DNMF_LAUNCH(reduce_partials_kernel, dim3(gx + extra, 1), dim3(256), 0, st, P, stride, ldp, nsplit, nsplit, out, ldo,
                           0L, rows, cols, rows_out, cols_out, gt);
    } else {
        const long ld2 = round_up(cols_out, 4), ys = (long)rows_out * ld2;
        DNMF_LAUNCH(reduce_partials_kernel, dim3(gx, ny), dim3(256), 0, st, P, stride, ldp, nsplit, REDUCE_SLICE,
                           scratch, ld2, ys, rows, cols, rows_out, cols_out, gt);
        DNMF_LAUNCH(reduce_partials_kernel, dim3(gx, 1), dim3(256), 0, st, (const float*)scratch, ys, ld2, ny, ny, out,
                           ldo, 0L, rows_out, cols_out, rows_out, cols_out, gt);
    }
    return check_launch("reduce_partials");
}

// ---- plans and predicates shared by csrc/dnmf.hip and csrc/dnmf_kl.hip
// ---- chunking heuristics (shared by the ws-size query and the launches)
struct TnPlan { int ncolblk; int nchunks; long rows_per_chunk; long ldp; long chunk_stride; };

TnPlan plan_tn(long nrows, long ycols, int kt, int nt, long min_rows = 256, long waves = 0) {
    TnPlan p;
    p.ncolblk = (int)cdiv(ycols, 32 * nt);
    static const long forced = tune("DNMF_TN_WAVES", 0);           // (experiments of the tuning build)
    const long target_waves = forced ? forced : waves ? waves : 2048;   // default: one resident round, 256 CUs x 2 waves/SIMD
    long nchunks = std::max<long>(1, target_waves / p.ncolblk);
    nchunks = std::min<long>(nchunks, std::max<long>(1, cdiv(nrows, min_rows)));
    p.rows_per_chunk = round_up(cdiv(nrows, nchunks), 16);
    p.nchunks = (int)cdiv(nrows, p.rows_per_chunk);
    p.ldp = (long)p.ncolblk * 32 * nt;
    p.chunk_stride = p.ldp * 32 * kt;
    return p;
}

// W^T A with one accumulator tile per column block (k <= 32) on fp32 A below 4 GiB: ONE wave per SIMD (1024 waves) -- half the
// partial slabs, and at this rank a single wave's loads keep up with the HBM (measured, k = 32: 0.202 -> 0.183 ms at 65536 x
// 4096, 0.435 -> 0.398 at 131072 x 4096, 0.375 -> 0.355 at 65536 x 8192, 0.186 -> 0.178 at 32768 x 8192; from 4 GiB on two
// waves per SIMD are 1-2 % faster, and at k = 64 the two are equal).  Wave counts that are not a multiple of 1024 lose 20 %.
inline long wta_waves(long m, long n, int kt, size_t elt) {
    return (kt == 1 && elt == 4 && (double)m * (double)n < (double)(1L << 30)) ? 1024 : 0;
}

// k <= 16 kernels (dnmf_k16.h): DNMF_K16=0 switches them off (A/B runs)
inline bool k16_on() {
    static const bool on = tune("DNMF_K16", 1) != 0;
    return on;
}
// row chunking of tn16_kernel: waves = nchunks x (n / (16 V)), 16-row partial slabs of ld = n
struct Tn16Plan { int ncolblk; int nchunks; long rows_per_chunk; };
Tn16Plan plan_tn16(long m, long n, int v) {
    Tn16Plan p;
    p.ncolblk = (int)(n / (16 * v));
    // 2 waves per SIMD.  (4 until round 3: 0.195 -> 0.170 ms at 65536 x 4096, 0.343 -> 0.324 at 32768 x 16384, 1.307 -> 1.285 at
    // 262144 x 8192, k = 16 -- half the partial slabs to write and reduce, and the HBM stream does not need the waves.)
    static const long target = tune("DNMF_TN16_WAVES", 2048);
    long nchunks = std::max<long>(1, target / std::max(1, p.ncolblk));
    nchunks = std::min<long>(nchunks, std::max<long>(1, cdiv(m, 256)));
    p.rows_per_chunk = round_up(cdiv(m, nchunks), 16);
    p.nchunks = (int)cdiv(m, p.rows_per_chunk);
    return p;
}

// Row chunking of kl_wtu16_kernel: waves = nchunks x (n / 64) and ALL of them must be resident at once -- a second,
// partly filled round of waves runs one wave per SIMD with 4 KiB in flight each and is bound by the HBM latency (measured:
// 4096 waves on 3072 slots took 0.50 ms, of which the last third of the waves 0.2 ms).  So the wave count is the largest
// multiple of the column blocks that fits WTU16_SLOTS (256 CUs x 4 SIMDs x the waves per SIMD the register count admits).
constexpr int WTU16_WAVES_PER_SIMD = 4;
Tn16Plan plan_wtu16(long m, long n) {
    Tn16Plan p;
    p.ncolblk = (int)(n / 64);
    const long slots = 1024L * WTU16_WAVES_PER_SIMD;
    long nchunks = std::max<long>(1, slots / std::max(1, p.ncolblk));
    nchunks = std::min<long>(nchunks, std::max<long>(1, cdiv(m, 64)));
    p.rows_per_chunk = round_up(cdiv(m, nchunks), 16);
    p.nchunks = (int)cdiv(m, p.rows_per_chunk);
    return p;
}

inline int tn_nt(int kt) { return kt == 4 ? 2 : 4; }  // column sets per wave in TN form

inline int kl_nt(int kt) {   // column sets per wave in kl_wtu (three live tiles: out, S/U, A)
    static const int nt1 = (int)tune("DNMF_WTU_NT1", 4);
    return kt == 1 ? nt1 : 2;
}

// Row chunking of kl_wtu_kernel: plan_tn's round of waves, with the chunk capped so that ONE 2 GiB buffer descriptor covers a
// chunk's rows of A (the pipelined path of the kernel addresses a chunk through MUBUF; a chunk beyond the window fell back to
// the predicated blocks -- round 4: the whole 131072 x 65536 matrix of BASELINE config 4 on one GPU ran W^T U at 43 TFLOP/s
// that way, 139 with the cap).  n stands in for the leading dimension (column slices of a wider block keep the plan of their
// own width and may still leave the window: they take the slower path, correctly).
struct KlWtuPlan { int nt; long rowblks_per_chunk; long nchunks; TnPlan tn; };
KlWtuPlan plan_kl_wtu(long m, long n, int kt) {
    KlWtuPlan q;
    q.nt = kl_nt(kt);
    q.tn = plan_tn(m, n, kt, q.nt);
    const long cap_rows = ((0x7fffffffL / 4 - 32 * q.nt) / std::max<long>(n, 1) - 33) / 32 * 32;       // (rows + 32) * n + CW floats < 2^31 bytes
    long rpc = std::max<long>(1, q.tn.rows_per_chunk / 32);
    if (cap_rows >= 32) rpc = std::min<long>(rpc, cap_rows / 32);
    q.rowblks_per_chunk = rpc;
    q.nchunks = cdiv(cdiv(m, 32), rpc);
    return q;
}

// zero-padded factor images of the KL products (see pad_factors)
size_t pad_bytes(long m, long n, int kp) {
    return align256((size_t)m * kp * sizeof(float)) + align256((size_t)kp * round_up(n, 4) * sizeof(float));
}

// Factors whose rank is not a whole number of 32-wide tiles, or whose rows are not 16-byte aligned -- an NMFk sweep visits
// k = 2, 3, 5, ... -- send the NN-form kernels (S = W H in accumulators) down their predicated paths: per-element loads
// behind exec-masked branches, at which hipcc drains vmcnt.  Measured on 32768 x 16384 (tools/klbench.py): a KL step takes
// 1.78 ms at k = 32, 1.90 ms at k = 8 / 16 / 20 and 2.56-2.61 ms at k = 3 / 5 / 13.  Instead the factors are copied into
// zero-padded images [m x KP] / [KP x n] at the end of the workspace (two strided device copies, a few MB against the GB of
// A) and the kernels run their interior paths on those; zero columns of W / zero rows of H contribute nothing and the
// outputs beyond k are never stored.
// one launch writes both images: Wp [m x kp] (columns >= k zero) and Hp [kp x ldhp] (rows >= k, columns >= n zero); a thread
// per output float4.  (Round 3: the two memsets + two 2D copies this replaces were four ~5 us launches per KL product --
// 9 % of a KL step at the NMFk sweep shape, 65536 x 4096, k = 8.)
__global__ __launch_bounds__(256) void pad_factors_kernel(const float* __restrict__ W, long ldw, long m, int k, float* __restrict__ Wp,
                                                          int kp, const float* __restrict__ H, long ldh, long n,
                                                          float* __restrict__ Hp, long ldhp, BatchTab bt) {
    REBASE(W); REBASE(Wp); REBASE(H); REBASE(Hp);
    const long wq = m * (kp / 4), hq = (long)kp * (ldhp / 4);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < wq + hq; idx += (long)gridDim.x * 256) {
        float d[4];
        if (idx < wq) {
            const long r = idx / (kp / 4);
            const int c = (int)(idx % (kp / 4)) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = c + e < k ? W[r * ldw + c + e] : 0.f;
            *reinterpret_cast<f32x4*>(Wp + r * kp + c) = f32x4{d[0], d[1], d[2], d[3]};
        } else {
            const long j = idx - wq;
            const long r = j / (ldhp / 4), c = (j % (ldhp / 4)) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = (r < k && c + e < n) ? H[r * ldh + c + e] : 0.f;
            *reinterpret_cast<f32x4*>(Hp + r * ldhp + c) = f32x4{d[0], d[1], d[2], d[3]};
        }
    }
}

bool pad_factors(const float*& W, long& ldw, const float*& H, long& ldh, int& k, long m, long n, int kp, void* ws,
                        size_t ws_bytes, size_t own_need, hipStream_t st) {
    const bool friendly = k == kp && aligned16(W) && ldw % 4 == 0 && aligned16(H) && ldh % 4 == 0;
    if (friendly || tune("DNMF_KL_PAD", 1) == 0) return false;
    const size_t pb = pad_bytes(m, n, kp);
    if (!ws || ws_bytes < align256(own_need) + pb) return false;
    char* base = (char*)ws + align256(own_need);
    float* Wp = (float*)base;
    const long ldhp = round_up(n, 4);
    float* Hp = (float*)(base + align256((size_t)m * kp * sizeof(float)));
    const long quads = m * (kp / 4) + (long)kp * (ldhp / 4);
    const unsigned grid = (unsigned)std::min<long>(cdiv(quads, 256), 4096);
    DNMF_LAUNCH(pad_factors_kernel, dim3(grid), dim3(256), 0, st, W, ldw, m, k, Wp, kp, H, ldh, n, Hp, ldhp);
    if (hipGetLastError() != hipSuccess) return false;
    W = Wp; ldw = kp; H = Hp; ldh = ldhp; k = kp;
    return true;
}

struct UhtPlan { int nsplit; long cols_per_split; };

// Column splits of kl_uht (rowtiles x nsplit workgroups).  Round 4: the grid is cut to WHOLE resident rounds -- the pipelined
// kernel (csrc/dnmf_kluht.h) holds 2 / 3 / 4 workgroups per CU at k = 128 / 64 / 32, and a last round that fills half the
// chip costs as much as a full one (32768 x 16384, k = 32: 1536 workgroups on 1024 slots were 1.5 rounds).  Model: time ~
// rounds x (column tiles per split + 3 tiles' worth of prologue / epilogue); fewest splits among the best (fewer slabs to
// write and reduce); every split keeps at least 8 column tiles.
inline int uht_wgs_per_cu(int kt) { return kt == 4 ? 2 : (kt == 2 ? 3 : 4); }      // = KlUhtOcc<KT> of csrc/dnmf_kluht.h
UhtPlan plan_uht(long m, long n, int kt) {
    UhtPlan u;
    const long rowtiles = cdiv(m, 128), slots = 256L * uht_wgs_per_cu(kt);
    const long max_ns = std::min<long>(32, std::max<long>(1, n / 256));
    long best_ns = 1;
    double best_t = 0;
    for (long ns = 1; ns <= max_ns; ++ns) {
        const long cps = round_up(cdiv(n, ns), 32);
        const long nsp = cdiv(n, cps);
        const double t = (double)cdiv(rowtiles * nsp, slots) * (double)(cps / 32 + 3);
        if (ns == 1 || t < best_t * 0.97) { best_t = t; best_ns = ns; }
    }
    u.cols_per_split = round_up(cdiv(n, best_ns), 32);          // 32 = BK, the column tile of the NT-shaped kernels
    u.nsplit = (int)cdiv(n, u.cols_per_split);
    return u;
}

template <typename TA> bool a_aligned(const TA* A) { return ((uintptr_t)A % (4 * sizeof(TA))) == 0; }

// the NT form reads 16 B per lane from A whatever its type: bf16 rows need lda % 8 == 0 and a 16-byte aligned base
template <typename TA> bool a_rows16(const TA* A, long lda) { return aligned16(A) && (lda * sizeof(TA)) % 16 == 0; }

}  // namespace
