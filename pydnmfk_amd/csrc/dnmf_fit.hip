// dnmf_fit.hip -- whole fits on one rank: `itr` update steps, the final normalisation and the two squared norms of the
// relative error in ONE library call, for one problem or for a batch of same-shape problems (blockIdx.z = problem).
// Replaces the Python loop of PyNMF.fit (reference pyDNMF.py:138-182 with p_r = p_c = 1) and, batched, the sequential
// perturbation fits of an NMFk sweep (reference pyDNMFk.py:226-231).  No kernel of its own: every launch below is a launch
// of the per-step entry points, in the order the per-step callers issue them -- problem z of a batch runs the instructions
// a single fit runs, on the same operands (csrc/dnmf_common.h "batched launches").
#include "dnmf_common.h"
#include "dnmf_small.h"

// csrc/dnmf.hip: offsets of the step workspace {G, S, x, partials, total}
__attribute__((visibility("hidden"))) void dnmf_ws_offsets_(long m, long n, int k, size_t out[5]);

namespace {

enum { FIT_MU_FRO = 0, FIT_MU_KL = 1, FIT_HALS_FRO = 2 };

inline size_t al256(size_t x) { return (x + 255) & ~size_t(255); }

// The persistent small-problem MU/KL fit (csrc/dnmf_small.h): which shapes take it and with what geometry.  fp32 A, k <= 32, the slab of A +
// all of H + the slab's W in the 160 KiB of LDS of a CU (8-wave workgroups = 128-row slabs when that fits, else 4-wave = 64-row), at
// most 64 slabs per problem -- in practice n up to ~500 columns and m up to 8192 rows: the reference's example sizes.
struct SmallPlan { int kp, nw, P; bool alds; long ns; size_t lds, part_floats, bytes; bool ok; };
SmallPlan small_kl_plan(long m, long n, int k) {
    SmallPlan s{};
    if (k < 1 || k > 32 || n > 4096 || tune("DNMF_SMALL_FIT", 1) == 0) return s;
    s.kp = k <= 16 ? 16 : 32;
    s.ns = round_up(n, 16);
    // geometry by shape alone (a batched fit must equal single fits bit for bit): 128-row slabs with A in LDS when that fits; else 96-row
    // slabs with A in LDS (round 6: at 16 < k <= 32 the 1024 x 256 examples keep A on chip this way, and 20 problems x 11 slabs cover 220
    // of the 256 CUs where 8 slabs covered 160); else 128-row slabs with A streamed from the L2 (H and the slab's W in LDS); else 64-row
    // slabs with A in LDS, or streamed (short, wide problems)
    struct Try { int nw; bool alds; };
    Try tries[5] = {{8, true}, {6, true}, {8, false}, {4, true}, {4, false}};
    if (tune("DNMF_SMALL_NW6", 0)) std::swap(tries[0], tries[1]);      // (tuning build: 96-row slabs first)
    for (const auto& t : tries) {
        const size_t lds = small_kl_lds(s.kp, t.nw, n, t.alds);
        const long P = cdiv(m, 16L * t.nw);
        if (lds <= 160 * 1024 && P <= 64 && (t.nw == 4 || P >= 2)) { s.nw = t.nw; s.alds = t.alds; s.P = (int)P; s.lds = lds; break; }
    }
    if (!s.nw) return s;
    s.part_floats = (size_t)s.P * s.kp * s.ns + (size_t)s.P * s.kp * s.kp;       // W^T U / W^T A partials + (KL) column sums or (FRO) W^T W per slab
    s.part_floats = (s.part_floats + 3) & ~size_t(3);                                // ... + the granules {element, step} of H, 16-byte aligned
    s.bytes = (((s.part_floats + 2 * (size_t)s.kp * s.ns) * sizeof(float) + 255) & ~size_t(255)) + 256;      // ... + the arrival counter
    s.ok = true;
    return s;
}

// HALS on the persistent kernel (small_hals_fit_kernel): A always streamed, so LDS holds H, the slab's W, the Gram matrix and the
// workgroup's share of W^T A; fp32 or bf16-stored A
struct HalsPlan { int kp, nw, P, cw; long ns; size_t lds, lds_bf16_resident, part_floats, slot_words, bytes; bool ok; };
HalsPlan small_hals_plan(long m, long n, int k) {
    HalsPlan s{};
    if (k < 1 || k > 32 || n > 4096 || tune("DNMF_SMALL_FIT", 1) == 0) return s;
    s.kp = k <= 16 ? 16 : 32;
    s.ns = round_up(n, 16);
    for (int nw : {8, 4}) {
        const long P = cdiv(m, 16L * nw);
        const int cw = (int)cdiv(s.ns, P);
        const size_t lds = small_hals_lds(s.kp, nw, n, cw, 0);
        if (lds <= 160 * 1024 && P <= 64 && (nw == 4 || P >= 2)) {
            s.nw = nw; s.P = (int)P; s.cw = cw; s.lds = lds;
            s.lds_bf16_resident = small_hals_lds(s.kp, nw, n, cw, 2);       // (the geometry never depends on the storage: only where A is read from)
            break;
        }
    }
    if (!s.nw) return s;
    s.part_floats = (size_t)s.P * s.kp * s.ns + (size_t)s.P * s.kp * s.kp;
    s.slot_words = (((size_t)2 * s.kp * s.P) + 3) & ~size_t(3);
    s.part_floats = (s.part_floats + 3) & ~size_t(3);
    s.bytes = (((s.part_floats + s.slot_words + 2 * (size_t)s.kp * s.ns) * 4 + 255) & ~size_t(255)) + 256;      // partials | slots | granules of H | counter
    s.ok = true;
    return s;
}

// MU/FRO on bf16-stored data (params.precision = 'bfloat16'): the Frobenius kernel with TA = bf16_t, its own plan (the slab is half the
// size: it fits LDS where the fp32 one streams)
struct FroBfPlan { int kp, nw, P; bool alds; long ns; size_t lds, part_floats, bytes; bool ok; };
FroBfPlan small_fro_bf16_plan(long m, long n, int k) {
    FroBfPlan s{};
    if (k < 1 || k > 32 || n > 4096 || tune("DNMF_SMALL_FIT", 1) == 0) return s;
    s.kp = k <= 16 ? 16 : 32;
    s.ns = round_up(n, 16);
    const struct { int nw; bool alds; } tries[4] = {{8, true}, {8, false}, {4, true}, {4, false}};
    for (const auto& t : tries) {
        const size_t lds = small_fro_lds(s.kp, t.nw, n, t.alds, 2);
        const long P = cdiv(m, 16L * t.nw);
        if (lds <= 160 * 1024 && P <= 64 && (t.nw == 4 || P >= 2)) { s.nw = t.nw; s.alds = t.alds; s.P = (int)P; s.lds = lds; break; }
    }
    if (!s.nw) return s;
    s.part_floats = (size_t)s.P * s.kp * s.ns + (size_t)s.P * s.kp * s.kp;
    s.part_floats = (s.part_floats + 3) & ~size_t(3);                                  // (the granules of H behind it: 16-byte aligned)
    s.bytes = (((s.part_floats + 2 * (size_t)s.kp * s.ns) * sizeof(float) + 255) & ~size_t(255)) + 256;
    s.ok = true;
    return s;
}

// per-problem workspace: [ step workspace | s: KP floats (column sums of W) | ss2: KP doubles | sq: 2 doubles | small-fit partials ]
struct FitWs { size_t g_off, s_off, part_off, step_total, cs_off, ss2_off, sq_off, small_off, total; };

FitWs fit_layout(long m, long n, int k) {
    size_t o[5];
    dnmf_ws_offsets_(m, n, k, o);
    const int kp = dnmf_kp(k);
    FitWs f;
    f.g_off = o[0]; f.s_off = o[1]; f.part_off = o[3];
    // the partial area also serves the calls made on the factor alone (the persistent HALS W sweep and the column sums of W ask
    // for dnmf_ws_bytes(m, k, k))
    f.step_total = std::max(o[4], o[3] + al256(dnmf_ws_bytes(m, k, k)));
    f.cs_off = al256(f.step_total);
    f.ss2_off = f.cs_off + al256((size_t)kp * sizeof(float));
    f.sq_off = f.ss2_off + al256((size_t)kp * sizeof(double));
    f.small_off = f.sq_off + 256;
    f.total = f.small_off + std::max(std::max(small_kl_plan(m, n, k).bytes, small_hals_plan(m, n, k).bytes), small_fro_bf16_plan(m, n, k).bytes);
    return f;
}

struct BatchGuard {      // the batch state of this thread is set for the duration of one fit call, whatever the exit path
    BatchCtx* c;
    explicit BatchGuard(BatchCtx* ctx) : c(ctx) {}
    ~BatchGuard() { c->B = 1; c->tab.n = 0; c->tab.z0 = 0; }
};

bool overlap(const BatchFam& a, const BatchFam& b) { return a.lo < b.hi && b.lo < a.hi; }
inline bool wide_fit_k(int k) { return k > DNMF_TUNED_MAX_K; }

template <bool BF>
int hals_step(const void* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps, int w_update,
              int clamp, int column_sweep, char* ws, const FitWs& f, void* stream) {
    float* G = (float*)(ws + f.g_off);
    float* Sb = (float*)(ws + f.s_off);
    void* part = ws + f.part_off;
    const size_t part_bytes = f.step_total - f.part_off;
    int rc;
    if (w_update) {                                                                   // FRO_HALS_update_W, dist_nmf.py:873-891
        if ((rc = dnmf_gram_hht(H, k, n, ldh, G, part, part_bytes, stream))) return rc;                  // :882
        if ((rc = BF ? dnmf_aht_bf16a(A, m, n, lda, H, k, ldh, Sb, k, stream)
                     : dnmf_aht((const float*)A, m, n, lda, H, k, ldh, Sb, k, stream))) return rc;       // :883
        if (column_sweep) rc = dnmf_hals_update_w(W, m, k, ldw, Sb, k, G, eps, (double*)(ws + f.ss2_off), stream);
        else rc = dnmf_hals_sweep_w(W, m, k, ldw, Sb, k, G, eps, part, part_bytes, stream);              // :884-891
        if (rc) return rc;
    }
    if ((rc = BF ? dnmf_wta_gram_bf16a(A, m, n, lda, W, k, ldw, Sb, n, G, part, part_bytes, stream)      // :902-903
                 : dnmf_wta_gram((const float*)A, m, n, lda, W, k, ldw, Sb, n, G, part, part_bytes, stream))) return rc;
    if ((rc = dnmf_hals_update_h(H, k, n, ldh, Sb, n, G, eps, stream))) return rc;                       // :905-909
    if (clamp) {                                                                                         // pyDNMF.py:170-172
        if ((rc = dnmf_clamp_min(H, k, n, ldh, eps, stream))) return rc;
        return dnmf_clamp_min(W, m, k, ldw, eps, stream);
    }
    return DNMF_OK;
}

unsigned long long g_small_patience = 200000000ull;

// a launch (or as few as keep every workgroup resident) of a persistent small-fit kernel
int resident_launch(void (*kern)(SmallKlArgs), int threads, size_t lds, int P, SmallKlArgs a, int batch, hipStream_t st, bool* taken, const char* what) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return fail(DNMF_EHIP, "small fit: cannot raise the dynamic LDS limit");
    int nb = 0, dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return fail(DNMF_EHIP, "small fit: device query failed");
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, lds) != hipSuccess || nb < 1) { clear_hip_error(); return DNMF_OK; }
    const long cap = (long)nb * cus;
    if (cap < P) return DNMF_OK;
    const int per_launch = (int)std::min<long>(batch, cap / P);
    const int each = (int)cdiv(batch, cdiv(batch, per_launch));
    for (int z0 = 0; z0 < batch; z0 += each) {
        a.z0 = z0;
        hipLaunchKernelGGL(kern, dim3((unsigned)P, 1, (unsigned)std::min(each, batch - z0)), dim3(threads), lds, st, a);
        const int rc = check_launch(what);
        if (rc) return rc;
    }
    *taken = true;
    return DNMF_OK;
}

// all `itr` MU/KL steps of `batch` small problems: as few launches as keep every workgroup of a launch resident at once
template <int KP, int NW, bool ALDS, bool FRO>
int small_kl_launch(const SmallPlan& sp, SmallKlArgs a, int batch, hipStream_t st, bool* taken) {
    return resident_launch(FRO ? small_fro_fit_kernel<KP, NW, ALDS, float> : small_kl_fit_kernel<KP, NW, ALDS>, 64 * NW, sp.lds, sp.P, a, batch, st, taken,
                           FRO ? "small_fro_fit_kernel" : "small_kl_fit_kernel");
}

int small_fit(bool fro, const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps, int w_update, int itr,
                 int batch, long a_stride, long w_stride, long h_stride, char* ws, const FitWs& f, void* stream, bool* taken) {
    *taken = false;
    const SmallPlan sp = small_kl_plan(m, n, k);
    if (!sp.ok || itr < 1) return DNMF_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SmallKlArgs a{};
    a.A = A; a.lda = lda; a.a_stride = a_stride; a.W = W; a.ldw = ldw; a.w_stride = w_stride; a.H = H; a.ldh = ldh; a.h_stride = h_stride;
    a.m = (int)m; a.n = (int)n; a.k = k; a.eps = eps; a.itr = itr; a.w_update = w_update;
    a.part = (float*)(ws + f.small_off); a.part_stride = (long)(f.total / sizeof(float));
    a.bar = (unsigned*)(ws + f.small_off + sp.bytes - 256); a.bar_stride = (long)(f.total / sizeof(unsigned));
    a.hg = a.part + sp.part_floats; a.hg_stride = a.part_stride;                       // [kp][ns] granules {H element, step}: zeroed below
#ifdef DNMF_TUNING
    a.w_update |= (int)tune("DNMF_SMALL_ABL", 0) << 8;
#endif
    a.patience = g_small_patience;                                     // ticks of the 100 MHz wall clock (2 s unless dnmf_fit_set_timeout)
    if (batch == 1) { a.a_stride = a.w_stride = a.h_stride = 0; }
    if (!fro && !w_update) {
        // W fixed: the columns of H are independent -- a workgroup per 16 columns, no barrier, nothing to keep resident (small_kl_hfit_kernel)
        constexpr int NW = 8;
        const size_t lds = small_kl_hfit_lds(sp.kp, NW, m);
        if (lds <= 160 * 1024) {
            const dim3 grid((unsigned)(sp.ns / 16), 1, (unsigned)batch);
            a.z0 = 0;
            const auto kern = sp.kp == 16 ? small_kl_hfit_kernel<16, NW> : small_kl_hfit_kernel<32, NW>;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return fail(DNMF_EHIP, "small fit: cannot raise the dynamic LDS limit");
            hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, st, a);
            const int rc = check_launch("small_kl_hfit_kernel");
            if (!rc) *taken = true;
            return rc;
        }
    }
    if (hipMemset2DAsync(a.bar, f.total, 0, 2 * sizeof(unsigned), (size_t)batch, st) != hipSuccess) return fail(DNMF_EHIP, "small fit: memset failed");
    if (hipMemset2DAsync(a.hg, f.total, 0, 2 * (size_t)sp.kp * sp.ns * sizeof(float), (size_t)batch, st) != hipSuccess) return fail(DNMF_EHIP, "small fit: memset failed");
#define SMALL_CASE(KP_, NW_, AL_)                                                                                                 \
    if (sp.kp == KP_ && sp.nw == NW_ && sp.alds == AL_)                                                                           \
        return fro ? small_kl_launch<KP_, NW_, AL_, true>(sp, a, batch, st, taken) : small_kl_launch<KP_, NW_, AL_, false>(sp, a, batch, st, taken)
    SMALL_CASE(16, 8, true); SMALL_CASE(16, 6, true); SMALL_CASE(16, 8, false); SMALL_CASE(16, 4, true); SMALL_CASE(16, 4, false);
    SMALL_CASE(32, 8, true); SMALL_CASE(32, 6, true); SMALL_CASE(32, 8, false); SMALL_CASE(32, 4, true); SMALL_CASE(32, 4, false);
#undef SMALL_CASE
    return DNMF_OK;
}

// all `itr` HALS steps of `batch` small problems on the persistent kernel (W updated: with W fixed the hoisted H-only loop of fit_impl is
// the better path)
template <int KP, int NW, typename TA, bool ALDS>
int small_hals_launch(const HalsPlan& hp, SmallKlArgs a, int batch, hipStream_t st, bool* taken) {
    return resident_launch(small_hals_fit_kernel<KP, NW, TA, ALDS>, 64 * NW, ALDS ? hp.lds_bf16_resident : hp.lds, hp.P, a, batch, st, taken, "small_hals_fit_kernel");
}

int small_hals_fit(bool bf, const void* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps, int itr, int batch,
                   long a_stride, long w_stride, long h_stride, char* ws, const FitWs& f, void* stream, bool* taken) {
    *taken = false;
    const HalsPlan hp = small_hals_plan(m, n, k);
    if (!hp.ok || itr < 1) return DNMF_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SmallKlArgs a{};
    a.A = (const float*)A; a.lda = lda; a.a_stride = a_stride; a.W = W; a.ldw = ldw; a.w_stride = w_stride; a.H = H; a.ldh = ldh; a.h_stride = h_stride;
    a.m = (int)m; a.n = (int)n; a.k = k; a.eps = eps; a.itr = itr; a.w_update = 1; a.cw = hp.cw;
    a.part = (float*)(ws + f.small_off); a.part_stride = (long)(f.total / sizeof(float));
    a.slots = (unsigned*)(ws + f.small_off) + hp.part_floats; a.slots_stride = (long)(f.total / sizeof(unsigned));
    a.bar = (unsigned*)(ws + f.small_off + hp.bytes - 256); a.bar_stride = (long)(f.total / sizeof(unsigned));
    a.hg = a.part + hp.part_floats + hp.slot_words; a.hg_stride = a.part_stride;          // [kp][ns] granules {H element, step}
    a.patience = g_small_patience;
    if (batch == 1) { a.a_stride = a.w_stride = a.h_stride = 0; }
    if (hipMemset2DAsync(a.hg, f.total, 0, 2 * (size_t)hp.kp * hp.ns * sizeof(float), (size_t)batch, st) != hipSuccess ||
        hipMemset2DAsync(a.slots, f.total, 0xff, hp.slot_words * sizeof(unsigned), (size_t)batch, st) != hipSuccess ||
        hipMemset2DAsync(a.bar, f.total, 0, 2 * sizeof(unsigned), (size_t)batch, st) != hipSuccess)
        return fail(DNMF_EHIP, "small fit: memset failed");
#define HALS_CASE(KP_, NW_)                                                                                                       \
    if (hp.kp == KP_ && hp.nw == NW_)                                                                                             \
        return !bf ? small_hals_launch<KP_, NW_, float, false>(hp, a, batch, st, taken)                                           \
                   : (hp.lds_bf16_resident <= 160 * 1024 ? small_hals_launch<KP_, NW_, bf16_t, true>(hp, a, batch, st, taken)         \
                                                         : small_hals_launch<KP_, NW_, bf16_t, false>(hp, a, batch, st, taken))
    HALS_CASE(16, 8); HALS_CASE(16, 4); HALS_CASE(32, 8); HALS_CASE(32, 4);
#undef HALS_CASE
    return DNMF_OK;
}

int small_fro_bf16_fit(const void* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps, int itr, int batch,
                       long a_stride, long w_stride, long h_stride, char* ws, const FitWs& f, void* stream, bool* taken) {
    *taken = false;
    const FroBfPlan sp = small_fro_bf16_plan(m, n, k);
    if (!sp.ok || itr < 1) return DNMF_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    SmallKlArgs a{};
    a.A = (const float*)A; a.lda = lda; a.a_stride = a_stride; a.W = W; a.ldw = ldw; a.w_stride = w_stride; a.H = H; a.ldh = ldh; a.h_stride = h_stride;
    a.m = (int)m; a.n = (int)n; a.k = k; a.eps = eps; a.itr = itr; a.w_update = 1;
    a.part = (float*)(ws + f.small_off); a.part_stride = (long)(f.total / sizeof(float));
    a.bar = (unsigned*)(ws + f.small_off + sp.bytes - 256); a.bar_stride = (long)(f.total / sizeof(unsigned));
    a.hg = a.part + sp.part_floats; a.hg_stride = a.part_stride;
    a.patience = g_small_patience;
    if (batch == 1) { a.a_stride = a.w_stride = a.h_stride = 0; }
    if (hipMemset2DAsync(a.bar, f.total, 0, 2 * sizeof(unsigned), (size_t)batch, st) != hipSuccess) return fail(DNMF_EHIP, "small fit: memset failed");
    if (hipMemset2DAsync(a.hg, f.total, 0, 2 * (size_t)sp.kp * sp.ns * sizeof(float), (size_t)batch, st) != hipSuccess) return fail(DNMF_EHIP, "small fit: memset failed");
#define FROBF_CASE(KP_, NW_, AL_)                                                                                                 \
    if (sp.kp == KP_ && sp.nw == NW_ && sp.alds == AL_)                                                                           \
        return resident_launch(small_fro_fit_kernel<KP_, NW_, AL_, bf16_t>, 64 * NW_, sp.lds, sp.P, a, batch, st, taken, "small_fro_fit_kernel(bf16)")
    FROBF_CASE(16, 8, true); FROBF_CASE(16, 8, false); FROBF_CASE(16, 4, true); FROBF_CASE(16, 4, false);
    FROBF_CASE(32, 8, true); FROBF_CASE(32, 8, false); FROBF_CASE(32, 4, true); FROBF_CASE(32, 4, false);
#undef FROBF_CASE
    return DNMF_OK;
}

int fit_impl(int method, bool bf, const void* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps,
             int w_update, int itr, int column_sweep, int batch, long a_stride, long w_stride, long h_stride, double* sq_out,
             void* ws, size_t ws_bytes, void* stream) {
    const int kp = dnmf_kp(k);
    if (kp < 0 || !A || !W || !H || !sq_out || !ws || m < 1 || n < 1 || itr < 0 || batch < 1 || lda < n || ldw < k || ldh < n)
        return fail(DNMF_EINVAL, "fit: bad arguments (m=%ld n=%ld k=%d itr=%d batch=%d)", m, n, k, itr, batch);
    if (bf && method == FIT_MU_KL) return fail(DNMF_EINVAL, "fit: bfloat16 storage of A is for the Frobenius updates");
    const FitWs f = fit_layout(m, n, k);
    if (ws_bytes < (size_t)batch * f.total) return fail(DNMF_EWS, "fit: workspace %zu < %d x %zu", ws_bytes, batch, f.total);
    BatchCtx* ctx = dnmf_batch_();
    if (ctx->B != 1) return fail(DNMF_EINVAL, "fit: called inside a batched fit");
    BatchGuard guard(ctx);
    const size_t ea = bf ? 2 : 4;
    if (batch > 1) {
        BatchFam fam[4] = {{(unsigned long)A, (unsigned long)A + ((size_t)(m - 1) * lda + n) * ea, (long)(a_stride * (long)ea)},
                           {(unsigned long)W, (unsigned long)W + ((size_t)(m - 1) * ldw + k) * 4, w_stride * 4},
                           {(unsigned long)H, (unsigned long)H + ((size_t)(k - 1) * ldh + n) * 4, h_stride * 4},
                           {(unsigned long)ws, (unsigned long)ws + f.total, (long)f.total}};
        for (int i = 0; i < 4; ++i) {
            if (fam[i].stride % 16 != 0 || (unsigned long)(fam[i].stride < 0 ? -fam[i].stride : fam[i].stride) < fam[i].hi - fam[i].lo)
                return fail(DNMF_EINVAL, "fit: stride of operand %d (%ld bytes) must be a multiple of 16 bytes and span one problem (%lu bytes)",
                            i, fam[i].stride, fam[i].hi - fam[i].lo);
            for (int j = 0; j < i; ++j)
                if (overlap(fam[i], fam[j])) return fail(DNMF_EINVAL, "fit: operands %d and %d overlap", j, i);
            ctx->tab.f[i] = fam[i];
        }
        ctx->tab.n = 4;
        ctx->B = batch;
    }
    char* base = (char*)ws;
    int rc = DNMF_OK;
    bool small = false;
    const bool pers = dnmf_persistent_on_() != 0;
    if (!pers) column_sweep = 1;                                       // (the HALS steps below take the column launches)
    if (pers && method == FIT_HALS_FRO && w_update && !column_sweep) {
        const int B = ctx->B;
        ctx->B = 1;
        rc = small_hals_fit(bf, A, m, n, lda, W, ldw, H, ldh, k, eps, itr, batch, a_stride, w_stride, h_stride, base, f, stream, &small);
        ctx->B = B;
        if (rc) return rc;
    }
    if (pers && method == FIT_MU_FRO && bf && w_update) {
        const int B = ctx->B;
        ctx->B = 1;
        rc = small_fro_bf16_fit(A, m, n, lda, W, ldw, H, ldh, k, eps, itr, batch, a_stride, w_stride, h_stride, base, f, stream, &small);
        ctx->B = B;
        if (rc) return rc;
    }
    // (Frobenius MU with W fixed: the hoisted H-only loop below beats the barrier kernel)
    if (pers && (method == FIT_MU_KL || (method == FIT_MU_FRO && w_update)) && !bf) {
        // small fp32 problems: the whole loop as one persistent kernel per batch (csrc/dnmf_small.h); launched unbatched -- it indexes the problems itself
        const int B = ctx->B;
        ctx->B = 1;
        rc = small_fit(method == FIT_MU_FRO, (const float*)A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, itr, batch, a_stride, w_stride, h_stride, base, f, stream, &small);
        ctx->B = B;
        if (rc) return rc;
        // (the W-fixed KL kernel leaves W alone; the clamp of pyDNMF.py:155 after step 0 is this launch)
        if (small && method == FIT_MU_KL && !w_update && (rc = dnmf_clamp_min(W, m, k, ldw, eps, stream))) return rc;
    }
    // W fixed (the regression fit of an NMFk sweep, pyDNMFk.py:243-247): the Frobenius H updates read A only through W^T A, and W^T A and
    // W^T W are loop invariants once W has been clamped (the clamp after step 0 is the only thing that can still change W; it is idempotent
    // from then on).  Steps 0 and 1 run in full -- step 1 leaves W^T A and W^T W of the final W in the workspace --, every later step is
    // the H update alone on those: the same values the step would recompute, so the same bits, without the two passes over A.
    const bool h_only = !w_update && method != FIT_MU_KL && !wide_fit_k(k);
    const long ldatw = method == FIT_MU_FRO ? round_up(n, 4) : n;
    for (int i = 0; i < itr && !rc && !small; ++i) {                                  // pyDNMF.py:151-172
        const int clamp = (i % 10 == 0);
        if (h_only && i >= 2) {
            float* G = (float*)(base + f.g_off);
            float* Sb = (float*)(base + f.s_off);
            if (method == FIT_MU_FRO) rc = dnmf_mu_update_h(H, k, n, ldh, Sb, ldatw, G, eps, clamp, stream);          // dist_nmf.py:750-751
            else {
                rc = dnmf_hals_update_h(H, k, n, ldh, Sb, ldatw, G, eps, stream);                                     // :905-909
                if (!rc && clamp) rc = dnmf_clamp_min(H, k, n, ldh, eps, stream);
            }
            continue;
        }
        if (method == FIT_MU_FRO)
            rc = bf ? dnmf_mu_fro_step_bf16a(A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, clamp, ws, f.step_total, stream)
                    : dnmf_mu_fro_step((const float*)A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, clamp, ws, f.step_total, stream);
        else if (method == FIT_MU_KL)
            rc = dnmf_mu_kl_step((const float*)A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, clamp, ws, f.step_total, stream);
        else
            rc = bf ? hals_step<true>(A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, clamp, column_sweep, base, f, stream)
                    : hals_step<false>(A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, clamp, column_sweep, base, f, stream);
    }
    if (rc) return rc;
    // normalize_features (pyDNMF.py:185-194): s = column sums of W; W /= s + eps; H *= s^T
    float* s = (float*)(base + f.cs_off);
    void* part = base + f.part_off;
    const size_t part_bytes = f.step_total - f.part_off;
    if ((rc = dnmf_colsum(W, m, k, ldw, s, part, part_bytes, stream))) return rc;
    if ((rc = dnmf_scale_cols_div(W, m, k, ldw, s, eps, stream))) return rc;
    if ((rc = dnmf_scale_rows_mul(H, k, n, ldh, s, stream))) return rc;
    // relative_err (pyDNMF.py:205-218): sum (A - W H)^2 and sum A^2; the caller takes the square roots
    double* sq = (double*)(base + f.sq_off);
    if ((rc = bf ? dnmf_resid_sqnorm_ws_bf16a(A, m, n, lda, W, ldw, H, ldh, k, sq, ws, f.step_total, stream)
                 : dnmf_resid_sqnorm_ws((const float*)A, m, n, lda, W, ldw, H, ldh, k, sq, ws, f.step_total, stream))) return rc;
    if ((rc = bf ? dnmf_sqnorm_bf16a(A, m, n, lda, sq + 1, stream) : dnmf_sqnorm((const float*)A, m, n, lda, sq + 1, stream))) return rc;
    if (hipMemcpy2DAsync(sq_out, 2 * sizeof(double), sq, f.total, 2 * sizeof(double), (size_t)batch, hipMemcpyDeviceToDevice,
                         reinterpret_cast<hipStream_t>(stream)) != hipSuccess)
        return fail(DNMF_EHIP, "fit: copy of the squared norms failed");
    return DNMF_OK;
}

}  // namespace

// read-and-clear of the persistent small fit's sticky time-out word (csrc/dnmf_hals.hip: dnmf_hals_sweep_status reports it with the sweep's)
__attribute__((visibility("hidden"))) int dnmf_small_timeout_take_(unsigned* out) {
    unsigned v = 0;
    const unsigned zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_small_timeout), sizeof(v), 0, hipMemcpyDeviceToHost) != hipSuccess) return DNMF_EHIP;
    if (v && hipMemcpyToSymbol(HIP_SYMBOL(g_small_timeout), &zero, sizeof(zero), 0, hipMemcpyHostToDevice) != hipSuccess) return DNMF_EHIP;
    *out = v;
    return DNMF_OK;
}

__attribute__((visibility("hidden"))) void dnmf_team_set_patience_(unsigned long long ticks);   // csrc/dnmf_team.hip

extern "C" {

int dnmf_fit_set_timeout(double seconds) {
    if (!(seconds > 0.0) || seconds > 3600.0) return fail(DNMF_EINVAL, "fit_set_timeout: %g s", seconds);
    g_small_patience = (unsigned long long)(seconds * 1e8) + 1ull;
    dnmf_team_set_patience_(g_small_patience);
    return DNMF_OK;
}

int dnmf_hals_fit_persistent(long m, long n, int k) {
    return (dnmf_persistent_on_() && m >= 1 && n >= 1 && small_hals_plan(m, n, k).ok) ? 1 : 0;
}

int dnmf_mu_fit_persistent(long m, long n, int k) {
    return (dnmf_persistent_on_() && m >= 1 && n >= 1 && small_kl_plan(m, n, k).ok) ? 1 : 0;
}

size_t dnmf_ws_bytes_fit(long m, long n, int k, int batch) {
    if (dnmf_kp(k) < 0 || m < 1 || n < 1 || batch < 1) return 0;
    return (size_t)batch * fit_layout(m, n, k).total;
}

int dnmf_mu_fro_fit(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps, int w_update,
                    int itr, int batch, long a_stride, long w_stride, long h_stride, double* sq_out, void* ws, size_t ws_bytes,
                    void* stream) {
    return fit_impl(FIT_MU_FRO, false, A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, itr, 0, batch, a_stride, w_stride, h_stride,
                    sq_out, ws, ws_bytes, stream);
}
int dnmf_mu_fro_fit_bf16a(const void* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps,
                          int w_update, int itr, int batch, long a_stride, long w_stride, long h_stride, double* sq_out, void* ws,
                          size_t ws_bytes, void* stream) {
    return fit_impl(FIT_MU_FRO, true, A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, itr, 0, batch, a_stride, w_stride, h_stride,
                    sq_out, ws, ws_bytes, stream);
}
int dnmf_mu_kl_fit(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps, int w_update,
                   int itr, int batch, long a_stride, long w_stride, long h_stride, double* sq_out, void* ws, size_t ws_bytes,
                   void* stream) {
    return fit_impl(FIT_MU_KL, false, A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, itr, 0, batch, a_stride, w_stride, h_stride,
                    sq_out, ws, ws_bytes, stream);
}
int dnmf_hals_fro_fit(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps,
                      int w_update, int itr, int column_sweep, int batch, long a_stride, long w_stride, long h_stride, double* sq_out,
                      void* ws, size_t ws_bytes, void* stream) {
    return fit_impl(FIT_HALS_FRO, false, A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, itr, column_sweep, batch, a_stride, w_stride,
                    h_stride, sq_out, ws, ws_bytes, stream);
}
int dnmf_hals_fro_fit_bf16a(const void* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps,
                            int w_update, int itr, int column_sweep, int batch, long a_stride, long w_stride, long h_stride,
                            double* sq_out, void* ws, size_t ws_bytes, void* stream) {
    return fit_impl(FIT_HALS_FRO, true, A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, itr, column_sweep, batch, a_stride, w_stride,
                    h_stride, sq_out, ws, ws_bytes, stream);
}

}  // extern "C"
