// dnmf_f64_kl.h -- the two KL products in float64 WITHOUT the m x n quotient image (included by dnmf_f64.hip, inside its anonymous
// namespace, after the operand helpers):
//   uht  S[r][j] = sum_c U[r][c] H[j][c]      wtu  S[j][c] = sum_r W[r][j] U[r][c]      U = A / (W H + eps)   (dist_nmf.py:806-810)
// Round 5 went through U in memory (f64_nn_cols_kernel writes 8 m n bytes, the NT / TN kernels read them back: 1.03 + 0.77 ms and
// 1.03 + 0.62 ms at 65536 x 4096, k = 64, where the 4 m n k flops of one product pair are 0.87 ms of the fp64 matrix cores).  Here a
// wave forms a 16 x 16 tile of W H on the matrix cores, divides its four elements per lane in registers and feeds them straight into the
// second product -- the C/D layout of v_mfma_f64_16x16x4_f64 (col = i, row = q + 4 reg) IS an operand layout of the next MFMA when the
// first product is taken in the right orientation, so nothing moves between lanes:
//   uht: first product TRANSPOSED (H^T as A-operand, W^T as B-operand): lane (i, q) then holds U[row i][col q + 4 reg] = an A-operand
//        (row i, kk = q) of step `reg` of U H^T;
//   wtu: first product as written (W as A-operand, H as B-operand): lane (i, q) holds U[row q + 4 reg][col i] = a B-operand
//        (kk = q, col i) of step `reg` of W^T U.
// The 16 columns (uht) / the contraction index j (both) are permuted so that every operand is read as consecutive doubles of a row.
// k <= 64 (NT <= 4 tiles of 16): beyond, the accumulators plus two operand sets leave the register file (the caller keeps the image path).

// a / d for 0 < d < inf away from the ends of the exponent range (d = (W H) + eps): the sequence hipcc emits for a double division
// (reciprocal, two Newton steps, quotient, one residual correction) without its range scaling and special-case fix-up -- 8 instructions
// instead of 13, the same bits wherever no intermediate under- or overflows (div_pos is the fp32 twin, dnmf_common.h).  On gfx950 the
// fp64 vector instructions and the fp64 MFMAs do not overlap (tools/f64clock.hip: four IEEE divisions per 32 MFMAs cost 11 %).
__device__ __forceinline__ double div_pos64(double a, double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-d, q, a), r, q);
}

// ------------------------------------------------------------------------------------------------ uht: row strips
// one wave = RT 16-row tiles of A (its rows of W stay in registers), all columns of its split, 16 at a time; tile column of MFMA row
// rho = q + 4 reg is c0 + 4 q + reg (so a lane's four elements of A are consecutive), contraction index of step s in lane group q is
// j = q KS + s (KS consecutive doubles of a row of W).  Operands through buffer descriptors (dnmf_f64.hip "MUBUF operand loads"): no
// branch around a load, masks in the lane offsets; the column advance is the wave-uniform SGPR offset.  VEC: n % 4 == 0.
// No row mask anywhere: a row beyond m reads zeros and only feeds its own, never stored, output row.  FULL: no column mask either.
template <int NT, int RT, bool VEC, bool FULL, int D = 2>
__global__ __launch_bounds__(256) void f64_kl_uht_kernel(const double* __restrict__ A, long lda, long m, long n, const double* __restrict__ W, long ldw,
                                                         const double* __restrict__ H, long ldh, int kc, double eps, double* __restrict__ out, long ldo,
                                                         long split_stride, long cols_per_split) {
    constexpr int KS = 4 * NT;
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const long r0 = ((long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * (16 * RT);
    if (r0 >= m) return;
    const long cb = (long)blockIdx.y * cols_per_split;
    const long ce = cb + cols_per_split < n ? cb + cols_per_split : n;
    const long rows = m - r0 < 16 * RT ? m - r0 : 16 * RT;
    const i32x4 ad = rsrc64(A + r0 * lda, ((rows - 1) * lda + n) * 8), wd = rsrc64(W + r0 * ldw, ((rows - 1) * ldw + kc) * 8);
    const i32x4 hd = rsrc64(H, ((long)(kc - 1) * ldh + n) * 8);
    int avo[RT];
    double wreg[RT][KS];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        avo[rt] = (int)(((16 * rt + i) * lda + 4 * q) * 8);
#pragma unroll
        for (int v = 0; v < NT; ++v) {
            double t4[4];
            ldq<4, false>(t4, wd, (int)(((16 * rt + i) * ldw + q * KS + 4 * v) * 8), 0, clampi(kc - (q * KS + 4 * v), 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) wreg[rt][4 * v + e] = t4[e];
        }
    }
    const int c1 = 4 * (i & 3) + (i >> 2);
    int h1vo[KS], h2vo[NT];                          // first product: H[j = q KS + s][column of MFMA row i]; second: H[16 t + i][c0 + 4 q ..]
#pragma unroll
    for (int s = 0; s < KS; ++s) h1vo[s] = q * KS + s < kc ? (int)(((q * KS + s) * ldh + c1) * 8) : BUF_OOB;
#pragma unroll
    for (int t = 0; t < NT; ++t) h2vo[t] = 16 * t + i < kc ? (int)(((16 * t + i) * ldh + 4 * q) * 8) : BUF_OOB;
    f64x4 acc[RT][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[rt][t] = f64x4{0.0, 0.0, 0.0, 0.0};
    auto load_a = [&](auto edge, double (&a)[RT][4], long c0) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge)::value;
        const int so = sgpr(c0 * 8);
        const int nv = EDGE ? clampi(ce - (c0 + 4 * q), 4) : 4;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) ldq<4, VEC || !EDGE>(a[rt], ad, avo[rt], so, nv);
    };
    auto load_h = [&](auto edge, double (&h1)[KS], double (&h2)[NT][4], long c0) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge)::value;
        const int so = sgpr(c0 * 8);
        const int nv = EDGE ? clampi(ce - (c0 + 4 * q), 4) : 4, n1 = EDGE ? (c0 + c1 < ce ? 1 : 0) : 1;
#pragma unroll
        for (int s = 0; s < KS; ++s) { double t1[1]; ldq<1, true>(t1, hd, h1vo[s], so, n1); h1[s] = t1[0]; }
#pragma unroll
        for (int t = 0; t < NT; ++t) ldq<4, VEC || !EDGE>(h2[t], hd, h2vo[t], so, nv);
    };
    auto tile = [&](auto edge, const double (&a)[RT][4], const double (&h1)[KS], const double (&h2)[NT][4], long c0) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge)::value;
        const int nv = EDGE ? clampi(ce - (c0 + 4 * q), 4) : 4;
        f64x4 s4[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {                              // the RT first products are independent chains
            s4[rt] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KS; ++s) s4[rt] = MFMA64(h1[s], wreg[rt][s], s4[rt]);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            double u[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u[e] = div_pos64(a[rt][e], s4[rt][e] + eps);
                if constexpr (EDGE) u[e] = e < nv ? u[e] : 0.0;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[rt][t] = MFMA64(u[e], h2[t][e], acc[rt][t]);
        }
    };
    // Two rings: the pieces of A come from HBM and are small (RT x 4 doubles a lane) -- DA tiles ahead; the operands from H come from
    // the L2 and are large -- one tile ahead.  Nothing is conditional around a load (f64_nt_kernel): FULL = every split is a whole number
    // of groups of U tiles (the host checks), the loads of the last groups that would run past the split re-read its last tile into ring
    // slots nobody consumes; otherwise every tile carries its column mask and a tile beyond the split is all zeros.
    constexpr int DA = D + 2, U = DA % 2 == 0 ? DA : 2 * DA;
    const long ntiles = cdiv(ce - cb, 16);
    auto tcol = [&](long t) __attribute__((always_inline)) { return cb + 16 * (FULL ? (t < ntiles ? t : ntiles - 1) : t); };
    using EdgeT = std::integral_constant<bool, !FULL>;
    double a[DA][RT][4], h1[2][KS], h2[2][NT][4];
    static_for<0, DA - 1>([&](auto d) __attribute__((always_inline)) { load_a(EdgeT{}, a[d], tcol(d)); });
    load_h(EdgeT{}, h1[0], h2[0], cb);
    for (long it = 0; it < ntiles; it += U)
        static_for<0, U>([&](auto d) __attribute__((always_inline)) {
            load_a(EdgeT{}, a[(d + DA - 1) % DA], tcol(it + d + (DA - 1)));
            load_h(EdgeT{}, h1[(d + 1) % 2], h2[(d + 1) % 2], tcol(it + d + 1));
            __builtin_amdgcn_sched_barrier(0);                    // (loads first, see f64_nt_kernel)
            tile(EdgeT{}, a[d % DA], h1[d % 2], h2[d % 2], cb + 16 * (it + d));
        });
    double* o = out + (long)blockIdx.y * split_stride;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long row = r0 + 16 * rt + q + 4 * r;
                const int col = 16 * t + i;
                if (row < m && col < kc) o[row * ldo + col] = acc[rt][t][r];
            }
}

// ------------------------------------------------------------------------------------------------ wtu: column strips
// one wave = 16 CT columns (tile cb holds the columns c0 + CT i + cb, as in f64_tn_kernel: CT consecutive doubles of a row per lane) --
// its k x 16 CT piece of H stays in registers -- and a range of rows, 16 at a time; partial slabs per row range, reduced in slab order
// by f64_reduce_kernel.  Output row tile t holds the factor columns (t / VJ) 16 VJ + VJ i + t % VJ (f64_tn_kernel's permutation).
// Per-tile descriptors for A and W (rows beyond the range lie outside the byte count).  VEC: n % CT == 0 and kc % 4 == 0.
// Columns need no mask (a column beyond n reads zeros and only feeds its own, never stored, output column); the last tile of a row
// range masks its rows (EDGE), full tiles carry no mask.
template <int NT, int CT, bool VEC, int D = 2>
__global__ __launch_bounds__(256) void f64_kl_wtu_kernel(const double* __restrict__ A, long lda, long m, long n, const double* __restrict__ W, long ldw,
                                                         const double* __restrict__ H, long ldh, int kc, double eps, long rows_per_chunk, int ncolblk,
                                                         long nwaves, double* __restrict__ P, long chunk_stride, long ldp) {
    constexpr int KS = 4 * NT, VJ = NT < 4 ? NT : 4, NV = NT / VJ;
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const long gw = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (gw >= nwaves) return;
    const long chunk = gw / ncolblk, c0 = (gw % ncolblk) * (16 * CT);
    const long rb = chunk * rows_per_chunk;
    const long re = rb + rows_per_chunk < m ? rb + rows_per_chunk : m;
    const int nva = clampi(n - (c0 + CT * i), CT);
    double hb[KS][CT];
    {
        const i32x4 hd = rsrc64(H, ((long)(kc - 1) * ldh + n) * 8);
#pragma unroll
        for (int s = 0; s < KS; ++s) ldq<CT, VEC>(hb[s], hd, q * KS + s < kc ? (int)(((q * KS + s) * ldh + c0 + CT * i) * 8) : BUF_OOB, 0, nva);
    }
    f64x4 acc[NT][CT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int cb = 0; cb < CT; ++cb) acc[t][cb] = f64x4{0.0, 0.0, 0.0, 0.0};
    int avo[4], w2vo[4], nv1[NT], nv2[NV];
    const int w1vo = (int)((i * ldw + q * KS) * 8);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        avo[r] = (int)(((q + 4 * r) * lda + c0 + CT * i) * 8);
        w2vo[r] = (int)(((q + 4 * r) * ldw + VJ * i) * 8);
    }
#pragma unroll
    for (int v = 0; v < NT; ++v) nv1[v] = clampi(kc - (q * KS + 4 * v), 4);
#pragma unroll
    for (int v = 0; v < NV; ++v) nv2[v] = clampi(kc - (16 * VJ * v + VJ * i), VJ);
    auto load_a = [&](double (&av)[4][CT], long r0) __attribute__((always_inline)) {
        const i32x4 ad = rsrc64(A + r0 * lda, ((re - r0 - 1) * lda + n) * 8);
#pragma unroll
        for (int r = 0; r < 4; ++r) ldq<CT, VEC>(av[r], ad, avo[r], 0, nva);
    };
    auto load_w = [&](double (&w1)[KS], double (&w2)[4][NV][VJ], long r0) __attribute__((always_inline)) {
        const i32x4 wd = rsrc64(W + r0 * ldw, ((re - r0 - 1) * ldw + kc) * 8);
#pragma unroll
        for (int v = 0; v < NT; ++v) {
            double t4[4];
            ldq<4, VEC>(t4, wd, w1vo + 32 * v, 0, nv1[v]);
#pragma unroll
            for (int e = 0; e < 4; ++e) w1[4 * v + e] = t4[e];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int v = 0; v < NV; ++v) ldq<VJ, VEC>(w2[r][v], wd, w2vo[r] + 16 * VJ * 8 * v, 0, nv2[v]);
    };
    auto tile = [&](auto edge, const double (&w1)[KS], const double (&w2)[4][NV][VJ], const double (&av)[4][CT], long r0) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge)::value;
        f64x4 s4[CT];
#pragma unroll
        for (int cb = 0; cb < CT; ++cb) s4[cb] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int cb = 0; cb < CT; ++cb) s4[cb] = MFMA64(w1[s], hb[s][cb], s4[cb]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double u[CT];
#pragma unroll
            for (int cb = 0; cb < CT; ++cb) {
                u[cb] = div_pos64(av[r][cb], s4[cb][r] + eps);
                if constexpr (EDGE) u[cb] = r0 + q + 4 * r < re ? u[cb] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int cb = 0; cb < CT; ++cb) acc[t][cb] = MFMA64(w2[r][t / VJ][t % VJ], u[cb], acc[t][cb]);
        }
    };
    // rings: A DA tiles ahead (from HBM), the rows of W one tile ahead (from the L2); nothing conditional around a load (f64_nt_kernel):
    // the trip count is rounded up, a tile beyond the row range has descriptors of zero bytes, reads zeros and is masked like the last rows
    constexpr int DA = D + 2, U = DA % 2 == 0 ? DA : 2 * DA;
    double w1[2][KS], w2[2][4][NV][VJ], av[DA][4][CT];
    static_for<0, DA - 1>([&](auto d) __attribute__((always_inline)) { load_a(av[d], rb + 16 * d); });
    load_w(w1[0], w2[0], rb);
    for (long r = rb; r < re; r += 16 * U)
        static_for<0, U>([&](auto d) __attribute__((always_inline)) {
            load_a(av[(d + DA - 1) % DA], r + 16 * (d + DA - 1));
            load_w(w1[(d + 1) % 2], w2[(d + 1) % 2], r + 16 * (d + 1));
            __builtin_amdgcn_sched_barrier(0);                    // (loads first, see f64_nt_kernel)
            tile(std::true_type{}, w1[d % 2], w2[d % 2], av[d % DA], r + 16 * d);
        });
    double* o = P + chunk * chunk_stride;
    const bool vst = (ldp % 2 == 0) && (((uintptr_t)P & 15) == 0) && (chunk_stride % 2 == 0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ip = q + 4 * r;
            const int j = (t / VJ) * 16 * VJ + VJ * ip + t % VJ;
            if (j >= kc) continue;
            double* dst = o + (long)j * ldp + c0 + CT * i;
            if (vst && CT >= 2 && c0 + CT * i + CT <= n) {
#pragma unroll
                for (int cb = 0; cb + 1 < CT; cb += 2) *reinterpret_cast<f64x2*>(dst + cb) = f64x2{acc[t][cb][r], acc[t][cb + 1][r]};
            } else {
#pragma unroll
                for (int cb = 0; cb < CT; ++cb)
                    if (c0 + CT * i + cb < n) dst[cb] = acc[t][cb][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------ plans
constexpr int KL64_MAX_K = 64;
constexpr int kl_rt(int nt) { return nt <= 2 ? 4 : 2; }           // uht: 16-row tiles per wave
constexpr int kl_ct(int nt) { return nt <= 2 ? 4 : 2; }           // wtu: 16-column tiles per wave
struct KlUhtPlan { int rt; long nsplit, cps; };
inline KlUhtPlan plan_kl_uht(long m, long n, int k) {
    KlUhtPlan p;
    p.rt = kl_rt(tiles16(k));
    while (p.rt > 1 && cdiv(m, 16 * p.rt) < 1024) p.rt >>= 1;      // (a wave per SIMD first)
    const long wr = cdiv(m, 16 * p.rt);
    long ns = std::max<long>(1, std::min<long>(2048 / wr, n / 512));
    p.cps = round_up(cdiv(n, ns), 64);                             // (whole groups of four tiles where n allows: the mask-free kernel)
    p.nsplit = cdiv(n, p.cps);
    return p;
}
struct KlWtuPlan { int ct, ncolblk; long nchunks, rows_per_chunk; };
inline KlWtuPlan plan_kl_wtu(long m, long n, int k) {
    KlWtuPlan p;
    p.ct = kl_ct(tiles16(k));
    p.ncolblk = (int)cdiv(n, 16 * p.ct);
    long nch = std::max<long>(1, 2048 / p.ncolblk);
    nch = std::min<long>(nch, std::max<long>(1, cdiv(m, 64)));
    p.rows_per_chunk = round_up(cdiv(m, nch), 16);
    p.nchunks = cdiv(m, p.rows_per_chunk);
    return p;
}
inline size_t kl64_ws_bytes(long m, long n, int k) {
    if (k > KL64_MAX_K) return 0;
    const size_t kp = 16 * tiles16(k), D = sizeof(double);
    const KlUhtPlan u = plan_kl_uht(m, n, k);
    size_t b = u.nsplit > 1 ? (size_t)u.nsplit * m * kp * D : 0;
    b = std::max(b, (size_t)plan_kl_wtu(m, n, k).nchunks * kp * round_up(n, 16) * D);
    return b;
}
