// dnmf_split.h -- the two big contractions (A H^T, W^T A) on the bf16 matrix cores with fp32-grade products ("bf16x6").
// Part of libdnmf_hip.so (translation unit csrc/dnmf_split.hip).
//
// Why: from k = 33 on the fp32 kernels of dnmf_nt.h / dnmf_tn.h are bound by v_mfma_f32_32x32x2_f32 (256 flop per CU
// and clock, 85-88 % busy: at k = 64 2.3 ms per pass over the 8.6 GB of A, where the HBM alone would need 1.1-1.4 ms).
// v_mfma_f32_32x32x16_bf16 does 16x the work per clock, so even six of them per fp32 product leave the pass memory bound.
//
// Arithmetic: every fp32 operand x is cut into three bf16 pieces x = x1 + x2 + x3, x1 = rne_bf16(x), x2 = rne_bf16(x - x1),
// x3 = rne_bf16(x - x1 - x2); both subtractions are exact in fp32 and |x2| <= 2^-8 |x|, |x3| <= 2^-16 |x|, |x - x1 - x2 - x3|
// <= 2^-24 |x| (three 8-bit significands cover the 24 bits of fp32 unless x is within 2^16 of the smallest normal number).
// A product is taken as  x y ~ x1 y1 + (x1 y2 + x2 y1) + (x1 y3 + x2 y2 + x3 y1): every piece product is exact in the fp32
// accumulator (8 x 8 significant bits), the three dropped terms are bounded by (2^-24 + 2^-24 + 2^-32) |x y| and have random
// sign (round-to-nearest residuals), i.e. the product is as good as ONE fp32 rounding of x y, which is what the fp32 MFMA
// chain commits per term as well.  Accumulation is fp32 in both; the 32x32x16 instruction adds 16 products per accumulator
// update where the fp32 one adds 2, so six updates per 16 contraction indices stand against eight.  Measured against a
// float64 product the two paths are equally close (tests/test_gpu_split.py).
//
// Layout: the streamed operand A is read ONCE from HBM (fp32, cut in registers right before its MFMAs; or bf16-stored, then it
// is its own single piece and nothing is cut).  For W^T A the contraction runs over the rows of A, so a lane's 8-row x NT-column
// block already holds its fragments (of NT column-interleaved output tiles) and A never touches LDS; for A H^T the MFMA wants
// a row of A per lane, and A goes through LDS with coalesced loads.  The small operand (H, or W transposed) is cut once per
// call into a bf16 image [piece][k][index] in the workspace and staged through LDS.
#pragma once
#include "dnmf_common.h"
#include "dnmf_nt.h"
#include "dnmf_tn.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// (lo, hi) -> packed round-to-nearest-even bf16 pair: v_cvt_pk_bf16_f32, through the compiler (a vector fptrunc), NOT inline
// asm: hipcc's hazard recognizer does not see inside an asm statement, and a cvt that overwrites a register an in-flight MFMA
// still reads as its operand was scheduled without the wait states -- run-to-run different bits in tnx_kernel once its
// cuts were interleaved with the MFMAs of the previous column.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int cvt_pk_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}

// two floats -> their three bf16 pieces, packed pairwise (11 VALU instructions)
__device__ __forceinline__ void split_pair(float a, float b, unsigned int& s1, unsigned int& s2, unsigned int& s3) {
    s1 = cvt_pk_bf16(a, b);
    float ra = a - __uint_as_float(s1 << 16), rb = b - __uint_as_float(s1 & 0xffff0000u);
    s2 = cvt_pk_bf16(ra, rb);
    ra -= __uint_as_float(s2 << 16);
    rb -= __uint_as_float(s2 & 0xffff0000u);
    s3 = cvt_pk_bf16(ra, rb);
}

__device__ __forceinline__ void split8(const float (&v)[8], u32x4& s1, u32x4& s2, u32x4& s3) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned int a, b, c;
        split_pair(v[2 * p], v[2 * p + 1], a, b, c);
        s1[p] = a; s2[p] = b; s3[p] = c;
    }
}

// the six piece products of one 32 x 32 x 16 tile step, small terms first
__device__ __forceinline__ void mfma_x6(f32x16& acc, const u32x4& a1, const u32x4& a2, const u32x4& a3, const u32x4& b1,
                                        const u32x4& b2, const u32x4& b3) {
    acc = mfma_bf16(a3, b1, acc);
    acc = mfma_bf16(a2, b2, acc);
    acc = mfma_bf16(a1, b3, acc);
    acc = mfma_bf16(a2, b1, acc);
    acc = mfma_bf16(a1, b2, acc);
    acc = mfma_bf16(a1, b1, acc);
}

// bf16-STORED streamed operand: x is its own (only) piece, so a product needs the three pieces of the other factor only
__device__ __forceinline__ void mfma_x3(f32x16& acc, const u32x4& x, const u32x4& y1, const u32x4& y2, const u32x4& y3, bool x_is_a) {
    if (x_is_a) {
        acc = mfma_bf16(x, y3, acc);
        acc = mfma_bf16(x, y2, acc);
        acc = mfma_bf16(x, y1, acc);
    } else {
        acc = mfma_bf16(y3, x, acc);
        acc = mfma_bf16(y2, x, acc);
        acc = mfma_bf16(y1, x, acc);
    }
}

// ---------------------------------------------------------------------------------------------- cutting the small operand
// S[s][r][c] = piece s of Y[r][c] for r < rows_pad, c < ld_s (bf16; zero outside [yrows x ycols]).  One thread = 8 columns.
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* __restrict__ Y, long ldy, int yrows, long ycols,
                                                          bf16_t* __restrict__ S, long ld_s, long split_stride, int rows_pad) {
    const long c8 = ld_s / 8;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows_pad * c8) return;
    const int r = (int)(idx / c8);
    const long c = (idx % c8) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (r < yrows && c + e < ycols) ? Y[r * ldy + c + e] : 0.f;
    u32x4 s1, s2, s3;
    split8(v, s1, s2, s3);
    bf16_t* dst = S + r * ld_s + c;
    *reinterpret_cast<u32x4*>(dst) = s1;
    *reinterpret_cast<u32x4*>(dst + split_stride) = s2;
    *reinterpret_cast<u32x4*>(dst + 2 * split_stride) = s3;
}

// S[s][c][r] = piece s of W[r][c] (the image is W TRANSPOSED) for c < 32 * gridDim.y, r < min(ld_s, 64 * gridDim.x); zero
// outside [m x k].
// A wave takes 64 rows x 8 columns: lane (cb = l & 7, rb = l >> 3) reads W[r0 + 8 rb + i][c0 + cb], i = 0..7, and stores the
// eight row pieces of its column as 16 bytes; the lanes rb = 0..7 of one column fill one whole 128-B line.
__global__ __launch_bounds__(256) void split3_cols_kernel(const float* __restrict__ W, long ldw, long m, int k,
                                                          bf16_t* __restrict__ S, long ld_s, long split_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cb = lane & 7, rb = lane >> 3;
    const long r0 = (long)blockIdx.x * 64 + 8 * rb;
    const int c = blockIdx.y * 32 + wave * 8 + cb;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (r0 + i < m && c < k) ? W[(r0 + i) * ldw + c] : 0.f;
    u32x4 s1, s2, s3;
    split8(v, s1, s2, s3);
    if (r0 >= ld_s) return;                                         // images narrower than 64 rows (H^T with KP = 32)
    bf16_t* dst = S + c * ld_s + r0;
    *reinterpret_cast<u32x4*>(dst) = s1;
    *reinterpret_cast<u32x4*>(dst + split_stride) = s2;
    *reinterpret_cast<u32x4*>(dst + 2 * split_stride) = s3;
}

// ---------------------------------------------------------------------------------------------- small-operand tile in LDS
// [piece 3][row 32*KT][XKT contraction indices = CR 16-byte chunks]; the chunks of a row are XOR-swizzled (128-byte rows:
// (row >> 1) & 7, 64-byte rows: (row >> 2) & 3), which makes the ds_read_b128 of lanes (row = li, same chunk) conflict free
// (as lds_idx of dnmf_nt.h).
constexpr int XK = 64;
struct SplitOperand { const bf16_t* S; long split_stride; long ld; };   // S[piece][row][index]

template <int CR>
__device__ __forceinline__ int tile_swz(int row) { return (row / (16 / CR)) & (CR - 1); }

template <int KT, int CR, int T = 256>
__device__ __forceinline__ void tile_load(u32x4 (&v)[3 * 32 * KT * CR / T], const SplitOperand& o, long k0, int tid) {
    static_assert(3 * 32 * KT * CR % T == 0, "whole 16-byte pieces per thread");
#pragma unroll
    for (int i = 0; i < 3 * 32 * KT * CR / T; ++i) {
        const int p = tid + T * i, s = p / (32 * KT * CR), row = (p / CR) % (32 * KT), ch = p % CR;
        v[i] = *reinterpret_cast<const u32x4*>(o.S + s * o.split_stride + row * o.ld + k0 + ch * 8);
    }
}

template <int KT, int CR, int T = 256>
__device__ __forceinline__ void tile_store(char* tile, const u32x4 (&v)[3 * 32 * KT * CR / T], int tid) {
#pragma unroll
    for (int i = 0; i < 3 * 32 * KT * CR / T; ++i) {
        const int p = tid + T * i, s = p / (32 * KT * CR), row = (p / CR) % (32 * KT), ch = p % CR;
        *reinterpret_cast<u32x4*>(tile + ((s * 32 * KT + row) * CR + (ch ^ tile_swz<CR>(row))) * 16) = v[i];
    }
}

// ---------------------------------------------------------------------------------------------- A H^T (+ fused W update)
// Workgroup = 4 waves x 32 rows of A; wave tile 32 x KP.  A is staged through LDS as fp32 tiles of 128 rows x 32 columns with
// the coalesced loads and the swizzled image of dnmf_nt.h (a lane per row reading its own 64 bytes was measured first: 2.9
// TB/s, the texture path then sees 64 separate lines per load instruction).  A tile is two MFMA steps: step u, lane (li, h)
// contracts over columns 16 u + 8 h + [0, 8) of the tile = 16-B chunks 4 u + 2 h, + 1 of its fp32 row = chunk 2 u + h of the
// H tile [piece 3][row 32*KT][32 indices = 64 B], whose four chunks are swizzled with (row >> 2) & 3.
// NSET tiles are in flight: the loads of tile t + NSET are issued at the top of tile t into the register set that tile t left
// free, tile t + 1 (loaded NSET - 1 tiles ago) goes to the other LDS stage after the MFMAs of tile t.
constexpr int XT = 32;                                                   // contraction indices per NT tile

template <int KT, bool INTERIOR, bool NTX, int NSET, typename TX, int NW = 4>
__device__ __forceinline__ void ntx_mainloop(f32x16 (&acc)[1][KT], const TX* __restrict__ X, long ldx, long nrows, long row0,
                                             const SplitOperand& ys, long cbeg, long cend, float* smem) {
    // bf16-stored X: a tile is the same 128 bytes of every row = 64 contraction indices, kept in LDS as it lies in HBM; a
    // fragment read IS the MFMA operand (no cutting), and a product is three MFMAs (the pieces of H only).
    constexpr bool B16 = std::is_same<TX, bf16_t>::value;
    constexpr int XTI = B16 ? 64 : XT, CR = XTI / 8;                 // indices per tile, 16-byte chunks per H-tile row
    constexpr int BM = 32 * NW, T = 64 * NW;                       // rows and threads of the workgroup
    constexpr int XB = BM * 128, HB = 3 * 32 * KT * CR * 16, STAGE = XB + HB;      // bytes
    char* lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = (int)((cend - cbeg) / XTI);
    if (nk <= 0) return;
    // rotated tile order per workgroup: row tiles are a power-of-two pitch apart (see nt_mainloop_)
    const int kshift = (int)((blockIdx.x * 37u) % (unsigned)nk);
    auto col_of = [&](int t) {
        t = t < nk ? t : nk - 1;                         // past the end: the last tile again (never used)
        t += kshift;
        t = t >= nk ? t - nk : t;
        return cbeg + (long)t * XTI;
    };
    f32x4 xv[NSET][4];                                                // BM * 8 chunks / T threads = 4
    constexpr int NH = 3 * 32 * KT * CR / T;                          // 16-byte pieces of the H tile per thread
    u32x4 hv[NSET][NH];
    auto load = [&](f32x4 (&x)[4], u32x4 (&hh)[NH], int t) {
        const long c0 = col_of(t);
        if constexpr (B16) stage_load_xb<BM, T, true, INTERIOR, NTX>(x, X, ldx, nrows, cend, row0, c0, tid);
        else stage_load<BM, T, true, INTERIOR, NTX>(x, X, ldx, nrows, cend, row0, c0, tid);
        tile_load<KT, CR, T>(hh, ys, c0, tid);
    };
    auto store = [&](char* st, const f32x4 (&x)[4], const u32x4 (&hh)[NH]) {
        stage_store<BM, T>(reinterpret_cast<float*>(st), x, tid);
        tile_store<KT, CR, T>(st + XB, hh, tid);
    };
    const int xrow = wave * 32 + li;
    const int hsw = tile_swz<CR>(li);
    // B fragments of step u, tile jt
    auto hfrag = [&](const char* hc, int u, int jt, u32x4& b1, u32x4& b2, u32x4& b3) {
        const char* f = hc + ((jt * 32 + li) * CR + ((2 * u + h) ^ hsw)) * 16;
        b1 = *reinterpret_cast<const u32x4*>(f);
        b2 = *reinterpret_cast<const u32x4*>(f + 32 * KT * CR * 16);
        b3 = *reinterpret_cast<const u32x4*>(f + 2 * 32 * KT * CR * 16);
    };
    auto compute = [&](const char* st) {
        const float* xc = reinterpret_cast<const float*>(st);
        const char* hc = st + XB;
        if constexpr (B16) {
#pragma unroll
            for (int u = 0; u < XTI / 16; ++u) {
                const u32x4 a1 = *reinterpret_cast<const u32x4*>(&xc[lds_idx(xrow, 2 * u + h)]);
#pragma unroll
                for (int jt = 0; jt < KT; ++jt) {
                    u32x4 b1, b2, b3;
                    hfrag(hc, u, jt, b1, b2, b3);
                    mfma_x3(acc[0][jt], a1, b1, b2, b3, true);
                }
            }
        } else {
            // fp32 A: two steps per tile.  The cut of step 1 (4 pairs x 11 VALU instructions) is issued in four pieces between
            // the MFMAs of step 0, whose products alternate between the KT accumulators; the order is pinned with
            // sched_barrier -- left alone, hipcc emits a step's cut as one block in front of its MFMA chain and the wave's
            // matrix pipe idles through it (MFMA busy 41 %).
            float v0[8], v1[8];
            {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(&xc[lds_idx(xrow, 2 * h)]);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(&xc[lds_idx(xrow, 2 * h + 1)]);
                const f32x4 lo1 = *reinterpret_cast<const f32x4*>(&xc[lds_idx(xrow, 4 + 2 * h)]);
                const f32x4 hi1 = *reinterpret_cast<const f32x4*>(&xc[lds_idx(xrow, 4 + 2 * h + 1)]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = lo[e]; v0[4 + e] = hi[e]; v1[e] = lo1[e]; v1[4 + e] = hi1[e]; }
            }
            u32x4 a1, a2, a3, n1, n2, n3;
            split8(v0, a1, a2, a3);
            u32x4 b[KT][3];
#pragma unroll
            for (int jt = 0; jt < KT; ++jt) hfrag(hc, 0, jt, b[jt][0], b[jt][1], b[jt][2]);
            __builtin_amdgcn_sched_barrier(0);
            // step 0: products p = 0..5 in the order of mfma_x6 (small terms first), round robin over the accumulators
            static_for<0, 6>([&](auto p_) {
                constexpr int pr = decltype(p_)::value;
#pragma unroll
                for (int jt = 0; jt < KT; ++jt) {
                    const u32x4& xa = pr == 0 ? a3 : (pr == 1 || pr == 3) ? a2 : a1;
                    const u32x4& xb = (pr == 0 || pr == 3 || pr == 5) ? b[jt][0] : (pr == 1 || pr == 4) ? b[jt][1] : b[jt][2];
                    acc[0][jt] = mfma_bf16(xa, xb, acc[0][jt]);
                }
                if constexpr (pr < 4) {                              // one pair of step 1 after each of the first four rounds
                    unsigned int s1, s2, s3;
                    split_pair(v1[2 * pr], v1[2 * pr + 1], s1, s2, s3);
                    asm volatile("" : "+v"(s1), "+v"(s2), "+v"(s3));   // (no instruction: keeps LLVM from sinking the cut to its use)
                    n1[pr] = s1; n2[pr] = s2; n3[pr] = s3;
                }
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int jt = 0; jt < KT; ++jt) hfrag(hc, 1, jt, b[jt][0], b[jt][1], b[jt][2]);
            static_for<0, 6>([&](auto p_) {
                constexpr int pr = decltype(p_)::value;
#pragma unroll
                for (int jt = 0; jt < KT; ++jt) {
                    const u32x4& xa = pr == 0 ? n3 : (pr == 1 || pr == 3) ? n2 : n1;
                    const u32x4& xb = (pr == 0 || pr == 3 || pr == 5) ? b[jt][0] : (pr == 1 || pr == 4) ? b[jt][1] : b[jt][2];
                    acc[0][jt] = mfma_bf16(xa, xb, acc[0][jt]);
                }
            });
        }
    };
    static_for<0, NSET>([&](auto i_) { constexpr int i = decltype(i_)::value; load(xv[i], hv[i], i); });
    store(lds, xv[0], hv[0]);
    __syncthreads();
    // tile t: set t % NSET (already in LDS stage t & 1) is refilled with tile t + NSET; set (t + 1) % NSET = tile t + 1 goes to
    // the other stage after the MFMAs.  NSET tiles per trip, nk % NSET == 0 (n % 128 == 0 and one column split: 4 | n / 32,
    // 2 | n / 64).  No `if (t + i < nk)` around the MFMAs: a conditional there makes hipcc keep the accumulators in two
    // register sets and merge them with 32 v_mov per tile behind an MFMA-drain s_nop.
    for (int t = 0; t < nk; t += NSET) {
        static_for<0, NSET>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            load(xv[i], hv[i], t + i + NSET);
            compute(lds + (i & 1) * STAGE);
            store(lds + ((i + 1) & 1) * STAGE, xv[(i + 1) % NSET], hv[(i + 1) % NSET]);
            __syncthreads();
        });
    }
}


// tiles in flight: 4 register sets at KP = 64 with fp32 A (28 registers a set), 2 where a set is 40-64 registers (KP = 128, or
// bf16 A with its 64-index H tiles); bf16 A at KP = 128 stages 128 KiB, i.e. one workgroup per CU, and may use its registers
// NW = waves (32-row groups) per workgroup: 4 in the shipped library.  NW = 6 (192 rows, 166 registers with two register sets
// in flight, three waves per SIMD) exists for A/B runs: it is slower (2.16 vs 1.91 ms at the headline shape).
// (A second main loop -- A cut before LDS into wave-private piece tiles, two tiles in flight, 182 registers -- gave the same bits, 3 %
// faster timed alone and 1 % slower inside the iteration: the pass is bound by the power limit, not by its schedule.  Measured in
// rounds 3 / 4 (profiles/r03d_ntxproto.txt) and removed.)
template <int KT, int MODE, int AUX, typename TX = float, int NW = 4,
          int NSET = ((KT == 2 && std::is_same<TX, float>::value && NW == 4) ? 4 : 2)>
__global__ __launch_bounds__(64 * NW, (KT == 4 && std::is_same<TX, bf16_t>::value) ? 1 : 2) void ntx_kernel(NtArgs p, SplitOperand ys) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = 32 * NW;
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long row0 = (long)blockIdx.x * BM;
    const long cbeg = (long)blockIdx.y * p.cols_per_split;
    long cend = cbeg + p.cols_per_split;
    if (cend > p.ncols) cend = p.ncols;

    f32x16 acc[1][KT];
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][jt][r] = 0.f;

    const TX* X = static_cast<const TX*>(p.X);
    if (row0 + BM <= p.nrows) {
        ntx_mainloop<KT, true, AUX != 0, NSET, TX, NW>(acc, X, p.ldx, p.nrows, row0, ys, cbeg, cend, smem);
    } else ntx_mainloop<KT, false, false, 2, TX, NW>(acc, X, p.ldx, p.nrows, row0, ys, cbeg, cend, smem);

    if constexpr (MODE == NT_STORE) {
        float* out = p.out + (long)blockIdx.y * p.split_stride;
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = row0 + wave * 32 + crow(r, h);
                const int col = jt * 32 + li;
                if (p.store_all || (row < p.nrows && col < p.yrows)) out[row * p.ldo + col] = acc[0][jt][r];
            }
    } else {
        // W[rows] *= acc / (W[rows] G + eps): the k x k product stays on the fp32 path of dnmf_nt.h (dist_nmf.py:731-732)
        f32x16 acc2[1][KT];
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[0][jt][r] = 0.f;
        if (p.wfast) nt_mainloop<KT, 1, NW, 1, true>(acc2, p.W, p.ldw, p.nrows, row0, p.G, 32 * KT, 32 * KT, 0, p.k, smem);
        else nt_mainloop<KT, 1, NW, 1, false>(acc2, p.W, p.ldw, p.nrows, row0, p.G, 32 * KT, 32 * KT, 0, p.k, smem);
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = row0 + wave * 32 + crow(r, h);
                const int col = jt * 32 + li;
                if (row < p.nrows && col < p.k) {
                    const float w = p.W[row * p.ldw + col];
                    p.W[row * p.ldw + col] = w * (acc[0][jt][r] / (acc2[0][jt][r] + p.eps));
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------- W^T A (partial sums per row chunk)
// Workgroup = 4 waves on 4 adjacent blocks of 32 NT columns of one row chunk, sharing the staged W^T tiles (XKT rows of A per
// tile).  MFMA step = 16 rows: lane (g = l & 31, h = l >> 5) reads rows r + 8 h + i (i = 0..7), columns col0 + NT g + [0, NT)
// -- contiguous per row and half wave -- and column j of that block is its B fragment (k = 8 h + i) of output tile j, whose 32
// columns are col0 + NT g' + j.  acc[kt][j] (reg, lane) = C[kt * 32 + crow(reg, h)][col0 + NT li + j]: the NT tiles of a lane
// store as one vector.  NT = 4, XKT = 64 at KP = 64; KP = 128 has 4 x NT accumulator tiles, so NT = 2 there, and 32-row
// tiles keep two workgroups per CU.
template <int KT, int NT, int XKT, int AUX, typename TX = float>
__global__ __launch_bounds__(256, 2) void tnx_kernel(TnArgs p, SplitOperand ws) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    constexpr int CR = XKT / 8, NST = XKT / 16, TB = 3 * 32 * KT * CR * 16, CW = 32 * NT;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb4 = (p.ncolblk + 3) / 4;
    const long chunk = blockIdx.x / cb4;
    int colblk = (int)(blockIdx.x % cb4) * 4 + wave;
    const bool live = colblk < p.ncolblk;
    colblk = live ? colblk : p.ncolblk - 1;
    const long col0 = (long)colblk * CW;
    const long rbeg = chunk * p.rows_per_chunk;
    long rend = rbeg + p.rows_per_chunk;
    if (rend > p.nrows) rend = p.nrows;
    const int nch = (int)((rend - rbeg + XKT - 1) / XKT);
    // bf16-stored A: the lane's 8 x NT block arrives as NT / 2 dwords per row; column j of it, packed pairwise over the
    // rows, IS the B fragment (no cutting), and a product is three MFMAs (the pieces of W^T only)
    constexpr bool B16 = std::is_same<TX, bf16_t>::value;
    constexpr int EB = B16 ? 2 : 4, XW = B16 ? NT / 2 : NT;           // bytes per element, dwords per row and lane
    const TX* A = static_cast<const TX*>(p.Y) + rbeg * p.ldy + col0;
    const long left = ((p.nrows - rbeg) * p.ldy - col0) * EB;         // rows past the end of A read as zeros
    i32x4 rs = buf_rsrc(A);
    rs[2] = __builtin_amdgcn_readfirstlane((int)(left < 0x7fffffffL ? left : 0x7fffffffL));
    const int voff = (int)((8 * h * p.ldy + NT * li) * EB);
    const int rowb = (int)(p.ldy * EB);

    f32x16 acc[KT][NT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[kt][j][r] = 0.f;

    // two MFMA steps of A in flight per wave (the loads of step g + 2 are issued when step g has been consumed): with one, a
    // wave computes 48 MFMAs (0.8 us) and then waits out the rest of an HBM latency -- SQ_WAIT_ANY 46 % of the wave cycles
    float xr[2][8][XW];
    u32x4 wv[3 * KT * CR / 8];
    const int nsteps = NST * nch;
    auto issue = [&](float (&x)[8][XW], int g) {                   // rows of step g (past the end: the last step again, unused)
        g = g < nsteps ? g : nsteps - 1;
        const int sbase = g * 16 * rowb;
        static_for<0, 8>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            buf_load<XW, AUX>(x[i], rs, voff, sbase + i * rowb);
        });
    };
    issue(xr[0], 0);
    issue(xr[1], 1);
    tile_load<KT, CR>(wv, ws, rbeg, tid);
    tile_store<KT, CR>(lds, wv, tid);
    __syncthreads();
    int fo[NST];             // fragment of step st: row li, chunk 2 st + h
#pragma unroll
    for (int st = 0; st < NST; ++st) fo[st] = (li * CR + ((2 * st + h) ^ tile_swz<CR>(li))) * 16;
    static_assert(NST % 2 == 0, "the two register sets alternate with the step parity");
    for (int c = 0; c < nch; ++c) {
        const int cn = c + 1 < nch ? c + 1 : c;
        const char* cur = lds + (c & 1) * TB;
        tile_load<KT, CR>(wv, ws, rbeg + (long)cn * XKT, tid);
        static_for<0, NST>([&](auto st_) {
            constexpr int st = decltype(st_)::value;
            float (&x)[8][XW] = xr[st & 1];
            u32x4 a[KT][3];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const char* f = cur + fo[st] + kt * 32 * CR * 16;
                a[kt][0] = *reinterpret_cast<const u32x4*>(f);
                a[kt][1] = *reinterpret_cast<const u32x4*>(f + 32 * KT * CR * 16);
                a[kt][2] = *reinterpret_cast<const u32x4*>(f + 2 * 32 * KT * CR * 16);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                u32x4 b[B16 ? 1 : 3];
                if constexpr (B16) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        b[0][q] = __builtin_amdgcn_perm(__float_as_uint(x[2 * q + 1][j >> 1]), __float_as_uint(x[2 * q][j >> 1]),
                                                        (j & 1) ? 0x07060302u : 0x05040100u);
                } else {
                    const float v[8] = {x[0][j], x[1][j], x[2][j], x[3][j], x[4][j], x[5][j], x[6][j], x[7][j]};
                    split8(v, b[0], b[1], b[2]);
                }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    if constexpr (B16) mfma_x3(acc[kt][j], b[0], a[kt][0], a[kt][1], a[kt][2], false);
                    else mfma_x6(acc[kt][j], a[kt][0], a[kt][1], a[kt][2], b[0], b[1], b[2]);
                }
            }
            issue(x, NST * c + st + 2);
        });
        tile_store<KT, CR>(lds + ((c + 1) & 1) * TB, wv, tid);
        __syncthreads();
    }

    if (live) {
        float* Pc = p.P + chunk * p.chunk_stride;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float d[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) d[j] = acc[kt][j][r];
                store_vec<NT, true>(d, Pc + (long)(kt * 32 + crow(r, h)) * p.ldp, col0 + (long)NT * li, p.ldp, true);
            }
    }
}

}  // namespace
