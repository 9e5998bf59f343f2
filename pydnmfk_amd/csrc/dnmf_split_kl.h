// dnmf_split_kl.h -- the two KL products (dist_nmf.py:806-810) with every matrix product as six bf16 piece products (bf16x6,
// arithmetic: dnmf_split.h).  Part of libdnmf_hip.so (translation unit csrc/dnmf_split.hip).
//
// Both kernels keep the register-resident structure of the fp32 ones (dnmf_nn.h): S = W H is formed tile-wise in MFMA
// accumulators, turned into U = A / (S + eps) in place, and U -- cut into three bf16 pieces in registers -- is the operand
// of the second product without leaving the wave.  The fp32 kernels are bound by v_mfma_f32_32x32x2_f32 at every rank (two
// products per element of A: 0.87 ms per kernel on 32768 x 16384 at k <= 32, where the HBM needs 0.39 ms); with six bf16
// MFMAs per 16 contraction indices the matrix work drops to 3/8 of that.
//
// Operands.  The factors are cut once per call into bf16 images in the workspace (zero padded to KP and to whole tiles):
//   wp [piece][m_pad][KP]  W, rows as in memory;   ht [piece][n][KP]  H transposed
// i.e. both with the rank index contiguous: a fragment of S = W H is 8 consecutive k of a row (one 16-byte read).  The second
// product contracts over the rows i of W (W^T U) or the columns c of H (U H^T), i.e. over the ROWS of these images.  The C/D
// layout of a 32 x 32 tile puts rows 4 h + (r & 3) + 8 (r >> 2) into register r of lane half h: registers 8 s..8 s + 7 of a
// lane are the rows 16 s + 4 h + {0..3} and 16 s + 8 + 4 h + {0..3}, so the matching fragment of the other operand is twice
// "4 consecutive rows of one k" -- two ds_read_b64_tr_b16 of the image's LDS copy (the contraction index may be permuted
// as long as both operands agree).
#pragma once
#include "dnmf_split.h"

namespace {

struct KlxArgs {
    const float* A; long lda; long m; long n; float eps;
    int k;                                          // rank before padding: S = W H skips the steps whose 16 indices are all padding
    SplitOperand wp, ht;
    float* P; long chunk_stride; long ldp;          // W^T U partial slabs [chunk][KP][ldp]
    long nrowblk; int ncolblk; long rowblks_per_chunk; long nchunks;
    float* out; long ldo; long split_stride; long cols_per_split; int out_cols;   // U H^T (column split slabs)
};

__device__ __forceinline__ u32x4 ld16(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ u32x4 ld8x2(const bf16_t* p, long second) {           // elements [0, 4) and [second, second + 4)
    const u32x2 lo = *reinterpret_cast<const u32x2*>(p), hi = *reinterpret_cast<const u32x2*>(p + second);
    return u32x4{lo[0], lo[1], hi[0], hi[1]};
}

// (div_pos, the division of U = A / (S + eps): dnmf_common.h)

// swizzle key of a tile row whose length is CR 16-byte chunks (CR = 4, 8: 64- and 128-byte rows), see lds_idx / htile_store
template <int CR>
__device__ __forceinline__ int swz(int row) { return (row / (16 / CR)) & (CR - 1); }

// ---------------------------------------------------------------------------------------------- W^T U  (H-side numerator)
// Workgroup = 4 waves on one block of 32 NT columns, each walking its own chunk of 32-row blocks (as kl_wtu_kernel).  The
// block's H^T tile [piece][32 NT columns][KP] is staged once; lane li owns columns col0 + NT li + ne, tile row ne * 32 + li.
//
// The W rows of a 32-row block (3 pieces x 32 rows x KP bf16, contiguous in the wp image) are fetched with fully coalesced
// 16-byte loads and parked in a wave-private LDS image [piece][row][KP]; they serve BOTH products from there: S = W H reads a
// row's 8 consecutive k (ds_read_b128), W^T U needs 4 consecutive ROWS of one k per lane, which is what the transposing read
// ds_read_b64_tr_b16 delivers from the same image.  (A first version took the second kind from a transposed global image,
// 8 bytes per lane from 32 different lines per instruction: the texture path, one lane per clock on such loads, set the pace --
// 0.71 ms per 32768 x 16384 pass.)  The image's 16-byte chunks are XOR-swizzled per row so that the row writes, the row
// reads and the transposed reads (whole 64-byte half rows of 4 consecutive rows per 32 lanes) spread over the banks.
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KT>
__device__ __forceinline__ int wblk_off(int row, int chunk) {       // byte offset of 16-byte chunk `chunk` of row `row`
    if constexpr (KT == 1) return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);      // (row reads go in the lane groups of MI355X_MICROARCH.md, LDS table)
    else if constexpr (KT == 2) return row * 128 + ((chunk ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))) << 4);
    else return row * 256 + ((chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);   // cdna_hip_programming.md T10, image (b)
}

__device__ __forceinline__ u32x2 lds_tr16(const char* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(u32x2, v);
}

template <int KT, int NT>
__global__ __launch_bounds__(256, KT == 4 ? 1 : 2) void klx_wtu_kernel(KlxArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    constexpr int KP = 32 * KT, CW = 32 * NT, CR = 4 * KT;           // CR chunks of 8 k per row
    constexpr int HT_BYTES = 3 * CW * CR * 16, WB_PIECE = 32 * KP * 2, WB_BYTES = 3 * WB_PIECE;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long colblk = blockIdx.x % p.ncolblk;
    const long chunk = (blockIdx.x / p.ncolblk) * 4 + wid;
    const long col0 = colblk * CW;
    for (int idx = tid; idx < 3 * CW * CR; idx += 256) {
        const int s = idx / (CW * CR), x = (idx / CR) % CW, c = idx % CR;
        const int L = (x % NT) * 32 + x / NT;
        *reinterpret_cast<u32x4*>(lds + ((s * CW + L) * CR + (c ^ swz<CR>(L))) * 16) =
            ld16(p.ht.S + s * p.ht.split_stride + (col0 + x) * KP + c * 8);
    }
    __syncthreads();
    // no early return: ds_read_b64_tr_b16 needs every lane, and a wave without a chunk simply has no blocks to walk
    char* wb = lds + HT_BYTES + wid * WB_BYTES;                       // this wave's W block image

    f32x16 out[KT][NT];
#pragma unroll
    for (int ke = 0; ke < KT; ++ke)
#pragma unroll
        for (int ne = 0; ne < NT; ++ne)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[ke][ne][r] = 0.f;
    const long rb0 = chunk < p.nchunks ? chunk * p.rowblks_per_chunk : 0;
    long rb1 = chunk < p.nchunks ? rb0 + p.rowblks_per_chunk : 0;
    if (rb1 > p.nrowblk) rb1 = p.nrowblk;
    // A through a buffer descriptor that ends with the matrix: rows >= m read as 0 (U = 0 / (0 + eps) = 0)
    const float* Ab = p.A + rb0 * 32 * p.lda + col0;
    const long left = ((p.m - rb0 * 32) * p.lda - col0) * 4;
    i32x4 rs = buf_rsrc(Ab);
    rs[2] = __builtin_amdgcn_readfirstlane((int)(left < 0x7fffffffL ? left : 0x7fffffffL));
    const int voff = (int)((4 * h * p.lda + NT * li) * 4);
    const int rowb = (int)(p.lda * 4);
    const int key = swz<CR>(li);                                    // H^T tile rows ne * 32 + li share the key of li
    // addresses of this lane in the W block image: row reads (row li, chunk 2 s + h) and transposed reads (it SUPPLIES row
    // 4 h + q (+ 8, 16, 24), columns ke * 32 + 16 (g & 1) + 4 p.. of its 16-lane group g = lane >> 4; q = (lane >> 2) & 3,
    // p = lane & 3) and RECEIVES column ke * 32 + li of those four rows)
    int rd[2 * KT], tr[KT][4];
#pragma unroll
    for (int s = 0; s < 2 * KT; ++s) rd[s] = wblk_off<KT>(li, 2 * s + h);
    {
        const int q = (lane >> 2) & 3, pp = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int ke = 0; ke < KT; ++ke)
#pragma unroll
            for (int j = 0; j < 4; ++j) tr[ke][j] = wblk_off<KT>(4 * h + q + 8 * j, 4 * ke + 2 * g1 + (pp >> 1)) + 8 * (pp & 1);
    }

    float areg[16][NT];
    u32x4 wst[3][2 * KT];                                            // the next W block as loaded: 16-byte unit j * 64 + lane of a piece
    auto issue_block = [&](long rb) {
        const int sb = (int)((rb - rb0) * 32) * rowb;
        static_for<0, 16>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            buf_load<NT, 2>(areg[r], rs, voff, sb + ((r & 3) + 8 * (r >> 2)) * rowb);
        });
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < 2 * KT; ++j) wst[q][j] = ld16(p.wp.S + q * p.wp.split_stride + rb * 32 * KP + (j * 64 + lane) * 8);
    };
    if (rb0 < rb1) issue_block(rb0);
    for (long rb = rb0; rb < rb1; ++rb) {
        const long nxt = rb + 1 < rb1 ? rb + 1 : rb;               // past the end: this block again (unused)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < 2 * KT; ++j) {
                const int t = j * 64 + lane;
                *reinterpret_cast<u32x4*>(wb + q * WB_PIECE + wblk_off<KT>(t / CR, t % CR)) = wst[q][j];
            }
        f32x16 S[NT];
#pragma unroll
        for (int ne = 0; ne < NT; ++ne)
#pragma unroll
            for (int r = 0; r < 16; ++r) S[ne][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 2 * KT; ++s) {                          // phase 1: S = W H
            if (16 * s >= p.k) break;                               // (uniform; the loop is unrolled: a forward branch per step)
            const u32x4 a1 = *reinterpret_cast<const u32x4*>(wb + rd[s]);
            const u32x4 a2 = *reinterpret_cast<const u32x4*>(wb + WB_PIECE + rd[s]);
            const u32x4 a3 = *reinterpret_cast<const u32x4*>(wb + 2 * WB_PIECE + rd[s]);
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) {
                const char* f = lds + ((ne * 32 + li) * CR + ((2 * s + h) ^ key)) * 16;
                const u32x4 b1 = *reinterpret_cast<const u32x4*>(f);
                const u32x4 b2 = *reinterpret_cast<const u32x4*>(f + CW * CR * 16);
                const u32x4 b3 = *reinterpret_cast<const u32x4*>(f + 2 * CW * CR * 16);
                mfma_x6(S[ne], a1, a2, a3, b1, b2, b3);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)                                // phase 2: U = A / (S + eps)  (dist_nmf.py:806)
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) S[ne][r] = div_pos(areg[r][ne], S[ne][r] + p.eps);
        issue_block(nxt);                                           // both register sets are free: the next block's loads fly under phase 3
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {                            // phase 3: out += W^T U
            u32x4 w3[KT][3];
#pragma unroll
            for (int ke = 0; ke < KT; ++ke)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const u32x2 lo = lds_tr16(wb + q * WB_PIECE + tr[ke][2 * s2]);
                    const u32x2 hi = lds_tr16(wb + q * WB_PIECE + tr[ke][2 * s2 + 1]);
                    w3[ke][q] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) {
                const float v[8] = {S[ne][8 * s2], S[ne][8 * s2 + 1], S[ne][8 * s2 + 2], S[ne][8 * s2 + 3],
                                    S[ne][8 * s2 + 4], S[ne][8 * s2 + 5], S[ne][8 * s2 + 6], S[ne][8 * s2 + 7]};
                u32x4 u1, u2, u3;
                split8(v, u1, u2, u3);
#pragma unroll
                for (int ke = 0; ke < KT; ++ke) mfma_x6(out[ke][ne], w3[ke][0], w3[ke][1], w3[ke][2], u1, u2, u3);
            }
        }
    }
    if (chunk >= p.nchunks) return;
    float* Pc = p.P + chunk * p.chunk_stride;
#pragma unroll
    for (int ke = 0; ke < KT; ++ke)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float d[NT];
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) d[ne] = out[ke][ne][r];
            store_vec<NT, true>(d, Pc + (long)(ke * 32 + crow(r, h)) * p.ldp, col0 + (long)NT * li, p.ldp, true);
        }
}

// ---------------------------------------------------------------------------------------------- U H^T  (W-side numerator)
// Workgroup = 4 waves x 32 rows of A, walking 32-column tiles of its column split (as kl_uht_kernel): S is formed TRANSPOSED,
// S^T[c][i] = sum_k H[k][c] W[i][k] (A operand: the H^T tile, B operand: the lane's own W row, held in registers), so that
// the C/D registers run over the columns c of the tile and U^T is the B operand of (U H^T)^T[j][i] = sum_c H[j][c] U^T[c][i].
// That second product wants H[j][4 consecutive c] per lane: the transposing read of the SAME H^T tile [piece][c][KP] (image
// wblk_off, as the W block of klx_wtu_kernel).  A is staged through LDS with coalesced loads (a lane per row would read 16
// bytes of 64 different lines per instruction, see ntx_kernel): stage = [A 128 x 32 fp32, swizzled as lds_idx][H^T tile].
// NSET tiles are in flight in registers (see ntx_mainloop).
template <int KT>
struct UhtStage {
    static constexpr int CR = 4 * KT, KP = 32 * KT;
    static constexpr int A_BYTES = 128 * 32 * 4, HT_PIECE = 32 * KP * 2, BYTES = A_BYTES + 3 * HT_PIECE;
    static constexpr int NHT = (3 * 32 * CR + 255) / 256;          // 16-byte pieces per thread
};

template <int KT, bool INTERIOR, int NSET>
__device__ __forceinline__ void klx_uht_body(const KlxArgs& p, float* smem) {
    using St = UhtStage<KT>;
    constexpr int KP = 32 * KT, CR = 4 * KT;
    char* lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long row0 = (long)blockIdx.x * 128;
    const long arow = row0 + wave * 32 + li;
    const long cbeg = (long)blockIdx.y * p.cols_per_split;
    long cend = cbeg + p.cols_per_split;
    if (cend > p.n) cend = p.n;
    const int nt = (int)((cend - cbeg) / 32);

    f32x16 out[KT];
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[jt][r] = 0.f;
    u32x4 wr[2 * KT][3];                                            // W[arow][16 s + 8 h + (0..7)], three pieces
#pragma unroll
    for (int s = 0; s < 2 * KT; ++s)
#pragma unroll
        for (int q = 0; q < 3; ++q) wr[s][q] = ld16(p.wp.S + q * p.wp.split_stride + arow * KP + 16 * s + 8 * h);

    f32x4 xv[NSET][4];
    u32x4 hv[NSET][St::NHT];
    auto load = [&](f32x4 (&xa)[4], u32x4 (&hta)[St::NHT], int t) {
        t = t < nt ? t : nt - 1;                                    // past the end: the last tile again (never used)
        const long c0 = cbeg + (long)t * 32;
        stage_load<128, 256, true, INTERIOR, true>(xa, p.A, p.lda, p.m, cend, row0, c0, tid);
#pragma unroll
        for (int i = 0; i < St::NHT; ++i) {
            const int idx = tid + 256 * i;
            if (3 * 32 * CR % 256 == 0 || idx < 3 * 32 * CR) {
                const int s = idx / (32 * CR), u = idx % (32 * CR);
                hta[i] = ld16(p.ht.S + s * p.ht.split_stride + c0 * KP + u * 8);
            }
        }
    };
    auto store = [&](char* st, const f32x4 (&xa)[4], const u32x4 (&hta)[St::NHT]) {
        stage_store<128, 256>(reinterpret_cast<float*>(st), xa, tid);
#pragma unroll
        for (int i = 0; i < St::NHT; ++i) {
            const int idx = tid + 256 * i;
            if (3 * 32 * CR % 256 == 0 || idx < 3 * 32 * CR) {
                const int s = idx / (32 * CR), u = idx % (32 * CR);
                *reinterpret_cast<u32x4*>(st + St::A_BYTES + s * St::HT_PIECE + wblk_off<KT>(u / CR, u % CR)) = hta[i];
            }
        }
    };
    if (nt <= 0) return;
    const int xrow = wave * 32 + li;
    int rd[2 * KT], tr[KT][4];                                       // row reads / transposed reads of the H^T tile (see klx_wtu_kernel)
#pragma unroll
    for (int s = 0; s < 2 * KT; ++s) rd[s] = wblk_off<KT>(li, 2 * s + h);
    {
        const int q = (lane >> 2) & 3, pp = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int j = 0; j < 4; ++j) tr[jt][j] = wblk_off<KT>(4 * h + q + 8 * j, 4 * jt + 2 * g1 + (pp >> 1)) + 8 * (pp & 1);
    }
    auto compute = [&](const char* cur) {
        const float* xc = reinterpret_cast<const float*>(cur);
        const char* htc = cur + St::A_BYTES;
        f32x16 st;                                                   // S^T tile: registers = columns c, lanes = rows i
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 2 * KT; ++s) {
            if (16 * s >= p.k) break;                               // steps of pure padding (uniform)
            const u32x4 a1 = *reinterpret_cast<const u32x4*>(htc + rd[s]);
            const u32x4 a2 = *reinterpret_cast<const u32x4*>(htc + St::HT_PIECE + rd[s]);
            const u32x4 a3 = *reinterpret_cast<const u32x4*>(htc + 2 * St::HT_PIECE + rd[s]);
            mfma_x6(st, a1, a2, a3, wr[s][0], wr[s][1], wr[s][2]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {                               // U^T = A / (S^T + eps): register 4 g + e = column 8 g + 4 h + e
            const f32x4 a = *reinterpret_cast<const f32x4*>(&xc[lds_idx(xrow, 2 * g + h)]);
#pragma unroll
            for (int e = 0; e < 4; ++e) st[4 * g + e] = div_pos(a[e], st[4 * g + e] + p.eps);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const float v[8] = {st[8 * s2], st[8 * s2 + 1], st[8 * s2 + 2], st[8 * s2 + 3],
                                st[8 * s2 + 4], st[8 * s2 + 5], st[8 * s2 + 6], st[8 * s2 + 7]};
            u32x4 u1, u2, u3;
            split8(v, u1, u2, u3);
#pragma unroll
            for (int jt = 0; jt < KT; ++jt) {
                u32x4 a[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const u32x2 lo = lds_tr16(htc + q * St::HT_PIECE + tr[jt][2 * s2]);
                    const u32x2 hi = lds_tr16(htc + q * St::HT_PIECE + tr[jt][2 * s2 + 1]);
                    a[q] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                }
                mfma_x6(out[jt], a[0], a[1], a[2], u1, u2, u3);
            }
        }
    };
    static_for<0, NSET>([&](auto i_) { constexpr int i = decltype(i_)::value; load(xv[i], hv[i], i); });
    store(lds, xv[0], hv[0]);
    __syncthreads();
    // tile t: register set t % NSET (already in LDS stage t & 1) is refilled with tile t + NSET, set (t + 1) % NSET = tile t + 1
    // goes to the other stage after the MFMAs.  nt % NSET == 0: the host cuts the columns into multiples of 64 (plan_uhtx); a
    // conditional around the MFMAs would cost an accumulator copy per tile (see ntx_mainloop).
    for (int t0 = 0; t0 < nt; t0 += NSET) {
        static_for<0, NSET>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            load(xv[i], hv[i], t0 + i + NSET);
            compute(lds + (i & 1) * St::BYTES);
            store(lds + ((i + 1) & 1) * St::BYTES, xv[(i + 1) % NSET], hv[(i + 1) % NSET]);
            __syncthreads();
        });
    }
    // out[jt] (reg, lane): j = jt * 32 + crow(reg, h), i = arow; registers 4 g..4 g + 3 are four consecutive j
    if (arow < p.m) {
        float* dst = p.out + (long)blockIdx.y * p.split_stride + arow * p.ldo;
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int j = jt * 32 + 8 * g + 4 * h;
                if (j + 4 <= p.out_cols) *reinterpret_cast<f32x4*>(dst + j) = f32x4{out[jt][4 * g], out[jt][4 * g + 1], out[jt][4 * g + 2], out[jt][4 * g + 3]};
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (j + e < p.out_cols) dst[j + e] = out[jt][4 * g + e];
            }
    }
}

template <int KT>
__global__ __launch_bounds__(256, KT == 4 ? 1 : 2) void klx_uht_kernel(KlxArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NSET = 2;      // (4 sets at KT = 1: 229 registers, two waves per SIMD instead of three, no faster)
    if (((long)blockIdx.x + 1) * 128 <= p.m) klx_uht_body<KT, true, NSET>(p, smem);
    else klx_uht_body<KT, false, 2>(p, smem);
}

}  // namespace
