// dnmf_kl16.h -- the two KL products for rank k <= 16 on v_mfma_f32_16x16x4_f32 (dist_nmf.py:806-810; 2D :311-312, :337-338).
// Part of libdnmf_hip.so (kernels live in anonymous namespaces of the headers; the translation units csrc/*.hip include what they launch).
//
// KL is the reference's default objective (pyDNMF.py:70) and its NMFk examples sweep small ranks (k = 14..18 on swim).  The
// 32-wide kernels of dnmf_nn.h pad the rank to 32: per 32 x 32 tile of A they issue 8 + 16 v_mfma_f32_32x32x2_f32 (1536
// matrix-pipe cycles per 4 KiB of A; the S = W H product skips its all-padding steps, the second product cannot) and stay
// bound by the matrix pipe at 36-38 % of the HBM rate.  On the 16x16x4 instruction a 16 x 16 tile of A costs 4 + 4 MFMAs of
// 32 cycles: 1024 cycles per 4 KiB, which lets the pass run at the memory rate.  Operand maps (lane l, i16 = l & 15,
// kq = l >> 4):   A-operand A[i = i16][kk = kq]     B-operand B[kk = kq][j = i16]     C/D register r: C[4 kq + r][i16].
// The contraction order inside a product is free, so rank index jj = 4 kq + step: a lane's four A/B values of the S product
// are ONE 16-byte piece of a factor row.
//
// Both kernels require factors of exactly 16 columns / rows with 16-byte aligned rows (the host passes zero-padded images
// for k < 16, dnmf_kl.hip pad_factors: zero columns of W and zero rows of H contribute nothing to either product), A with
// 16-byte aligned rows, and a column count that is a whole number of tiles; every row index is clamped into range (the
// outputs of clamped rows are never stored, their contribution to a column sum is zeroed), so there are no edge branches.
//
// What bounds these kernels (tools/coissue.hip, measured): on gfx950 an fp32 VALU instruction does NOT overlap with fp32
// MFMAs of the same SIMD -- whichever wave issues it, it costs about 3.3 cycles of matrix-pipe time (the fp32-input MFMA
// runs at the fp32 vector rate) -- so the time of a 16 x 16 tile is 8 x 32 MFMA cycles + 3.3 x the vector instructions per
// lane.  The first version spent 41 vector instructions per tile and lane (7 per division, accumulator <-> VGPR copies,
// 64-bit address arithmetic): 53-55 % MFMA busy.  Hence: (i) this header is compiled in a translation unit of its own with
// -mllvm -amdgpu-mfma-vgpr-form (csrc/dnmf_kl16.hip, pydnmfk_amd/build.py): the accumulators live in VGPRs and the
// division reads them in place, no v_accvgpr copies; (ii) eps is the INITIAL value of the S accumulators (the C operand of
// the first MFMA of a chain), so S + eps costs nothing; (iii) the quotient is reciprocal times numerator (2
// instructions, <= 1.5 ulp, against the 1e-5 parity budget of a step: div16 below); (iv) global addresses are MUBUF
// descriptor + a loop-invariant lane offset + a scalar offset: no vector address arithmetic in the loops.
#pragma once
#include "dnmf_common.h"
#include "dnmf_nt.h"
#include "dnmf_k16.h"

namespace {

struct Kl16Args {
    const float* A; long lda; long m; long n;
    const float* W; long ldw;            // [m x 16]
    const float* H; long ldh;            // [16 x n]
    float eps;
    float* P; long chunk_stride; long ldp;   // partial slabs: wtu16 [chunk][16][ldp], uht16 [split][m][16]
    long rows_per_chunk; int nchunks; int ncolblk;   // wtu16 (rows_per_chunk x lda x 4 B below 2 GiB: one descriptor per chunk)
    long cols_per_split;                             // uht16 (multiple of 32)
    long hblk; long hextra;                          // uht16: H as column blocks [n / hblk][16][hblk] (see NnArgs in dnmf_nn.h); 0 = plain
};
__device__ __forceinline__ void rebase_args(Kl16Args& p, const BatchTab& bt) {
    rebase(p.A, bt); rebase(p.W, bt); rebase(p.H, bt); rebase(p.P, bt);
}

// a / d for d = S + eps > 0: v_rcp_f32 (1 ulp) times a -- at most 1.5 ulp from the exact quotient.  Every further vector
// instruction costs 3.3 matrix-pipe cycles per lane here (the residual correction that makes the quotient correctly rounded
// is two more: measured 480 -> 435 us on U H^T, 32768 x 16384); U is summed over thousands of rows or columns right after,
// where this error is far below the fp32 summation error of the sum itself, and the step's parity budget is 1e-5.
__device__ __forceinline__ float div16(float a, float d) { return a * __builtin_amdgcn_rcpf(d); }

// ================================================================================================ W^T U, k <= 16 (H side)
// P[chunk][j][c] = sum_{i in chunk} W[i][j] A[i][c] / (S[i][c] + eps),  S = W H.
// One wave owns 64 columns (lane i16 owns the four columns col0 + 4 i16 + ne: one 16-byte piece of every row of A -- a
// row's 256 bytes are contiguous across 16 lanes, an instruction covers four rows) and walks a chunk of 16-row blocks.  The
// 16 x 64 block of H those columns need is loop invariant: 16 registers per lane, no LDS.  Per block: S in C/D layout
// (register r <-> row 4 kq + r, lane <-> column), U = A / (S + eps) in place, and U is then the B operand of W^T U as it
// stands (the contraction index -- the row -- is the C/D row index, i.e. kk = kq with step r).  Two blocks in flight: the
// loads of block b+1 are issued before the products of block b.
__global__ __launch_bounds__(256) void kl_wtu16_kernel(Kl16Args p, BatchTab bt) {
    rebase_args(p, bt);
    const int lane = threadIdx.x & 63, i16 = lane & 15, kq = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long gw = (long)blockIdx.x * 4 + wid;
    const long chunk = gw / p.ncolblk, colblk = gw % p.ncolblk;
    if (chunk >= p.nchunks) return;
    const long col0 = colblk * 64 + 4 * i16;
    const long rbeg = chunk * p.rows_per_chunk;
    long rend = rbeg + p.rows_per_chunk;
    if (rend > p.m) rend = p.m;
    const long last = p.m - 1;

    f32x4 hreg[4];                                         // hreg[s][ne] = H[4 kq + s][col0 + ne]
#pragma unroll
    for (int s = 0; s < 4; ++s) hreg[s] = *reinterpret_cast<const f32x4*>(p.H + (long)(4 * kq + s) * p.ldh + col0);
    f32x4 out[4];                                          // out[ne][r] = WTU[j = 4 kq + r][col0 + ne]
#pragma unroll
    for (int ne = 0; ne < 4; ++ne) out[ne] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 epsv = {p.eps, p.eps, p.eps, p.eps};

    struct Blk { f32x4 a[4]; f32x4 wa; float wb[4]; };
    // descriptors at the chunk's first row; lane offsets (bytes): A[4 kq + r][col0], W[i16][4 kq], W[4 kq + r][i16]
    const i32x4 rsa = buf_rsrc(p.A + rbeg * p.lda + colblk * 64);
    const i32x4 rsw = buf_rsrc(p.W + rbeg * p.ldw);
    const int lda4 = (int)(p.lda * 4), ldw4 = (int)(p.ldw * 4);
    const int va = 4 * kq * lda4 + 16 * i16, vwa = i16 * ldw4 + 16 * kq, vwb = 4 * kq * ldw4 + 4 * i16;
    auto ld = [&](Blk& b, int blk) {                      // a block whose 16 rows are all inside the chunk: nothing but loads
        b.wa = buf_ld_f32x4(rsw, vwa, blk * 16 * ldw4, 0);                               // S product: W[row0 + i16][4 kq + s]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            b.a[r] = buf_ld_f32x4(rsa, va, (blk * 16 + r) * lda4, 2);                    // nontemporal: A is read once
            b.wb[r] = buf_ld_f32(rsw, vwb, (blk * 16 + r) * ldw4, 0);                    // second product: W[row][j = i16]
        }
    };
    auto mm = [&](const Blk& b) {
        f32x4 s[4];
#pragma unroll
        for (int ne = 0; ne < 4; ++ne) s[ne] = MFMA16(b.wa[0], hreg[0][ne], epsv);       // S + eps: eps is the initial value
#pragma unroll
        for (int st = 1; st < 4; ++st)
#pragma unroll
            for (int ne = 0; ne < 4; ++ne) s[ne] = MFMA16(b.wa[st], hreg[st][ne], s[ne]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ne = 0; ne < 4; ++ne) s[ne][r] = div16(b.a[r][ne], s[ne][r]);        // U (dist_nmf.py:806)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ne = 0; ne < 4; ++ne) out[ne] = MFMA16(b.wb[r], s[ne][r], out[ne]);
    };
    Blk b0, b1;
    const int nb = (int)((rend - rbeg) / 16);              // whole blocks; a ragged tail (last chunk only) follows
    if (nb > 0) {
        ld(b0, 0);
        int ib = 0;
        for (; ib + 2 <= nb; ib += 2) {
            ld(b1, ib + 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(b0);
            __builtin_amdgcn_sched_barrier(0);
            ld(b0, ib + 2 < nb ? ib + 2 : ib);                // past the end: re-read (unused)
            __builtin_amdgcn_sched_barrier(0);
            mm(b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ib < nb) mm(b0);
    }
    if (rbeg + (long)nb * 16 < rend) {                     // ragged tail: rows clamped into range, their W^T entries zeroed
        const long row0 = rbeg + (long)nb * 16;
        Blk t;
        long ra = row0 + i16;
        ra = ra < last ? ra : last;
        t.wa = *reinterpret_cast<const f32x4*>(p.W + ra * p.ldw + 4 * kq);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = row0 + 4 * kq + r;
            const long rc = row < last ? row : last;
            t.a[r] = *reinterpret_cast<const f32x4*>(p.A + rc * p.lda + col0);
            t.wb[r] = row < rend ? p.W[rc * p.ldw + i16] : 0.f;
        }
        mm(t);
    }
    float* Pc = p.P + chunk * p.chunk_stride + col0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        *reinterpret_cast<f32x4*>(Pc + (long)(4 * kq + r) * p.ldp) = f32x4{out[0][r], out[1][r], out[2][r], out[3][r]};
}

// ================================================================================================ U H^T, k <= 16 (W side)
// P[split][i][j] = sum_{c in split} (A[i][c] / (S[i][c] + eps)) H[j][c].
// The skeleton of nt16_kernel: workgroup = 4 waves x 32 rows (two 16-row sub-tiles per wave); tiles of 32 columns of A
// (128 x 128 B) and of H (16 x 128 B) go global -> VGPR -> LDS with coalesced 16-byte loads into the XOR-swizzled images of
// the NT kernels, two tiles in flight, branch free.  The fragment a lane reads from the A image -- the 16-byte piece
// (4 s + kq) of row i16 -- IS the C/D layout of the TRANSPOSED product S^T[c][i] (register e <-> column 16 s + 4 kq + e,
// lane <-> row), so S^T = H^T W^T is formed with the lane's own W row (16 bytes, held in registers for the whole kernel)
// as the B operand and four scalar LDS reads of H as the A operand, U^T replaces it in place and feeds the second product
// (U H^T)^T[j][i] = sum_c H[j][c] U^T[c][i] as its B operand, whose A operand H[j = i16][16 s + 4 kq + e] is one 16-byte
// LDS read.  blockIdx.y splits the columns; partial slabs are summed by reduce_partials.
__global__ __launch_bounds__(256) void kl_uht16_kernel(Kl16Args p, BatchTab bt) {
    rebase_args(p, bt);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BM = 128, XT = BM * BK, YT = 16 * BK, STAGE = XT + YT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, kq = lane >> 4;
    const long row0 = (long)blockIdx.x * BM;
    const long cbeg = (long)blockIdx.y * p.cols_per_split;
    long cend = cbeg + p.cols_per_split;
    if (cend > p.n) cend = p.n;
    const int nk = (int)((cend - cbeg) / BK);
    const long last = p.m - 1;

    // descriptors at (row0, cbeg) of A and (0, cbeg) of H; lane offsets in bytes (rows past the end re-read the last row)
    const i32x4 rsx = buf_rsrc(p.A + row0 * p.lda + cbeg);
    const i32x4 rsy = buf_rsrc(p.H + cbeg + (p.hblk ? (cbeg / p.hblk) * p.hextra : 0));
    int vx[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        long r = row0 + (tid >> 3) + 32 * it;
        r = r < last ? r : last;
        vx[it] = (int)((r - row0) * p.lda * 4) + (tid & 7) * 16;
    }
    const int ty = tid & 127;                              // threads 128..255 duplicate the H tile's loads and stores
    const int vy = (int)((ty >> 3) * p.ldh * 4) + (ty & 7) * 16;
    const int kshift = (int)((blockIdx.x * 37u) % (unsigned)nk);
    f32x4 x0[4], x1[4], y0, y1;
    auto load = [&](f32x4 (&xr)[4], f32x4& yv, int kt) {
        kt = kt < nk ? kt : nk - 1;
        kt += kshift;
        kt = kt >= nk ? kt - nk : kt;
        const int c0 = kt * (BK * 4);                      // wave uniform: the scalar offset of the loads
#pragma unroll
        for (int it = 0; it < 4; ++it) xr[it] = buf_ld_f32x4(rsx, vx[it], c0, 2);
        yv = buf_ld_f32x4(rsy, vy, c0, 0);
    };
    auto store = [&](float* st, const f32x4 (&xr)[4], const f32x4& yv) {
#pragma unroll
        for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(&st[lds_idx((tid >> 3) + 32 * it, tid & 7)]) = xr[it];
        *reinterpret_cast<f32x4*>(&st[XT + lds_idx(ty >> 3, ty & 7)]) = yv;
    };
    f32x4 wreg[2], out[2];                                 // wreg[rs][t] = W[row][4 kq + t]; out[rs][r] = UHT[row][4 kq + r]
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
        long r = row0 + wave * 32 + rs * 16 + i16;
        r = r < last ? r : last;
        wreg[rs] = *reinterpret_cast<const f32x4*>(p.W + r * p.ldw + 4 * kq);
        out[rs] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 epsv = {p.eps, p.eps, p.eps, p.eps};
    // LDS offsets of the fragment reads are loop invariant (floats): A rows, the H row of the second product, the four H
    // values of the first
    int oa[2][2], ohb[2], ohs[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int rs = 0; rs < 2; ++rs) oa[s][rs] = lds_idx(wave * 32 + rs * 16 + i16, 4 * s + kq);
        ohb[s] = XT + lds_idx(i16, 4 * s + kq);
#pragma unroll
        for (int t = 0; t < 4; ++t) ohs[s][t] = XT + lds_idx(4 * kq + t, 4 * s + (i16 >> 2)) + (i16 & 3);
    }
    auto group = [&](const float* st, int s) {
        f32x4 a[2], sT[2];
#pragma unroll
        for (int rs = 0; rs < 2; ++rs) a[rs] = *reinterpret_cast<const f32x4*>(&st[oa[s][rs]]);
        const f32x4 hb = *reinterpret_cast<const f32x4*>(&st[ohb[s]]);                    // H[i16][16 s + 4 kq + e]
        float hs[4];                                                                      // H[4 kq + t][16 s + i16]
#pragma unroll
        for (int t = 0; t < 4; ++t) hs[t] = st[ohs[s][t]];
#pragma unroll
        for (int rs = 0; rs < 2; ++rs) sT[rs] = MFMA16(hs[0], wreg[rs][0], epsv);         // S^T + eps
#pragma unroll
        for (int t = 1; t < 4; ++t)
#pragma unroll
            for (int rs = 0; rs < 2; ++rs) sT[rs] = MFMA16(hs[t], wreg[rs][t], sT[rs]);
#pragma unroll
        for (int rs = 0; rs < 2; ++rs)
#pragma unroll
            for (int e = 0; e < 4; ++e) sT[rs][e] = div16(a[rs][e], sT[rs][e]);           // U^T (dist_nmf.py:806)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int rs = 0; rs < 2; ++rs) out[rs] = MFMA16(hb[e], sT[rs][e], out[rs]);
    };
    float* st0 = smem;
    float* st1 = smem + STAGE;
    load(x0, y0, 0);
    store(st0, x0, y0);
    __syncthreads();
    load(x1, y1, 1);
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        load(x0, y0, kt + 2);
        group(st0, 0);
        __builtin_amdgcn_sched_barrier(0);
        store(st1, x1, y1);
        __builtin_amdgcn_sched_barrier(0);
        group(st0, 1);
        __syncthreads();
        load(x1, y1, kt + 3);
        group(st1, 0);
        __builtin_amdgcn_sched_barrier(0);
        store(st0, x0, y0);
        __builtin_amdgcn_sched_barrier(0);
        group(st1, 1);
        __syncthreads();
    }
    if (kt < nk) {
        group(st0, 0);
        group(st0, 1);
    }
    float* Pc = p.P + (long)blockIdx.y * p.chunk_stride;
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
        const long row = row0 + wave * 32 + rs * 16 + i16;
        if (row < p.m) *reinterpret_cast<f32x4*>(Pc + row * p.ldp + 4 * kq) = out[rs];
    }
}

constexpr size_t kl_uht16_lds_bytes() { return 2ul * (128 * BK + 16 * BK) * sizeof(float); }

}  // namespace
