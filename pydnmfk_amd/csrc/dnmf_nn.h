// dnmf_nn.h -- NN-small-k form: S = W H tiles in accumulators (residual norm, the two KL products).
// Part of libdnmf_hip.so (kernels live in anonymous namespaces of the headers; the translation units csrc/*.hip include what they launch).
#pragma once
#include "dnmf_common.h"
#include "dnmf_nt.h"

namespace {

// =============================================================================================== NN-small-k form
// S[i][c] = sum_j W[i][j] H[j][c] computed tile-wise in accumulators, never stored:
//   acc[mt][ne] (reg, lane) = S[i = row0 + mt*32 + crow(reg,h)] ... wait: here the MFMA M index is the A-row i,
//   so C/D rows are i and C/D columns (lanes) are the data columns c = col0 + 4*li + ne.
// Used for the residual norm (pyDNMF.py:205-218) and the KL H-side product W^T U (dist_nmf.py:806-808).
enum { NN_RESID = 0, NN_KL_WTU = 1 };

struct NnArgs {
    const float* A; long lda; long m; long n;
    const float* W; long ldw; const float* H; long ldh; int k;
    float eps; double* out;                          // NN_RESID
    float* P; long chunk_stride; long ldp;           // NN_KL_WTU partials [rowblk][KP][ldp]
    long nrowblk; int ncolblk;
    int pipe;                                        // NN_KL_*: software-pipelined interior path (DNMF_KL_PIPE=0 switches it off)
    long hblk; long hextra;                          // kl_uht: H given as column blocks [n / hblk][k][hblk] (an allgather's receive
                                                     // buffer): columns [q hblk, (q+1) hblk) live at H + q hextra with ldh = hblk;
                                                     // hblk = 0: one k x n matrix.  A column split never straddles a block.
    int kreal;                                       // NN_KL_*: the rank before zero padding to KP: S = W H skips the MFMA steps whose 8
                                                     // contraction indices are all padding (k <= 16: half of that product's matrix work)
    long rowtile0;                                   // kl_uht: first 128-row tile of this launch (the full tiles below it went to
                                                     // kl_uht_pipe_kernel, csrc/dnmf_kluht.h)
};
__device__ __forceinline__ void rebase_args(NnArgs& p, const BatchTab& bt) {
    rebase(p.A, bt); rebase(p.W, bt); rebase(p.H, bt); rebase(p.out, bt); rebase(p.P, bt);
}

// S tile for rows [row0, row0 + 32) x cols [col0, col0 + 32*NT): acc[ne] over contraction j in [0, KP)
template <int KT, int NT, bool FAST, bool INTERIOR>
__device__ __forceinline__ void nn_tile(f32x16 (&acc)[NT], const float* __restrict__ W, long ldw, long m, int k,
                                        const float* __restrict__ H, long ldh, long n, long row0, long col0, int li,
                                        int h) {
#pragma unroll
    for (int ne = 0; ne < NT; ++ne)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ne][r] = 0.f;
    const long wrow = row0 + li;
#pragma unroll
    for (int s = 0; s < 4 * KT; ++s) {  // 8 contraction indices per step: jj = 8s + 4h + e
        float a[4];
        load_tile_vec<4, FAST, INTERIOR>(a, W + wrow * ldw, 8 * s + 4 * h, k, wrow < m);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int jj = 8 * s + 4 * h + e;
            float b[NT];
            load_tile_vec<NT, FAST, INTERIOR>(b, H + (long)jj * ldh, col0 + NT * li, n, jj < k);
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) acc[ne] = MFMA32(a[e], b[ne], acc[ne]);
        }
    }
}

// INTERIOR: the W / H loads need no predication (k == KP and the tile is in bounds); AI: the A tile is in bounds
template <int KT, bool FAST, bool INTERIOR, bool AI, typename TA>
__device__ __forceinline__ float resid_tile(const NnArgs& p, long row0, long col0, int li, int h) {
    f32x16 acc[4];
    // bf16 A on the vector path: request the whole 32 x 128 tile (raw, 2 registers per row) BEFORE the W H product so
    // its latency hides under the MFMAs; widened at the point of use.
    constexpr bool PRE = FAST && AI && std::is_same<TA, bf16_t>::value;
    Raw<TA, 4> araw[PRE ? 16 : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            araw[r].load_nt(reinterpret_cast<const TA*>(p.A) + (row0 + crow(r, h)) * p.lda + col0 + 4 * li);
    }
    nn_tile<KT, 4, FAST, INTERIOR>(acc, p.W, p.ldw, p.m, p.k, p.H, p.ldh, p.n, row0, col0, li, h);
    float part = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long row = row0 + crow(r, h);
        float a[4];
        if constexpr (PRE) araw[r].get(a);
        else load_tile_vec<4, FAST, AI>(a, reinterpret_cast<const TA*>(p.A) + row * p.lda, col0 + 4 * li, p.n, row < p.m);
#pragma unroll
        for (int ne = 0; ne < 4; ++ne) {
            // rows >= m and cols >= n have a = 0 and acc = 0 (zero-filled operands) -> contribute 0
            const float d = a[ne] - acc[ne][r];
            part += d * d;
        }
    }
    return part;
}

// TA = storage type of A (float, or bf16_t: p.A then carries the bf16 pointer reinterpreted)
template <int KT, bool FAST, typename TA = float>
__global__ __launch_bounds__(256) void resid_kernel(NnArgs p, BatchTab bt) {
    rebase_args(p, bt);
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long gw = (long)blockIdx.x * 4 + wid;
    double total = 0.0;
    if (gw < p.nrowblk * p.ncolblk) {
        const long rowblk = gw / p.ncolblk, colblk = gw % p.ncolblk;
        const long row0 = rowblk * 32, col0 = colblk * 128;
        const bool inb = FAST && row0 + 32 <= p.m && col0 + 128 <= p.n;
        const bool interior = inb && p.k == 32 * KT;
        // k < KP: W / H loads stay predicated.  Un-predicating only the A loads pays for bf16 (they are hoisted above
        // the product, 3.8 -> 2.0 ms at 262144 x 8192, k = 16) but is slower for fp32 (2.3 -> 3.7 ms), so fp32 keeps them predicated.
        const bool ai = inb && std::is_same<TA, bf16_t>::value;
        total = (double)(interior ? resid_tile<KT, FAST, true, true, TA>(p, row0, col0, li, h)
                         : ai     ? resid_tile<KT, FAST, false, true, TA>(p, row0, col0, li, h)
                                  : resid_tile<KT, FAST, false, false, TA>(p, row0, col0, li, h));
    }
    block_atomic_sum(total, p.out);
}

// The same residual with the H operand staged through LDS (round 3).  resid_kernel above reads its B operand -- four floats
// of a row of H per MFMA group -- from global memory: 64 vector loads per 32 x 128 tile that all hit the cache but keep the
// texture path busy and leave the single wave of a tile waiting (58 % of the fp32 MFMA peak at k = 64).  Here a workgroup's
// four waves share one block of 128 columns, whose KP x 128 block of H is loop invariant and staged ONCE (lane-contiguous
// rows: conflict-free ds_read_b128 as the B operand), and each wave walks a chunk of 32-row blocks -- the structure of
// kl_wtu_kernel.  Interior chunks are software pipelined: the 16 A rows of a block are requested at the top of its product
// (their latency hides under 64 KT MFMAs), the W fragments of block b+1 during the product of block b (one load per MFMA
// group, second register set), the H values of step s+1 are read from LDS during step s.  Edge chunks (ragged rows or
// columns, k < KP) take resid_tile.
template <int KT, typename TA>
__device__ __forceinline__ double resid_chunk_pipe(const NnArgs& p, const float* smem, long rb0, long rb1, long col0, int li, int h) {
    constexpr int NT = 4, CW = 128, NS = 4 * KT;
    const TA* Ab = reinterpret_cast<const TA*>(p.A) + col0 + NT * li;     // + row * lda
    const float* Wf = p.W + 4 * h;                                        // + (row0 + li) * ldw + 8 s
    float w0[NS][4], w1[NS][4];
    double total = 0.0;
    auto block = [&](long rb, float (&wc)[NS][4], float (&wn)[NS][4], long nxt) {
        const long row0 = rb * 32;
        Raw<TA, NT> araw[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) araw[r].load_nt(Ab + (row0 + crow(r, h)) * p.lda);
        f32x16 acc[NT];
#pragma unroll
        for (int ne = 0; ne < NT; ++ne)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ne][r] = 0.f;
        float hb[2][4][NT];
#pragma unroll
        for (int e = 0; e < 4; ++e) load_vec_raw<NT>(hb[0][e], &smem[(4 * h + e) * CW + NT * li]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            load_vec_raw<4>(wn[s], Wf + (nxt * 32 + li) * p.ldw + 8 * s);        // next block's fragment of this step
            if (s + 1 < NS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) load_vec_raw<NT>(hb[(s + 1) & 1][e], &smem[(8 * (s + 1) + 4 * h + e) * CW + NT * li]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ne = 0; ne < NT; ++ne) acc[ne] = MFMA32(wc[s][e], hb[s & 1][e][ne], acc[ne]);
            __builtin_amdgcn_sched_barrier(0);
        }
        float part = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a[NT];
            araw[r].get(a);
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) {
                const float d = a[ne] - acc[ne][r];
                part = fmaf(d, d, part);
            }
        }
        total += (double)part;
    };
#pragma unroll
    for (int s = 0; s < NS; ++s) load_vec_raw<4>(w0[s], Wf + (rb0 * 32 + li) * p.ldw + 8 * s);
    long rb = rb0;
    for (; rb + 2 <= rb1; rb += 2) {
        block(rb, w0, w1, rb + 1);
        block(rb + 1, w1, w0, rb + 2 < rb1 ? rb + 2 : rb + 1);             // past the end: re-read (unused)
    }
    if (rb < rb1) block(rb, w0, w1, rb);
    return total;
}

template <int KT, bool FAST, typename TA = float>
__global__ __launch_bounds__(256, 1) void resid_lds_kernel(NnArgs p, long rowblks_per_chunk, BatchTab bt) {
    rebase_args(p, bt);   // one wave per SIMD: 224+ live registers (two waves spill)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KP = 32 * KT, CW = 128;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const long nchunks = cdiv(p.nrowblk, rowblks_per_chunk);
    const long colblk = blockIdx.x % p.ncolblk;
    const long chunk = (blockIdx.x / p.ncolblk) * 4 + wid;
    const long col0 = colblk * CW;
    for (int idx = tid; idx < KP * (CW / 4); idx += 256) {          // stage H[0:KP][col0:col0+CW] (zero outside k x n)
        const int jj = idx / (CW / 4), c4 = (idx % (CW / 4)) * 4;
        float d[4];
        load_vec<4, FAST>(d, p.H + (long)jj * p.ldh, col0 + c4, p.n, jj < p.k);
        *reinterpret_cast<f32x4*>(&smem[jj * CW + c4]) = f32x4{d[0], d[1], d[2], d[3]};
    }
    __syncthreads();
    double total = 0.0;
    if (chunk < nchunks) {
        const long rb0 = chunk * rowblks_per_chunk;
        long rb1 = rb0 + rowblks_per_chunk;
        if (rb1 > p.nrowblk) rb1 = p.nrowblk;
        if (FAST && p.k == KP && col0 + CW <= p.n && rb1 * 32 <= p.m) {
            total = resid_chunk_pipe<KT, TA>(p, smem, rb0, rb1, col0, li, h);
        } else {
            for (long rb = rb0; rb < rb1; ++rb) total += (double)resid_tile<KT, FAST, false, false, TA>(p, rb * 32, col0, li, h);
        }
    }
    block_atomic_sum(total, p.out);
}

// Per-column residual statistics (PyNMF.column_err, pyDNMF.py:221-239): num[c] += sum_i (A[i][c] - (W H)[i][c])^2 and
// den[c] += sum_i A[i][c]^2 over this rank's rows -- the residual kernel's tile loop with one accumulator pair per
// column instead of one scalar.  A wave owns a 128-column block and walks a chunk of 32-row blocks; fp32 partial sums per
// tile (16 rows), fp64 across tiles; the two lane halves are combined with one shuffle and 32 lanes add 4 columns each to
// the global fp64 arrays (atomics: these are statistics, like the norms of pyDNMF.py:205-218).
template <int KT, bool FAST, typename TA = float>
__global__ __launch_bounds__(256) void colerr_kernel(NnArgs p, double* __restrict__ num, double* __restrict__ den,
                                                     long rowblks_per_chunk, BatchTab bt) {
    rebase_args(p, bt); REBASE(num); REBASE(den);
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long gw = (long)blockIdx.x * 4 + wid;
    const long nchunks = cdiv(p.nrowblk, rowblks_per_chunk);
    if (gw >= nchunks * p.ncolblk) return;
    const long chunk = gw / p.ncolblk, colblk = gw % p.ncolblk;
    const long col0 = colblk * 128;
    double dn[4] = {0.0, 0.0, 0.0, 0.0}, dd[4] = {0.0, 0.0, 0.0, 0.0};
    long rb1 = (chunk + 1) * rowblks_per_chunk;
    if (rb1 > p.nrowblk) rb1 = p.nrowblk;
    for (long rb = chunk * rowblks_per_chunk; rb < rb1; ++rb) {
        const long row0 = rb * 32;
        f32x16 acc[4];     // (one predicated tile variant: the kernel runs once per k of an NMFk sweep, not per iteration)
        nn_tile<KT, 4, FAST, false>(acc, p.W, p.ldw, p.m, p.k, p.H, p.ldh, p.n, row0, col0, li, h);
        float pn[4] = {0.f, 0.f, 0.f, 0.f}, pd[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long row = row0 + crow(r, h);
            float a[4];
            load_tile_vec<4, FAST, false>(a, reinterpret_cast<const TA*>(p.A) + row * p.lda, col0 + 4 * li, p.n, row < p.m);
#pragma unroll
            for (int ne = 0; ne < 4; ++ne) {
                const float d = a[ne] - acc[ne][r];        // rows >= m / cols >= n: a = 0 and acc = 0
                pn[ne] = fmaf(d, d, pn[ne]);
                pd[ne] = fmaf(a[ne], a[ne], pd[ne]);
            }
        }
#pragma unroll
        for (int ne = 0; ne < 4; ++ne) { dn[ne] += (double)pn[ne]; dd[ne] += (double)pd[ne]; }
    }
#pragma unroll
    for (int ne = 0; ne < 4; ++ne) {
        dn[ne] += __shfl_xor(dn[ne], 32, 64);
        dd[ne] += __shfl_xor(dd[ne], 32, 64);
        const long c = col0 + 4 * li + ne;
        if (h == 0 && c < p.n) { atomicAdd(num + c, dn[ne]); atomicAdd(den + c, dd[ne]); }
    }
}

// U = A / (S + eps) of the KL products (dist_nmf.py:806), round 3: eps is the INITIAL value of the S accumulators and the
// quotient is a * v_rcp_f32(d) -- 2 vector instructions per element where hipcc's IEEE division sequence + the add are 11.
// On gfx950 every fp32 vector instruction takes ~3.3 cycles of matrix-pipe time (tools/coissue.hip), and the division sits
// between the two products of every element: 8 % of a KL pass at k = 128, 17 % at k = 64, 30 % at k = 32 before.  At most 1.5 ulp
// from the exact quotient; U is summed over thousands of rows / columns right after (same form as csrc/dnmf_kl16.h).
__device__ __forceinline__ float kl_quot(float a, float d) { return a * __builtin_amdgcn_rcpf(d); }

// KL H-side: P[chunk][j][c] = sum_{i in chunk} W[i][j] * A[i][c] / (S[i][c] + eps)          (dist_nmf.py:806,808)
// A workgroup = 4 waves that share one block of CW = 32*NT columns and each own a chunk of 32-row blocks.  The
// KP x CW block of H those columns need is loop invariant: it is staged ONCE per workgroup into LDS (row jj,
// lane-contiguous columns -> conflict-free ds_read_b64/b128 as the B operand of S = W H).  Per row block a wave forms
// S (NN tile), turns it into U in place (same C/D registers) and feeds U as the B operand of W^T U: the contraction
// index i is the C/D row index, i.e. it lives in registers, which is exactly the B-operand layout (row pairs
// (rho, rho+4)).  The A tile is requested before the S product so its latency hides under it.
// one 32-row block of the KL H-side product (see kl_wtu_kernel); smem = the workgroup's KP x CW block of H
template <int KT, int NT, bool FAST, bool INTERIOR>
__device__ __forceinline__ void kl_wtu_block(f32x16 (&out)[KT][NT], const NnArgs& p, const float* smem, long row0,
                                             long col0, int li, int h) {
    constexpr int CW = 32 * NT;
    float areg[16][NT];   // A[row0 + crow(r,h)][col0 + NT*li + ne], requested first
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long row = row0 + crow(r, h);
        load_tile_vec<NT, FAST, INTERIOR>(areg[r], p.A + row * p.lda, col0 + NT * li, p.n, row < p.m);
    }
    f32x16 acc[NT];
#pragma unroll
    for (int ne = 0; ne < NT; ++ne)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ne][r] = 0.f;          // (eps is added in the quotient below: this block has no scheduling
                                                                // barriers -- HAZARD 2 in dnmf_common.h; the pipelined chunk keeps the eps start)
    const long wrow = row0 + li;
#pragma unroll
    for (int s = 0; s < 4 * KT; ++s) {  // S = W H: contraction jj = 8s + 4h + e
        float a[4];
        load_tile_vec<4, FAST, INTERIOR>(a, p.W + wrow * p.ldw, 8 * s + 4 * h, p.k, wrow < p.m);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int jj = 8 * s + 4 * h + e;
            float b[NT];
            load_vec_raw<NT>(b, &smem[jj * CW + NT * li]);
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) acc[ne] = MFMA32(a[e], b[ne], acc[ne]);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int ne = 0; ne < NT; ++ne) acc[ne][r] = kl_quot(areg[r][ne], acc[ne][r] + p.eps);  // U (dist_nmf.py:806)
    // out[ke][ne] += sum_i W[i][KT*li + ke] * U[i][c]: A-operand lane (li, h) holds W[row0 + crow(r,h)][KT*li + ke]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long row = row0 + crow(r, h);
        float w[KT];
        load_tile_vec<KT, FAST, INTERIOR>(w, p.W + row * p.ldw, (long)KT * li, p.k, row < p.m);
#pragma unroll
        for (int ke = 0; ke < KT; ++ke)
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) out[ke][ne] = MFMA32(w[ke], acc[ne][r], out[ke][ne]);
    }
}


// The same row blocks, software pipelined across blocks for a chunk whose tiles are all in bounds (k == KP): every load
// is unconditional and is issued one phase before its data is needed, ONE load per MFMA group, with the issue order
// pinned --
//   phase 1 (S = W H, 16 steps):   step s issues the load of w3[s] (the W rows the third phase of THIS block needs)
//   phase 2 (U = A / (S + eps))
//   phase 3 (out += W^T U, 16 steps): step r issues the loads of areg[r] and wfrag[r] of the NEXT block
// -- and each register set is refilled in the phase after the one that consumed it, so nothing is double buffered.
// With predicated loads (kl_wtu_block) hipcc drains vmcnt(0) at every one of the 48 loads of a block and the single
// wave per SIMD waits out each latency: MFMA busy 51 % at a 2.39 GHz clock (32768 x 32768, k = 128).
template <int KT, int NT, bool SKIP>
__device__ __forceinline__ void kl_wtu_chunk_pipe(f32x16 (&out)[KT][NT], const NnArgs& p, const float* smem, long rb0,
                                                  long rb1, long col0, int li, int h) {
    constexpr int CW = 32 * NT;
    float areg[16][NT], wfrag[4 * KT][4], w3[16][KT];
    // MUBUF addressing (round 3): descriptors at the chunk's first row; a lane's offset is loop invariant, the row / block
    // part of every address is a scalar offset -- no vector address arithmetic beside the MFMAs.  (The caller checks that
    // one chunk of A and of W fits the 2 GiB window.)
    const i32x4 rsa = buf_rsrc(p.A + rb0 * 32 * p.lda + col0), rsw = buf_rsrc(p.W + rb0 * 32 * p.ldw);
    const int lda4 = (int)(p.lda * 4), ldw4 = (int)(p.ldw * 4);
    const int va = 4 * h * lda4 + NT * li * 4;              // A[.. + crow(r, h)][col0 + NT li]: crow = (r&3) + 8 (r>>2) + 4 h
    const int vwf = li * ldw4 + 16 * h;                     // W[.. + li][8 s + 4 h]          (S product: lane = row li)
    const int vw3 = 4 * h * ldw4 + KT * li * 4;             // W[.. + crow(s, h)][KT li]      (third phase: lane = column group)
    auto urow = [](int r) { return (r & 3) + 8 * (r >> 2); };
    auto issue_next = [&](int rbr, int r) {                 // block rb0 + rbr: A row crow(r,h) and W fragment s = r
        buf_load<NT, 2>(areg[r], rsa, va, (rbr * 32 + urow(r)) * lda4);
        if (r < 4 * KT) buf_load<4, 0>(wfrag[r], rsw, vwf, rbr * 32 * ldw4 + 32 * r);
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) issue_next(0, r);
    static_assert(4 * KT <= 16, "one W fragment per third-phase step");
    const int nrb = (int)(rb1 - rb0);
    for (int rbr = 0; rbr < nrb; ++rbr) {
        const int nxt = rbr + 1 < nrb ? rbr + 1 : rbr;      // past the end: re-read this block (unused)
        f32x16 acc[NT];
#pragma unroll
        for (int ne = 0; ne < NT; ++ne)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ne][r] = p.eps;
        float hb[2][4][NT];                                 // H operand of step s+1 read from LDS during step s
#pragma unroll
        for (int e = 0; e < 4; ++e) load_vec_raw<NT>(hb[0][e], &smem[(4 * h + e) * CW + NT * li]);
#pragma unroll
        for (int s = 0; s < 4 * KT; ++s) {                  // phase 1
            if (s < 16) buf_load<KT, 0>(w3[s], rsw, vw3, (rbr * 32 + urow(s)) * ldw4);
            if (4 * KT < 16 && s == 4 * KT - 1) {
#pragma unroll
                for (int q = 4 * KT; q < 16; ++q) buf_load<KT, 0>(w3[q], rsw, vw3, (rbr * 32 + urow(q)) * ldw4);
            }
            if (s + 1 < 4 * KT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) load_vec_raw<NT>(hb[(s + 1) & 1][e], &smem[(8 * (s + 1) + 4 * h + e) * CW + NT * li]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!SKIP || 8 * s < p.kreal) {                      // wave uniform; the loads above stay unconditional (pipeline).  SKIP: KT = 1
                                                                 // with a rank of at most 24 only (round 4: the test itself -- a scalar
                                                                 // branch per step -- cut the block into basic blocks hipcc schedules one by one)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ne = 0; ne < NT; ++ne) acc[ne] = MFMA32(wfrag[s][e], hb[s & 1][e][ne], acc[ne]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)                        // phase 2
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) acc[ne][r] = kl_quot(areg[r][ne], acc[ne][r]);   // U (dist_nmf.py:806)
#pragma unroll
        for (int r = 0; r < 16; ++r) {                      // phase 3
            issue_next(nxt, r);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ke = 0; ke < KT; ++ke)
#pragma unroll
                for (int ne = 0; ne < NT; ++ne) out[ke][ne] = MFMA32(w3[r][ke], acc[ne][r], out[ke][ne]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int KT, int NT, bool FAST, int OCC = (KT == 2 ? 2 : 1)>
__global__ __launch_bounds__(256, OCC) void kl_wtu_kernel(NnArgs p, long rowblks_per_chunk, BatchTab bt) {
    rebase_args(p, bt);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KP = 32 * KT, CW = 32 * NT;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const long nchunks = cdiv(p.nrowblk, rowblks_per_chunk);
    const long colblk = blockIdx.x % p.ncolblk;
    const long chunk = (blockIdx.x / p.ncolblk) * 4 + wid;
    const long col0 = colblk * CW;
    // stage H[0:KP][col0:col0+CW] (zero outside k x n)
    for (int idx = tid; idx < KP * (CW / 4); idx += 256) {
        const int jj = idx / (CW / 4), c4 = (idx % (CW / 4)) * 4;
        float d[4];
        load_vec<4, FAST>(d, p.H + (long)jj * p.ldh, col0 + c4, p.n, jj < p.k);
        *reinterpret_cast<f32x4*>(&smem[jj * CW + c4]) = f32x4{d[0], d[1], d[2], d[3]};
    }
    __syncthreads();
    if (chunk >= nchunks) return;

    f32x16 out[KT][NT];
#pragma unroll
    for (int ke = 0; ke < KT; ++ke)
#pragma unroll
        for (int ne = 0; ne < NT; ++ne)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[ke][ne][r] = 0.f;
    long rb1 = (chunk + 1) * rowblks_per_chunk;
    if (rb1 > p.nrowblk) rb1 = p.nrowblk;
    // A chunk that is completely in bounds takes the software-pipelined path; edges keep the predicated blocks.
    // (A merely branch-free block, its 16 A loads issued as one VMEM block ahead of the MFMAs, was 8-20 % slower than
    // the predicated one.)
    const long rb0 = chunk * rowblks_per_chunk;
    if (FAST && p.pipe && p.k == KP && col0 + CW <= p.n && rb1 * 32 <= p.m &&
        buf_window_ok((rb1 - rb0 + 1) * 32, p.lda, CW) && buf_window_ok((rb1 - rb0 + 1) * 32, p.ldw, KP)) {   // (one descriptor per chunk)
        if (KT == 1 && p.kreal <= 24) kl_wtu_chunk_pipe<KT, NT, KT == 1>(out, p, smem, rb0, rb1, col0, li, h);
        else kl_wtu_chunk_pipe<KT, NT, false>(out, p, smem, rb0, rb1, col0, li, h);
    } else {
        for (long rb = rb0; rb < rb1; ++rb) kl_wtu_block<KT, NT, FAST, false>(out, p, smem, rb * 32, col0, li, h);
    }
    float* Pc = p.P + chunk * p.chunk_stride;
#pragma unroll
    for (int ke = 0; ke < KT; ++ke)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = KT * crow(r, h) + ke;
            float d[NT];
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) d[ne] = out[ke][ne][r];
            store_vec<NT, true>(d, Pc + (long)j * p.ldp, col0 + (long)NT * li, p.ldp, true);
        }
}

// KL W-side: UHT[i][j] = sum_c (A[i][c] / (S[i][c] + eps)) * H[j][c]                     (dist_nmf.py:806,810)
// The contraction index c of the second product must end up in registers, so S is formed TRANSPOSED:
//   S^T[c][i] = sum_jj H[jj][c] W[i][jj]   MFMA M index = c (A-operand lane (c,h) = H[jj][c]), N index = i (B-operand
//   lane (i,h) = W[i][jj], the lane's own W row, held in registers for the whole kernel).
// C/D then has lane = row i of A and registers = columns c; A is read in that layout (four 16-B pieces per lane and
// 32-column tile, prefetched one tile ahead), U^T replaces S^T in place and is the B operand of
//   (U H^T)^T[j][i] = sum_c H[j][c] U^T[c][i]   (A-operand lane (j,h) = H[j][c]).
// Workgroup = 4 waves x 32 rows; the k x 32 tile of H is staged once per workgroup into LDS (same swizzled image as the
// NT tiles: ds_read_b32 along a row for the first product, ds_read_b128 across rows for the second) and double
// buffered, one barrier per tile.  blockIdx.y splits the columns; partial UHT slabs are summed by reduce_partials.
template <int KT, bool FAST, bool INTERIOR>
__device__ __forceinline__ void kl_uht_body(const NnArgs& p, float* __restrict__ out_base, long ldo, long split_stride,
                                            long cols_per_split, int out_cols, float* smem) {
    constexpr int KP = 32 * KT, T = 256, STAGE = KP * BK, NY = KP / (T / 8);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    const long rtile = (long)blockIdx.x + p.rowtile0;
    const long arow = rtile * 128 + wave * 32 + li;
    const bool rok = arow < p.m;
    const long cbeg = (long)blockIdx.y * cols_per_split;
    long cend = cbeg + cols_per_split;
    if (cend > p.n) cend = p.n;
    const long nt = (cend - cbeg + BK - 1) / BK;
    const float* Hb = p.H + (p.hblk ? (cbeg / p.hblk) * p.hextra : 0);    // this split's column block of H (block uniform)
    // interior workgroups (round 3): A pieces and H tiles through MUBUF descriptors at (first row of the workgroup, cbeg) --
    // a loop-invariant lane offset + the tile's scalar column offset, no vector address arithmetic beside the MFMAs
    const bool buf = INTERIOR && buf_window_ok(128, p.lda, cend - cbeg) && buf_window_ok(KP, p.ldh, cend - cbeg);
    i32x4 rsa = {0, 0, 0, 0}, rsh = {0, 0, 0, 0};
    int va = 0, vh[NY];
    if constexpr (INTERIOR) {
        rsa = buf_rsrc(p.A + rtile * 128 * p.lda + cbeg);
        rsh = buf_rsrc(Hb + cbeg);
        va = (int)((wave * 32 + li) * p.lda * 4) + 16 * h;
        stage_offsets<KP, T>(vh, p.ldh, tid);
    }

    f32x16 out[KT];  // (U H^T)^T tile: rows j (KT tiles of 32), lanes i
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[jt][r] = 0.f;
    float wreg[4 * KT][4];   // W[arow][8s + 4h + e]
#pragma unroll
    for (int s = 0; s < 4 * KT; ++s) load_tile_vec<4, FAST, INTERIOR>(wreg[s], p.W + arow * p.ldw, 8 * s + 4 * h, p.k, rok);

    f32x4 hst[NY];
    float a_cur[4][4];
    const bool hrows_in = p.k >= KP;
    if (nt > 0) {
        if (buf && cbeg + BK <= cend) stage_load_buf<NY, false>(hst, rsh, vh, 0);
        else if (hrows_in && cbeg + BK <= cend) stage_load<KP, T, FAST, true>(hst, Hb, p.ldh, p.k, cend, 0, cbeg, tid);
        else stage_load<KP, T, FAST, false>(hst, Hb, p.ldh, p.k, cend, 0, cbeg, tid);
        stage_store<KP, T>(smem, hst, tid);
        if (buf && cbeg + BK <= cend) {
#pragma unroll
            for (int g = 0; g < 4; ++g) buf_load<4, 0>(a_cur[g], rsa, va + 32 * g, 0);
        } else if (INTERIOR && cbeg + BK <= cend) {
#pragma unroll
            for (int g = 0; g < 4; ++g) load_tile_vec<4, FAST, INTERIOR>(a_cur[g], p.A + arow * p.lda, cbeg + 8 * g + 4 * h, cend, rok);
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) load_vec<4, FAST>(a_cur[g], p.A + arow * p.lda, cbeg + 8 * g + 4 * h, cend, rok);
        }
    }
    __syncthreads();
    // (A branch-free variant of this loop -- next tile requested unconditionally at the top of the tile or spread over
    // the second product, last tile peeled -- measured 5-20 % slower than this one: 5.6-6.6 vs 5.3 ms at 32768^2, k = 128.)
    for (long t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < nt;
        const long c1 = cbeg + (t + 1) * BK;
        const float* Hs = smem + cur * STAGE;
        f32x16 st;  // S^T tile: rows c, lanes i
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = p.eps;
        {   // The four H values of step s+1 are read while the four (dependent) MFMAs of step s run: hipcc otherwise reads
            // ONE value ahead and every MFMA of this product waits out an LDS latency (single wave per SIMD at k = 128).
            float hv[2][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[0][e] = Hs[lds_idx(4 * h + e, li >> 2) + (li & 3)];
#pragma unroll
            for (int s = 0; s < 4 * KT; ++s) {
                if (s + 1 < 4 * KT) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[(s + 1) & 1][e] = Hs[lds_idx(8 * (s + 1) + 4 * h + e, li >> 2) + (li & 3)];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (KT > 1 || 8 * s < p.kreal) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) st = MFMA32(hv[s & 1][e], wreg[s][e], st);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) st[4 * g + e] = kl_quot(a_cur[g][e], st[4 * g + e]);   // U^T (dist_nmf.py:806)
        // the A registers are free now: fetch the next tile's pieces (and the next H tile) under the second product
        if (more) {
            const int so = (int)((c1 - cbeg) * 4);             // block uniform
            if (buf && c1 + BK <= cend) stage_load_buf<NY, false>(hst, rsh, vh, so);
            else if (hrows_in && c1 + BK <= cend) stage_load<KP, T, FAST, true>(hst, Hb, p.ldh, p.k, cend, 0, c1, tid);
            else stage_load<KP, T, FAST, false>(hst, Hb, p.ldh, p.k, cend, 0, c1, tid);
            if (buf && c1 + BK <= cend) {
#pragma unroll
                for (int g = 0; g < 4; ++g) buf_load<4, 0>(a_cur[g], rsa, va + 32 * g, so);
            } else if (INTERIOR && c1 + BK <= cend) {
#pragma unroll
                for (int g = 0; g < 4; ++g) load_tile_vec<4, FAST, INTERIOR>(a_cur[g], p.A + arow * p.lda, c1 + 8 * g + 4 * h, cend, rok);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) load_vec<4, FAST>(a_cur[g], p.A + arow * p.lda, c1 + 8 * g + 4 * h, cend, rok);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int jt = 0; jt < KT; ++jt) {
                const f32x4 hh = *reinterpret_cast<const f32x4*>(&Hs[lds_idx(jt * 32 + li, 2 * g + h)]);
#pragma unroll
                for (int e = 0; e < 4; ++e) out[jt] = MFMA32(hh[e], st[4 * g + e], out[jt]);
            }
        if (more) stage_store<KP, T>(smem + (cur ^ 1) * STAGE, hst, tid);
        __syncthreads();
    }
    // out[jt] (reg, lane): j = jt*32 + crow(reg, h), i = arow; registers 4g..4g+3 are 4 consecutive j
    float* dst = out_base + (long)blockIdx.y * split_stride + arow * ldo;
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float d[4] = {out[jt][4 * g], out[jt][4 * g + 1], out[jt][4 * g + 2], out[jt][4 * g + 3]};
            store_tile_vec<4, FAST, INTERIOR>(d, dst, jt * 32 + 8 * g + 4 * h, out_cols, rok);
        }
}

template <int KT, bool FAST>
__global__ __launch_bounds__(256, KT == 4 ? 1 : 2) void kl_uht_kernel(NnArgs p, float* __restrict__ out_base, long ldo,
                                                        long split_stride, long cols_per_split, int out_cols, BatchTab bt) {
    rebase_args(p, bt); REBASE(out_base);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // block-uniform: all 128 rows in bounds and no rank padding -> branch-free W / A / output accesses
    const bool interior = FAST && p.k == 32 * KT && ((long)blockIdx.x + p.rowtile0 + 1) * 128 <= p.m && out_cols >= 32 * KT;
    if (interior) kl_uht_body<KT, FAST, true>(p, out_base, ldo, split_stride, cols_per_split, out_cols, smem);
    else kl_uht_body<KT, FAST, false>(p, out_base, ldo, split_stride, cols_per_split, out_cols, smem);
}


}  // namespace
