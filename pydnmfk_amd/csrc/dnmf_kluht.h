// dnmf_kluht.h -- the KL W-side product U H^T for 16 < k <= 128 as a software-pipelined kernel (round 4).
// Part of libdnmf_hip.so (kernels live in anonymous namespaces of the headers; the translation units csrc/*.hip include what they launch).
//
//   UHT[i][j] = sum_c (A[i][c] / ((W H)[i][c] + eps)) * H[j][c]                      (dist_nmf.py:806,810; 2D :321-343)
//
// Same operand maps, same arithmetic and the same summation order as kl_uht_body (csrc/dnmf_nn.h) -- the results are bit
// identical -- but only for workgroups whose 128 rows and whose 32-column tiles are all in bounds, with k = KP, so that the
// loop has no branch and every load is unconditional.  What the round-3 kernel lost (profiles/r03c_kl_*: 73.5 % MFMA busy at
// an unthrottled 2.35 GHz, i.e. schedule, not power), read off its ISA:
//   * the second product read its H fragment (ds_read_b128) and waited for it (lgkmcnt(0)) in front of every group of four
//     MFMAs -- an LDS latency per 256 matrix cycles, 16 KT times per tile;
//   * two vector instructions of address arithmetic per ds_read_b32 of the first product (64 reads per tile at k = 128), and
//     on gfx950 a vector instruction of ANY wave takes matrix-pipe time (tools/coissue.hip);
//   * the S^T accumulator shared its registers with out[0], copied away and back through v_accvgpr_mov around every tile, and
//     the A pieces of the next tile were rotated through a second register set with 16 moves.
// Here: every LDS address is a loop-invariant lane offset + an immediate (the two LDS stages are two copies of the tile
// code), H fragments of BOTH products are read one MFMA group ahead, the next tile's H block is requested at the top of a
// tile and written to the other stage before the second-to-last MFMA group, the barrier sits between the last two groups
// (its wait overlaps the MFMAs already issued), and the first H values of the next tile are read right behind it.
#pragma once
#include "dnmf_common.h"
#include "dnmf_nt.h"

namespace {

struct KlUhtArgs {
    const float* A; long lda; long n;
    const float* W; long ldw;
    const float* H; long ldh;
    float eps;
    long hblk; long hextra;              // H as column blocks (see NnArgs in dnmf_nn.h); 0 = one k x n matrix
    float* out; long ldo; long split_stride; long cols_per_split;
};
__device__ __forceinline__ void rebase_args(KlUhtArgs& p, const BatchTab& bt) {
    rebase(p.A, bt); rebase(p.W, bt); rebase(p.H, bt); rebase(p.out, bt);
}

__device__ __forceinline__ float klu_quot(float a, float d) { return a * __builtin_amdgcn_rcpf(d); }

// waves per SIMD the register budget is cut for (KT = 4: out 64 + W row 64 + S^T 16 + A 16 + H stage 16 + fragments 16)
template <int KT> struct KlUhtOcc { static constexpr int value = KT == 4 ? 2 : (KT == 2 ? 3 : 4); };

// AUXA: cache policy of the A loads.  A lane reads 16 bytes of ITS row, so one instruction touches 32 rows x 32 bytes and a
// row's 128-byte line is completed by four instructions: with the nontemporal hint every one of them went to L2 as its own
// 32-byte request (TCP_TCC_READ_REQ: 7.2e7 per launch = 30 B per request at 32768 x 16384, k = 32); plain loads merge in L1
// (0.69 -> 0.65 ms there, +2.5 % at k = 64, nothing at k = 128).
// A2: the A pieces of tile t+1 land in a second register set requested at the top of tile t (a whole tile of latency);
// otherwise they refill the one set right after the quotient consumed it (second product + next first product of latency).
// ABL (tuning build only, tools/kluht_ab.py): ablations that give wrong results but tell where the time goes -- 1: no
// barrier, 2: no quotient, 4: A pieces loaded once, 8: H tile loaded / staged once, 16: no second product, 32: no first product, 64: line-coalesced A requests, 128: one LDS read per product and tile
// MR (round 5): 32-row groups per wave.  MR = 2: a wave owns 64 rows -- two S^T tiles, two out strips, two sets of A
// pieces -- and every H fragment read from LDS (both products) feeds TWO MFMAs: half the LDS reads per matrix instruction, the
// one term of the k <= 64 kernels' instruction mix that a wider tile can shrink.  Same arithmetic per element in the same order:
// bit identical to MR = 1.  The workgroup then covers 256 rows.
template <int KT, bool A2, int OCC = KlUhtOcc<KT>::value, int ABL = 0, int AUXA = 0, int MR = 1>
__global__ __launch_bounds__(256, OCC) void kl_uht_pipe_kernel(KlUhtArgs p, BatchTab bt) {
    static_assert(MR == 1 || ABL == 0, "the ablations exist for the one-group kernel");
    rebase_args(p, bt);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KP = 32 * KT, STAGE = KP * BK, NY = KP / 32, NG = 4 * KT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    const long arow0 = (long)blockIdx.x * 128 * MR;
    const long arow = arow0 + wave * 32 * MR + li;                 // the lane's row of group 0; group r: + 32 r
    const long cbeg = (long)blockIdx.y * p.cols_per_split;
    long cend = cbeg + p.cols_per_split;
    if (cend > p.n) cend = p.n;
    const int nt = (int)((cend - cbeg) / BK);
    const float* Hb = p.H + (p.hblk ? (cbeg / p.hblk) * p.hextra : 0);    // this split's column block of H (block uniform)
    const i32x4 rsa = buf_rsrc(p.A + arow0 * p.lda + cbeg), rsh = buf_rsrc(Hb + cbeg);
    const int va = (ABL & 64) ? (int)((wave * 32 + (lane >> 3)) * p.lda * 4) + 16 * (lane & 7)    // (ablation: line-coalesced requests, wrong data)
                              : (int)((wave * 32 * MR + li) * p.lda * 4) + 16 * h;     // A[arow][c0 + 8 g + 4 h ..+3] at va + 32 g (+ tile offset)
    int vam[MR];                                                    // group r: 32 r rows further down
#pragma unroll
    for (int r = 0; r < MR; ++r) vam[r] = va + (int)(32 * r * p.lda * 4);
    const int lda32 = (int)(p.lda * 32);
    int vh[NY];
    stage_offsets<KP, 256>(vh, p.ldh, tid);

    // LDS image of a tile: row jj = 32 floats, 16-byte chunk c4 stored at c4 ^ ((jj >> 1) & 7)  (lds_idx, dnmf_nt.h)
    // first product, lane (c = li, h) reads H[jj = 8 s + 4 h + e][c]: (jj >> 1) & 7 = 4 (s & 1) + 2 h + (e >> 1)
    int ax[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ax[q] = 4 * h * BK + ((((li >> 2) ^ (4 * (q >> 1) + 2 * h + (q & 1))) << 2) | (li & 3));
    // second product, lane (j = li, h) reads H[jt 32 + li][8 g + 4 h ..+3] = chunk 2 g + h of row jt 32 + li
    int ay[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) ay[g] = li * BK + (((2 * g + h) ^ ((li >> 1) & 7)) << 2);
    // staging: thread t owns chunk t & 7 of rows (t >> 3) + 32 it
    const int aw = (tid >> 3) * BK + (((tid & 7) ^ ((tid >> 4) & 7)) << 2);

    f32x16 out[MR][KT];  // (U H^T)^T tile: rows j (KT tiles of 32), lanes i
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[m][jt][r] = 0.f;
    float wreg[MR][NG][4];   // W[arow + 32 m][8 s + 4 h + e]: the lane's own rows, B operand of the first product for the whole kernel
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int s = 0; s < NG; ++s) load_vec_raw<4>(wreg[m][s], p.W + (arow + 32 * m) * p.ldw + 8 * s + 4 * h);

    f32x4 hst[NY];
    float a0[MR][4][4], a1[A2 ? MR : 1][A2 ? 4 : 1][4];
    float hv[2][4];
    f32x4 hh[2];

    auto load_a = [&](auto& a, int so) {
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if constexpr ((ABL & 64) != 0) buf_load<4, AUXA>(a[m][g], rsa, va, so + g * lda32);
                else buf_load<4, AUXA>(a[m][g], rsa, vam[m] + 32 * g, so);
    };
    auto store_h = [&](float* stage) {
#pragma unroll
        for (int it = 0; it < NY; ++it) *reinterpret_cast<f32x4*>(&stage[aw + it * 32 * BK]) = hst[it];
    };
    auto read_hv = [&](float (&v)[4], const float* Hs, auto S) {
        constexpr int s = decltype(S)::value;
        if constexpr ((ABL & 128) != 0) { if (s != 0) return; }        // (ablation: the first step's values for every step)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = Hs[ax[2 * (s & 1) + (e >> 1)] + (8 * s + e) * BK];
    };
    auto read_hh = [&](f32x4& v, const float* Hs, auto Q) {
        constexpr int q = decltype(Q)::value, g = q / KT, jt = q % KT;
        if constexpr ((ABL & 128) != 0) { if (q != 0) return; }
        v = *reinterpret_cast<const f32x4*>(&Hs[ay[g] + jt * 32 * BK]);
    };

    // one 32-column tile: H tile in stage CUR (visible), hv[0] = its first-step values (already read), ac = its A pieces;
    // tn = the tile to request (clamped to the last one: the loop is branch free, a request past the end re-reads it unused)
    auto tile = [&](auto CUR, auto& ac, auto& an, int tn) {
        constexpr int cur = decltype(CUR)::value;
        const float* Hs = smem + cur * STAGE;
        float* Hn = smem + (cur ^ 1) * STAGE;
        const int so = tn * (BK * 4);                       // wave uniform
        if constexpr (!(ABL & 8)) stage_load_buf<NY, false>(hst, rsh, vh, so);
        if constexpr (A2 && !(ABL & 4)) load_a(an, so);
        f32x16 st[MR];   // S^T tiles (rows c, lanes i), eps = initial value: S + eps costs nothing
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[m][r] = p.eps;
        static_for<0, NG>([&](auto S) {
            constexpr int s = decltype(S)::value;
            if constexpr (s + 1 < NG) read_hv(hv[(s + 1) & 1], Hs, std::integral_constant<int, s + 1>{});
            else read_hh(hh[0], Hs, std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MR; ++m)                    // (group by group: four MFMAs on one chain behind its eps block, HAZARD 2)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if constexpr (!(ABL & 32)) st[m] = MFMA32(hv[s & 1][e], wreg[m][s][e], st[m]);
                    else st[m][e] += hv[s & 1][e] * wreg[m][s][e];
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; e += 2)
                    if constexpr (!(ABL & 2)) {                                   // U^T (dist_nmf.py:806): two reciprocals, ONE packed multiply
                        const f32x2 r = {__builtin_amdgcn_rcpf(st[m][4 * g + e]), __builtin_amdgcn_rcpf(st[m][4 * g + e + 1])};
                        const f32x2 q = f32x2{ac[m][g][e], ac[m][g][e + 1]} * r;
                        st[m][4 * g + e] = q[0]; st[m][4 * g + e + 1] = q[1];
                    } else if (g == 0 && e == 0) st[m][0] += ac[m][0][0] + ac[m][1][1] + ac[m][2][2] + ac[m][3][3];
        if constexpr (!A2 && !(ABL & 4)) load_a(ac, so);
        static_for<0, NG>([&](auto Q) {
            constexpr int q = decltype(Q)::value, g = q / KT, jt = q % KT;
            if constexpr (q + 1 < NG) read_hh(hh[(q + 1) & 1], Hs, std::integral_constant<int, q + 1>{});
            if constexpr (q == NG - 2 && !(ABL & 8)) store_h(Hn);
            if constexpr (q == NG - 1) {
                if constexpr (!(ABL & 1)) __syncthreads();
                read_hv(hv[0], Hn, std::integral_constant<int, 0>{});
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if constexpr (!(ABL & 16)) out[m][jt] = MFMA32(hh[q & 1][e], st[m][4 * g + e], out[m][jt]);
                    else out[m][jt][e] += hh[q & 1][e] * st[m][4 * g + e];
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    stage_load_buf<NY, false>(hst, rsh, vh, 0);
    load_a(a0, 0);
    store_h(smem);
    __syncthreads();
    read_hv(hv[0], smem, std::integral_constant<int, 0>{});
    const int last = nt - 1;
    int t = 0;
    if constexpr (A2) {
        if constexpr ((ABL & 4) != 0) load_a(a1, 0);
        for (; t + 2 <= nt; t += 2) {
            tile(std::integral_constant<int, 0>{}, a0, a1, t + 1);
            tile(std::integral_constant<int, 1>{}, a1, a0, t + 2 < nt ? t + 2 : last);
        }
        if (t < nt) tile(std::integral_constant<int, 0>{}, a0, a1, last);
    } else {
        for (; t + 2 <= nt; t += 2) {
            tile(std::integral_constant<int, 0>{}, a0, a1, t + 1);
            tile(std::integral_constant<int, 1>{}, a0, a1, t + 2 < nt ? t + 2 : last);
        }
        if (t < nt) tile(std::integral_constant<int, 0>{}, a0, a1, last);
    }

    // out[jt] (reg, lane): j = jt*32 + crow(reg, h), i = arow; registers 4g..4g+3 are 4 consecutive j
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        float* dst = p.out + (long)blockIdx.y * p.split_stride + (arow + 32 * m) * p.ldo;
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(dst + jt * 32 + 8 * g + 4 * h) =
                    f32x4{out[m][jt][4 * g], out[m][jt][4 * g + 1], out[m][jt][4 * g + 2], out[m][jt][4 * g + 3]};
    }
}

}  // namespace
