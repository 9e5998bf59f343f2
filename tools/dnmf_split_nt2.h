// tools/dnmf_split_nt2.h (round 4: moved out of pydnmfk_amd/csrc -- an experiment that measured no gain, kept for tools/ntxproto.hip)
// dnmf_split_nt2.h -- A H^T in the bf16x6 arithmetic, second main loop: A is cut BEFORE it goes to LDS, into a tile that is
// private to the wave that owns its 32 rows.
//
// ntx_mainloop (dnmf_split.h) stages A as fp32 through a workgroup-shared LDS image and cuts a lane's fragment right before
// its MFMAs: the cut (44 vector instructions per step) sits in the dependency chain fragment read -> cut -> MFMA of every
// tile, between two workgroup barriers.  Here a wave loads the 32 x 32 tile of its own rows with coalesced 16-byte loads (8
// lanes per 128-byte row segment), cuts each loaded quad where it lies (the cut is elementwise, so it does not care about
// the MFMA layout) and writes the three bf16 piece images [piece][row][32 indices = 64 B] into its own 6 KiB of LDS; a
// fragment read IS the MFMA operand.  LDS operations of one wave execute in order, so the pieces of tile t + 1 are written
// while the fragment reads of tile t are still in flight -- no barrier and no second stage for A; the cut of tile t + 1 runs
// in the shadow of the MFMAs of tile t.  Only the H tile is shared (one barrier per tile, two stages, as before).
// Same products in the same order as ntx_mainloop: bit-identical results (tools/ntxproto.hip compares them).
#pragma once          // included by dnmf_split.h (after its small-operand tile helpers)

namespace {

constexpr int NT2_PX = 3 * 32 * 64;                                   // bytes of a wave's private A-piece tile
template <int KT> constexpr size_t nt2_lds_bytes() { return 4 * NT2_PX + 2 * 3 * 32 * KT * 64; }

__device__ __forceinline__ void split4(const f32x4& v, u32x2& s1, u32x2& s2, u32x2& s3) {
    unsigned int a, b, c;
    split_pair(v[0], v[1], a, b, c);
    s1[0] = a; s2[0] = b; s3[0] = c;
    split_pair(v[2], v[3], a, b, c);
    s1[1] = a; s2[1] = b; s3[1] = c;
}

template <int KT, int NSET, bool NTL, int ABL = 0>
__device__ __forceinline__ void ntx2_mainloop(f32x16 (&acc)[1][KT], const float* __restrict__ X, long ldx, long row0,
                                              const SplitOperand& ys, long cbeg, long cend, float* smem) {
    static_assert(NSET % 2 == 0, "the H register sets alternate with the tile parity");
    constexpr int CR = 4, HROWS = 32 * KT, HPIECE = HROWS * CR * 16, HB = 3 * HPIECE, T = 256, NH = HB / 16 / T;
    static_assert(HB % (16 * T) == 0, "whole 16-byte pieces per thread");
    char* lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* px = lds + wave * NT2_PX;
    char* hst = lds + 4 * NT2_PX;
    const int nk = (int)((cend - cbeg) / XT);
    if (nk <= 0) return;
    const int kshift = (int)((blockIdx.x * 37u) % (unsigned)nk);
    auto col_of = [&](int t) {                                       // column offset of tile t from cbeg
        t = t < nk ? t : nk - 1;
        t += kshift;
        t = t >= nk ? t - nk : t;
        return t * XT;
    };
    // A: lane (r8 = lane >> 3, c = lane & 7) of load i reads the 16 bytes c of row 8 i + r8 of the wave's 32 rows
    const i32x4 rsx = buf_rsrc(X + (row0 + wave * 32) * ldx + cbeg);
    const int vox = (lane >> 3) * (int)(ldx * 4) + (lane & 7) * 16;
    const int rstep = (int)(ldx * 32);                               // 8 rows in bytes
    // ... and writes its four indices (8 bytes per piece) to chunk c >> 1 of that row; chunks are XOR-swizzled with
    // (row >> 2) & 3 = (2 i + (r8 >> 2)) & 3, i.e. loads 1 and 3 flip bit 1 of the chunk of loads 0 and 2
    const int wq = ((lane & 7) >> 1) ^ ((lane >> 5) & 1);
    const int wofs0 = (lane >> 3) * 64 + wq * 16 + (lane & 1) * 8;
    const int wofs1 = (lane >> 3) * 64 + (wq ^ 2) * 16 + (lane & 1) * 8;
    // fragment of MFMA step u: chunk 2 u + h of row li (the same offsets in the A-piece tile and in every 32-row block of H)
    const int fsw = (li >> 2) & 3;
    const int fofs0 = li * 64 + ((h ^ fsw) * 16), fofs1 = li * 64 + (((2 + h) ^ fsw) * 16);
    // H tile: thread -> 16-byte pieces tid + 256 i of [piece][row][chunk]
    const i32x4 rsh = buf_rsrc(ys.S + cbeg);
    int voh[NH], hdst[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int p = tid + T * i, s = p / (HROWS * CR), row = (p / CR) % HROWS, ch = p % CR;
        voh[i] = (int)((s * ys.split_stride + row * ys.ld) * 2) + ch * 16;
        hdst[i] = ((s * HROWS + row) * CR + (ch ^ tile_swz<CR>(row))) * 16;
    }
    f32x4 xv[NSET][4];
    u32x4 hv[2][NH];
    auto loadx = [&](f32x4 (&x)[4], int t) {
        const int so = col_of(t) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = buf_ld_f32x4(rsx, vox, so + i * rstep, NTL ? 2 : 0);
    };
    auto loadh = [&](u32x4 (&hh)[NH], int t) {
        const int so = col_of(t) * 2;
#pragma unroll
        for (int i = 0; i < NH; ++i) hh[i] = __builtin_bit_cast(u32x4, buf_ld_f32x4(rsh, voh[i], so, 0));
    };
    auto cut1 = [&](const f32x4& x, int i) {                           // load i of a tile -> the wave's piece images
        u32x2 s1, s2, s3;
        if constexpr (ABL & 2) {                                       // ablation: no cut (wrong results)
            s1 = u32x2{__float_as_uint(x[0]), __float_as_uint(x[1])};
            s2 = u32x2{__float_as_uint(x[2]), __float_as_uint(x[3])};
            s3 = u32x2{__float_as_uint(x[1]), __float_as_uint(x[2])};
        } else split4(x, s1, s2, s3);
        char* d = px + ((i & 1) ? wofs1 : wofs0) + i * 512;
        *reinterpret_cast<u32x2*>(d) = s1;
        *reinterpret_cast<u32x2*>(d + 2048) = s2;
        *reinterpret_cast<u32x2*>(d + 4096) = s3;
    };
    auto storeh = [&](char* st, const u32x4 (&hh)[NH]) {
#pragma unroll
        for (int i = 0; i < NH; ++i) *reinterpret_cast<u32x4*>(st + hdst[i]) = hh[i];
    };
    auto tile = [&](const char* hc, const f32x4 (&xn)[4]) {
        u32x4 A[2][3], B[2][KT][3];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 3; ++s) A[u][s] = *reinterpret_cast<const u32x4*>(px + (u ? fofs1 : fofs0) + s * 2048);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int jt = 0; jt < KT; ++jt)
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    if constexpr ((ABL & 4) != 0) {                    // ablation: half the H fragment reads (wrong results)
                        if (jt > 0) { B[u][jt][s] = B[u][0][s]; continue; }
                    }
                    B[u][jt][s] = *reinterpret_cast<const u32x4*>(hc + (u ? fofs1 : fofs0) + jt * 2048 + s * HPIECE);
                }
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 12>([&](auto p_) {
                constexpr int u = decltype(p_)::value / 6, pr = decltype(p_)::value % 6;
                constexpr int sa = pr == 0 ? 2 : (pr == 1 || pr == 3) ? 1 : 0;
                constexpr int sb = (pr == 0 || pr == 3 || pr == 5) ? 0 : (pr == 1 || pr == 4) ? 1 : 2;
#pragma unroll
                for (int jt = 0; jt < KT; ++jt) {
                    if constexpr (ABL & 1) {                           // ablation: no MFMA (the fragments are still read)
                        if constexpr (pr == 5) asm volatile("" :: "v"(A[u][0]), "v"(A[u][1]), "v"(A[u][2]), "v"(B[u][jt][0]), "v"(B[u][jt][1]), "v"(B[u][jt][2]));
                    } else acc[0][jt] = mfma_bf16(A[u][sa], B[u][jt][sb], acc[0][jt]);
                }
                // the next tile's pieces: one load's quad after each of the first four rounds of step 0 (after ALL fragment
                // reads of this tile in program order -- the wave's LDS operations execute in that order)
                if constexpr (u == 0 && pr < 4) cut1(xn[pr], pr);
                __builtin_amdgcn_sched_barrier(0);
            });
    };
    static_for<0, NSET>([&](auto i_) { constexpr int i = decltype(i_)::value; loadx(xv[i], i); });
    loadh(hv[0], 0);
    loadh(hv[1], 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) cut1(xv[0][i], i);
    storeh(hst, hv[0]);
    __syncthreads();
    for (int t = 0; t < nk; t += NSET) {
        static_for<0, NSET>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            loadx(xv[i], t + i + NSET);                  // (tile t + i left this set when it was cut, one tile ago)
            loadh(hv[i & 1], t + i + 2);
            tile(hst + (i & 1) * HB, xv[(i + 1) % NSET]);
            storeh(hst + ((i + 1) & 1) * HB, hv[(i + 1) & 1]);
            __syncthreads();
        });
    }
}

}  // namespace
