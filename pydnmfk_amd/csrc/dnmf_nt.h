// dnmf_nt.h -- NT form: C[i][j] = sum_c X[i][c] Y[j][c] through LDS (A H^T, H H^T, W (H H^T); fp32 and bf16-stored X).
// Part of libdnmf_hip.so (kernels live in anonymous namespaces of the headers; the translation units csrc/*.hip include what they launch).
#pragma once
#include "dnmf_common.h"

namespace {

// =============================================================================================== NT form
constexpr int BK = 32;  // contraction tile (floats): 128-B LDS rows

// LDS tile = rows x 32 floats; the eight 16-B chunks of a row are XOR-swizzled with (row >> 1) & 7 so that
// a ds_read_b128 by lanes (row = li, chunk = 2s + h) is bank-conflict free (rows of one 16-lane group map
// to distinct 16-B slots of the 256-B bank row).
__device__ __forceinline__ int lds_idx(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 2); }

enum { NT_STORE = 0, NT_FUSED_W = 1 };

// The small operand given as COLUMN BLOCKS [n / blk][yrows][blk] (the receive buffer of an allgather of k x blk slices, 2D
// grids: dist_nmf.py:195-197): the tile at absolute column c0 of block q = c0 / blk lives at Y + q * extra with ldy = blk,
// extra = (yrows - 1) * blk.  q comes from ONE scalar multiply-high on 16-column units (blk is a multiple of 32, so a block
// has at least two units and magic = ceil(2^32 / (blk / 16)) fits 32 bits; exact while (n / 16) * (blk / 16) < 2^32, which
// the host checks) -- branch free: a plain k x n matrix is magic = extra = 0.  (A first version tested `tiles == 0` per
// tile: two scalar branches in every k-tile of the main loops.)
struct YBlk { unsigned magic; long extra; };
__device__ __forceinline__ long yblk_off(const YBlk& yb, long c0) {
    return (long)__umulhi((unsigned)(c0 >> 4), yb.magic) * yb.extra;
}

struct NtArgs {
    const void* X; long ldx; long nrows; long ncols;   // streamed operand (float, or bf16 bits: TX of nt_kernel); contraction over ncols
    const float* Y; long ldy; int yrows;               // small operand [yrows x ncols]
    long cols_per_split;                               // contraction range per blockIdx.y (multiple of BK)
    float* out; long ldo; long split_stride; int store_all;
    float* W; long ldw; const float* G; float eps; int k;   // NT_FUSED_W
    int wfast;                                              // NT_FUSED_W: rows of W are 16-byte aligned (k, ldw % 4 == 0)
    YBlk yb;                                                // Y as column blocks (zero = one matrix)
};
__device__ __forceinline__ void rebase_args(NtArgs& p, const BatchTab& bt) {
    rebase(p.X, bt); rebase(p.Y, bt); rebase(p.out, bt); rebase(p.W, bt); rebase(p.G, bt);
}

// Stage a tile of R rows x BK floats: thread t owns 16-B chunk (t & 7) of rows (t >> 3) + it * T/8.
// INTERIOR (compile time): the whole tile is in bounds -> plain loads with no exec-masked branches, so hipcc can
// keep several tiles' loads in flight with counted vmcnt instead of draining with vmcnt(0).
template <int R, int T, bool FAST, bool INTERIOR, bool NTL = false, typename TX = float>
__device__ __forceinline__ void stage_load(f32x4 (&v)[(R + T / 8 - 1) / (T / 8)], const TX* __restrict__ X, long ldx,
                                           long nrows, long cend, long row0, long c0, int tid) {
    constexpr int RP = T / 8, NP = (R + RP - 1) / RP;
    const int ch = tid & 7;
    const long c = c0 + ch * 4;
    if constexpr (FAST && INTERIOR) {
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int rl = it * RP + (tid >> 3);
            if (R % RP == 0 || rl < R) {
                if constexpr (std::is_same<TX, float>::value) {
                    const f32x4* src = reinterpret_cast<const f32x4*>(X + (row0 + rl) * ldx + c);
                    v[it] = NTL ? __builtin_nontemporal_load(src) : *src;
                } else {
                    float d[4];
                    if constexpr (NTL) load_vec_raw_nt<4>(d, X + (row0 + rl) * ldx + c);
                    else load_vec_raw<4>(d, X + (row0 + rl) * ldx + c);
                    v[it] = f32x4{d[0], d[1], d[2], d[3]};
                }
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int rl = it * RP + (tid >> 3);
            const long r = row0 + rl;
            float d[4];
            load_vec<4, FAST>(d, X + r * ldx, c, cend, r < nrows && (R % RP == 0 || rl < R));
            v[it] = f32x4{d[0], d[1], d[2], d[3]};
        }
    }
}

// MUBUF flavour of the interior stage_load (round 3): the thread's chunk of row it * T/8 + (tid >> 3) sits at byte offset
// vo[it] from the descriptor's base (the tile origin of the workgroup), the k-tile at the wave-uniform byte offset `so` -- no
// vector address arithmetic per load (the pointer form costs ~2 v_lshl_add_u64 per load, and on gfx950 every fp32 vector
// instruction takes matrix-pipe time).  Rows beyond R carry BUF_OOB (the load returns 0, no branch).
template <int R, int T>
__device__ __forceinline__ void stage_offsets(int (&vo)[(R + T / 8 - 1) / (T / 8)], long ld, int tid) {
    constexpr int RP = T / 8, NP = (R + RP - 1) / RP;
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const int rl = it * RP + (tid >> 3);
        vo[it] = (R % RP == 0 || rl < R) ? (int)(rl * ld * 4) + (tid & 7) * 16 : BUF_OOB;
    }
}
template <int NP, bool NTL>
__device__ __forceinline__ void stage_load_buf(f32x4 (&v)[NP], i32x4 rs, const int (&vo)[NP], int so) {
#pragma unroll
    for (int it = 0; it < NP; ++it) v[it] = buf_ld_f32x4(rs, vo[it], so, NTL ? 2 : 0);
}
// one 2 GiB descriptor covers a workgroup's rows of X / all rows of Y over its column range
__device__ __forceinline__ bool buf_window_ok(long rows, long ld, long ncols) { return (rows * ld + ncols) * 4 < 0x7fffffffL; }

template <int R, int T>
__device__ __forceinline__ void stage_store(float* tile, const f32x4 (&v)[(R + T / 8 - 1) / (T / 8)], int tid) {
    constexpr int RP = T / 8, NP = (R + RP - 1) / RP;
    const int ch = tid & 7;
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const int r = it * RP + (tid >> 3);
        if (R % RP == 0 || r < R) *reinterpret_cast<f32x4*>(&tile[lds_idx(r, ch)]) = v[it];
    }
}

// LDS-DMA: one wave-instruction moves 64 x 16 B global -> LDS with no VGPR destination.  The LDS side is lane-linear
// (wave-uniform base + lane * 16 B), the global side is per lane -- so the XOR swizzle of the tile image is applied to
// the SOURCE address (cdna_hip_programming.md rule 21): lane L of the instruction that covers tile rows 8q..8q+7 fills
// (row 8q + L/8, slot L%8) and therefore fetches chunk slot ^ ((row>>1)&7) of that row.
template <bool NTL>
__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, NTL ? 2 : 0);
}

template <int R, int NW, bool NTL>
__device__ __forceinline__ void dma_tile(float* tile, const float* __restrict__ X, long ldx, long row0, long c0,
                                         int wave, int lane) {
#pragma unroll
    for (int q0 = 0; q0 < R / 8; q0 += NW) {
        const int q = q0 + wave;
        if (R / 8 % NW == 0 || q < R / 8) {
            const int row = 8 * q + (lane >> 3), chunk = (lane & 7) ^ ((row >> 1) & 7);
            glds16<NTL>(X + (row0 + row) * ldx + c0 + chunk * 4, tile + 8 * q * BK);
        }
    }
}

// sum the contraction slices of a KS > 1 workgroup: slice s > 0 parks its accumulators in LDS (lane-contiguous, conflict
// free), slice 0 adds them in slice order.  Needs NRG*MT*KT*1024*(KS-1) floats of LDS.
template <int KT, int MT, int NRG, int KS>
__device__ __forceinline__ void sum_slices(f32x16 (&acc)[MT][KT], float* smem, int rg, int ks, int lane) {
    if (ks > 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int jt = 0; jt < KT; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    smem[((((ks - 1) * NRG + rg) * MT + mt) * KT + jt) * 1024 + r * 64 + lane] = acc[mt][jt][r];
    }
    __syncthreads();
    if (ks == 0) {
#pragma unroll
        for (int q = 0; q < KS - 1; ++q)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int jt = 0; jt < KT; ++jt)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[mt][jt][r] += smem[(((q * NRG + rg) * MT + mt) * KT + jt) * 1024 + r * 64 + lane];
    }
    __syncthreads();
}

// acc[mt][jt] += X[row0 + rg*32*MT + mt*32 + .][cbeg:cend] . Y[jt*32 + .][cbeg:cend]^T
// NW waves per workgroup = (NW / KS) row groups x KS contraction slices: with KS = 2 the two waves that share a row
// group each take half of every k-tile's fragment groups and the partial accumulators are summed through LDS at the
// end (result in the slice-0 waves).  KS = 2 doubles the waves per SIMD when the shard has too few row tiles to fill
// the chip (m_l = 32768 at 8 GPUs = 256 tiles = one 4-wave workgroup per CU).
// PF = 1: the loads for tile t+1 are issued at the top of tile t (conditional on there being one: this is the generic /
// edge loop; interior tiles of fp32 X run the branch-free nt_mainloop_p2 with two tiles in flight).
template <int KT, int MT, int NW, int KS, bool FAST, int PF, bool STAGGER, bool INTERIOR, bool NTX = false, bool DMA = false, typename TX = float>
__device__ __forceinline__ void nt_mainloop_(f32x16 (&acc)[MT][KT], const TX* __restrict__ X, long ldx, long nrows,
                                            long row0, const float* __restrict__ Y, long ldy, int yrows, long cbeg,
                                            long cend, float* smem, YBlk yb = YBlk{0, 0}) {
    constexpr int NRG = NW / KS;             // row groups (waves along M)
    constexpr int BM = 32 * MT * NRG, KP = 32 * KT, T = 64 * NW;
    constexpr int STAGE = (BM + KP) * BK;    // floats per pipeline stage: [X tile | Y tile]
    constexpr int NS = BK / 8;               // fragment groups per k-tile
    static_assert(NS % KS == 0, "contraction slices must divide the fragment groups");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    const int rg = wave % NRG, ks = wave / NRG;
    f32x4 xv[(BM + T / 8 - 1) / (T / 8)], yv[(KP + T / 8 - 1) / (T / 8)];
    const long nk = (cend - cbeg + BK - 1) / BK;
    if (nk > 0) {
        // Every workgroup walks the k-tiles in a rotated order starting at a different tile: row tiles are a
        // power-of-two pitch apart in memory, so workgroups marching in lockstep over the same columns would hit
        // the same L2 / HBM channels at the same time.  (A sum over tiles: order only changes fp32 rounding.)
        const long kshift = STAGGER ? (long)((blockIdx.x * 37u) % (unsigned long)nk) : 0;
        if constexpr (DMA && INTERIOR && FAST && std::is_same<TX, float>::value) {
            // LDS-DMA staging: no staging VGPRs, no ds_write; the DMA of tile t+1 flies during the MFMAs of tile t and
            // is retired (vmcnt(0)) right before the tile barrier.
            const int wv = __builtin_amdgcn_readfirstlane(wave);
            auto issue = [&](long kt, float* stage) {
                kt += kshift;
                kt = kt >= nk ? kt - nk : kt;
                const long c0 = cbeg + kt * BK;
                dma_tile<BM, NW, NTX>(stage, X, ldx, row0, c0, wv, lane);
                dma_tile<KP, NW, false>(stage + BM * BK, Y + yblk_off(yb, c0), ldy, 0, c0, wv, lane);
            };
            issue(0, smem);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (long kt = 0; kt < nk; ++kt) {
                const int cur = kt & 1;
                if (kt + 1 < nk) issue(kt + 1, smem + (cur ^ 1) * STAGE);
                const float* xc = smem + cur * STAGE;
                const float* yc = xc + BM * BK;
#pragma unroll
                for (int sl = 0; sl < NS / KS; ++sl) {
                    const int s = ks * (NS / KS) + sl;
                    f32x4 a[MT], b[KT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        a[mt] = *reinterpret_cast<const f32x4*>(&xc[lds_idx(rg * 32 * MT + mt * 32 + li, 2 * s + h)]);
#pragma unroll
                    for (int jt = 0; jt < KT; ++jt)
                        b[jt] = *reinterpret_cast<const f32x4*>(&yc[lds_idx(jt * 32 + li, 2 * s + h)]);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int jt = 0; jt < KT; ++jt) acc[mt][jt] = MFMA32(a[mt][e], b[jt][e], acc[mt][jt]);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        } else {
        {
            const long c0 = cbeg + kshift * BK;
            stage_load<BM, T, FAST, INTERIOR, NTX>(xv, X, ldx, nrows, cend, row0, c0, tid);
            stage_load<KP, T, FAST, INTERIOR>(yv, Y + yblk_off(yb, c0), ldy, yrows, cend, 0, c0, tid);
        }
        stage_store<BM, T>(smem, xv, tid);
        stage_store<KP, T>(smem + BM * BK, yv, tid);
        __syncthreads();
        // MFMAs of one staged tile (this wave's share of its fragment groups)
        auto compute = [&](const float* xc) {
            const float* yc = xc + BM * BK;
#pragma unroll
            for (int sl = 0; sl < NS / KS; ++sl) {
                const int s = ks * (NS / KS) + sl;
                f32x4 a[MT], b[KT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    a[mt] = *reinterpret_cast<const f32x4*>(&xc[lds_idx(rg * 32 * MT + mt * 32 + li, 2 * s + h)]);
#pragma unroll
                for (int jt = 0; jt < KT; ++jt)
                    b[jt] = *reinterpret_cast<const f32x4*>(&yc[lds_idx(jt * 32 + li, 2 * s + h)]);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int jt = 0; jt < KT; ++jt) acc[mt][jt] = MFMA32(a[mt][e], b[jt][e], acc[mt][jt]);
            }
        };
        auto load_tile = [&](f32x4 (&xr)[(BM + T / 8 - 1) / (T / 8)], f32x4 (&yr)[(KP + T / 8 - 1) / (T / 8)], long kt) {
            kt += kshift;                     // rotated tile order (see kshift)
            kt = kt >= nk ? kt - nk : kt;
            const long c0 = cbeg + kt * BK;
            stage_load<BM, T, FAST, INTERIOR, NTX>(xr, X, ldx, nrows, cend, row0, c0, tid);
            stage_load<KP, T, FAST, INTERIOR>(yr, Y + yblk_off(yb, c0), ldy, yrows, cend, 0, c0, tid);
        };
        static_assert(PF == 1, "one k-tile in flight here; two tiles in flight: nt_mainloop_p2");
        for (long kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const bool more = kt + 1 < nk;
            if (more) load_tile(xv, yv, kt + 1);
            compute(smem + cur * STAGE);
            if (more) {
                stage_store<BM, T>(smem + (cur ^ 1) * STAGE, xv, tid);
                stage_store<KP, T>(smem + (cur ^ 1) * STAGE + BM * BK, yv, tid);
            }
            __syncthreads();
        }
        }   // register-staged path
    }
    if constexpr (KS > 1) {
        static_assert(NRG * MT * KT * 1024 * (KS - 1) <= 2 * STAGE, "reduction buffer exceeds the staging LDS");
        sum_slices<KT, MT, NRG, KS>(acc, smem, rg, ks, lane);
    }
}


// Interior tiles, two k-tiles in flight (fp32 X).  The PF = 1 loop above keeps ONE tile of loads in flight per
// workgroup; a shard with only as many row tiles as CUs (m_l = 32768: one 4-wave workgroup per CU) is then paced by
// the HBM latency, not by the MFMAs (MFMA busy 65 % vs 83 % with two workgroups per CU).  Here the loads of tile t+2
// are issued at the top of tile t into a second register set, and the tile that arrived one tile ago is written to
// the other LDS stage BEFORE the last fragment group, so its ds_writes and the barrier overlap MFMAs.  The loop body
// is branch-free (two tiles per trip, prefetches past the end clamp to the last tile and are never used): with
// conditional loads hipcc drains vmcnt(0) at every join and the second tile in flight is lost.
template <int KT, int MT, int NW, int KS, bool STAGGER, bool NTX>
__device__ __forceinline__ void nt_mainloop_p2(f32x16 (&acc)[MT][KT], const float* __restrict__ X, long ldx, long row0,
                                               const float* __restrict__ Y, long ldy, long cbeg, long nk, float* smem,
                                               YBlk yb = YBlk{0, 0}) {
    constexpr int NRG = NW / KS, BM = 32 * MT * NRG, KP = 32 * KT, T = 64 * NW;
    constexpr int STAGE = (BM + KP) * BK, NS = BK / 8 / KS;   // NS = fragment groups per tile of ONE wave (slice ks)
    constexpr int NPX = (BM + T / 8 - 1) / (T / 8), NPY = (KP + T / 8 - 1) / (T / 8);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    const int rg = wave % NRG, ks = wave / NRG;
    f32x4 x0[NPX], y0[NPY], x1[NPX], y1[NPY];
    const int kshift = STAGGER ? (int)((blockIdx.x * 37u) % (unsigned)nk) : 0;
    const i32x4 rsx = buf_rsrc(X + row0 * ldx + cbeg), rsy = buf_rsrc(Y + cbeg);
    int vox[NPX], voy[NPY];
    stage_offsets<BM, T>(vox, ldx, tid);
    stage_offsets<KP, T>(voy, ldy, tid);
    const int nki = (int)nk;
    auto load = [&](f32x4 (&xr)[NPX], f32x4 (&yr)[NPY], long ktl) {
        int kt = (int)ktl;
        kt = kt < nki ? kt : nki - 1;
        kt += kshift;
        kt = kt >= nki ? kt - nki : kt;
        const int so = kt * (BK * 4);                      // wave uniform
        stage_load_buf<NPX, NTX>(xr, rsx, vox, so);
        stage_load_buf<NPY, false>(yr, rsy, voy, so + (int)(yblk_off(yb, cbeg + (long)kt * BK) * 4));
    };
    auto store = [&](float* st, const f32x4 (&xr)[NPX], const f32x4 (&yr)[NPY]) {
        stage_store<BM, T>(st, xr, tid);
        stage_store<KP, T>(st + BM * BK, yr, tid);
    };
    auto group = [&](const float* xc, int sl) {
        const float* yc = xc + BM * BK;
        const int s = ks * NS + sl;
        f32x4 a[MT], b[KT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            a[mt] = *reinterpret_cast<const f32x4*>(&xc[lds_idx(rg * 32 * MT + mt * 32 + li, 2 * s + h)]);
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
            b[jt] = *reinterpret_cast<const f32x4*>(&yc[lds_idx(jt * 32 + li, 2 * s + h)]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int jt = 0; jt < KT; ++jt) acc[mt][jt] = MFMA32(a[mt][e], b[jt][e], acc[mt][jt]);
    };
    float* st0 = smem;
    float* st1 = smem + STAGE;
    load(x0, y0, 0);
    store(st0, x0, y0);
    __syncthreads();
    load(x1, y1, 1);
    long kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
        load(x0, y0, kt + 2);
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) group(st0, s);
        __builtin_amdgcn_sched_barrier(0);
        store(st1, x1, y1);
        __builtin_amdgcn_sched_barrier(0);
        group(st0, NS - 1);
        __syncthreads();
        load(x1, y1, kt + 3);
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) group(st1, s);
        __builtin_amdgcn_sched_barrier(0);
        store(st0, x0, y0);
        __builtin_amdgcn_sched_barrier(0);
        group(st1, NS - 1);
        __syncthreads();
    }
    if (kt < nk) {
#pragma unroll
        for (int s = 0; s < NS; ++s) group(st0, s);
        __syncthreads();
    }
    if constexpr (KS > 1) {
        static_assert(NRG * MT * KT * 1024 * (KS - 1) <= 2 * STAGE, "reduction buffer exceeds the staging LDS");
        sum_slices<KT, MT, NRG, KS>(acc, smem, rg, ks, lane);
    }
}


// nt_mainloop_p2 with THREE k-tiles in flight (three register sets).  A shard with at most one workgroup per CU (32768
// rows) has a single wave per SIMD and nothing else to cover the loaded HBM latency, which exceeds two tile times
// there: 0.43 -> 0.39-0.40 ms at 32768 x 8192, k = 64 (four in flight: 0.41).  No gain once two or three workgroups
// share a CU, and the extra registers cost 2 % at 262144 rows, so the launch picks this loop from the grid size.
template <int KT, int MT, int NW, bool STAGGER, bool NTX>
__device__ __forceinline__ void nt_mainloop_p3t(f32x16 (&acc)[MT][KT], const float* __restrict__ X, long ldx, long row0,
                                                 const float* __restrict__ Y, long ldy, long cbeg, long nk, float* smem,
                                                 YBlk yb = YBlk{0, 0}) {
    constexpr int BM = 32 * MT * NW, KP = 32 * KT, T = 64 * NW;
    constexpr int STAGE = (BM + KP) * BK, NS = BK / 8;
    constexpr int NPX = (BM + T / 8 - 1) / (T / 8), NPY = (KP + T / 8 - 1) / (T / 8);
    const int tid = threadIdx.x, lane = tid & 63, rg = tid >> 6, li = lane & 31, h = lane >> 5;
    f32x4 x0[NPX], y0[NPY], x1[NPX], y1[NPY], x2[NPX], y2[NPY];
    const int kshift = STAGGER ? (int)((blockIdx.x * 37u) % (unsigned)nk) : 0;
    const i32x4 rsx = buf_rsrc(X + row0 * ldx + cbeg), rsy = buf_rsrc(Y + cbeg);
    int vox[NPX], voy[NPY];
    stage_offsets<BM, T>(vox, ldx, tid);
    stage_offsets<KP, T>(voy, ldy, tid);
    const int nki = (int)nk;
    auto load = [&](f32x4 (&xr)[NPX], f32x4 (&yr)[NPY], long ktl) {
        int kt = (int)ktl;
        kt = kt < nki ? kt : nki - 1;
        kt += kshift;
        kt = kt >= nki ? kt - nki : kt;
        const int so = kt * (BK * 4);                      // wave uniform
        stage_load_buf<NPX, NTX>(xr, rsx, vox, so);
        stage_load_buf<NPY, false>(yr, rsy, voy, so + (int)(yblk_off(yb, cbeg + (long)kt * BK) * 4));
    };
    auto store = [&](float* st, const f32x4 (&xr)[NPX], const f32x4 (&yr)[NPY]) {
        stage_store<BM, T>(st, xr, tid);
        stage_store<KP, T>(st + BM * BK, yr, tid);
    };
    auto group = [&](const float* xc, int s) {
        const float* yc = xc + BM * BK;
        f32x4 a[MT], b[KT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            a[mt] = *reinterpret_cast<const f32x4*>(&xc[lds_idx(rg * 32 * MT + mt * 32 + li, 2 * s + h)]);
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
            b[jt] = *reinterpret_cast<const f32x4*>(&yc[lds_idx(jt * 32 + li, 2 * s + h)]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int jt = 0; jt < KT; ++jt) acc[mt][jt] = MFMA32(a[mt][e], b[jt][e], acc[mt][jt]);
    };
    // tile in `cur`; (xl, yl) receive tile t+3; (xr, yr) hold tile t+1, written into `nxt`
    auto tile = [&](const float* cur, float* nxt, f32x4 (&xl)[NPX], f32x4 (&yl)[NPY], const f32x4 (&xr)[NPX],
                    const f32x4 (&yr)[NPY], long t3) {
        load(xl, yl, t3);
#pragma unroll
        for (int s = 0; s < NS - 1; ++s) group(cur, s);
        __builtin_amdgcn_sched_barrier(0);
        store(nxt, xr, yr);
        __builtin_amdgcn_sched_barrier(0);
        group(cur, NS - 1);
        __syncthreads();
    };
    float* st0 = smem;
    float* st1 = smem + STAGE;
    load(x0, y0, 0);
    store(st0, x0, y0);
    __syncthreads();
    load(x1, y1, 1);
    load(x2, y2, 2);
    long kt = 0;
    // invariant at the top of a trip: st0 = tile kt, x1 = tile kt+1, x2 = tile kt+2, x0 free (6 tiles: lcm of the three
    // register sets and the two LDS stages)
    for (; kt + 6 <= nk; kt += 6) {
        tile(st0, st1, x0, y0, x1, y1, kt + 3);
        tile(st1, st0, x1, y1, x2, y2, kt + 4);
        tile(st0, st1, x2, y2, x0, y0, kt + 5);
        tile(st1, st0, x0, y0, x1, y1, kt + 6);
        tile(st0, st1, x1, y1, x2, y2, kt + 7);
        tile(st1, st0, x2, y2, x0, y0, kt + 8);
    }
    const long rem = nk - kt;   // 0..5 tiles left: the same sequence, stopping after the last real tile
    if (rem >= 1) tile(st0, st1, x0, y0, x1, y1, kt + 3);
    if (rem >= 2) tile(st1, st0, x1, y1, x2, y2, kt + 4);
    if (rem >= 3) tile(st0, st1, x2, y2, x0, y0, kt + 5);
    if (rem >= 4) tile(st1, st0, x0, y0, x1, y1, kt + 6);
    if (rem >= 5) tile(st0, st1, x1, y1, x2, y2, kt + 7);
}

template <int KT, int MT, int NW, int KS, bool FAST, int PF = 1, bool STAGGER = false, bool NTX = false, bool DMA = false, typename TX = float>
__device__ __forceinline__ void nt_mainloop(f32x16 (&acc)[MT][KT], const TX* __restrict__ X, long ldx, long nrows,
                                            long row0, const float* __restrict__ Y, long ldy, int yrows, long cbeg,
                                            long cend, float* smem, YBlk yb = YBlk{0, 0}) {
    constexpr int BM = 32 * MT * (NW / KS), KP = 32 * KT;
    // block-uniform: every tile this workgroup stages is fully in bounds
    const bool interior = FAST && row0 + BM <= nrows && yrows >= KP && (cend - cbeg) % BK == 0;
    // the two- / three-tile loops address their tiles through MUBUF descriptors: 2 GiB windows from the tile origin
    const bool bufok = buf_window_ok(BM, ldx, cend - cbeg) &&
                       buf_window_ok(KP, ldy, (cend - cbeg) + yblk_off(yb, cend - 1));   // (blocked Y: up to the last block)
    constexpr int PF1 = (PF == 3 || PF == 4) ? 1 : PF;
    if constexpr (PF == 4 && KS == 1 && std::is_same<TX, float>::value) {
        if (interior && bufok) nt_mainloop_p3t<KT, MT, NW, STAGGER, NTX>(acc, X, ldx, row0, Y, ldy, cbeg, (cend - cbeg) / BK, smem, yb);
        else nt_mainloop_<KT, MT, NW, KS, FAST, 1, STAGGER, false>(acc, X, ldx, nrows, row0, Y, ldy, yrows, cbeg, cend, smem, yb);
    } else if constexpr (PF == 3 && std::is_same<TX, float>::value) {
        if (interior && bufok) nt_mainloop_p2<KT, MT, NW, KS, STAGGER, NTX>(acc, X, ldx, row0, Y, ldy, cbeg, (cend - cbeg) / BK, smem, yb);
        else nt_mainloop_<KT, MT, NW, KS, FAST, 1, STAGGER, false>(acc, X, ldx, nrows, row0, Y, ldy, yrows, cbeg, cend, smem, yb);
    } else {
        if (interior) nt_mainloop_<KT, MT, NW, KS, FAST, PF1, STAGGER, true, NTX, DMA>(acc, X, ldx, nrows, row0, Y, ldy, yrows, cbeg, cend, smem, yb);
        else nt_mainloop_<KT, MT, NW, KS, FAST, PF1, STAGGER, false>(acc, X, ldx, nrows, row0, Y, ldy, yrows, cbeg, cend, smem, yb);
    }
}


// ---------------------------------------------------------------------------------------------- NT form, bf16-stored X
// Same structure as nt_mainloop_ (register-staged double-buffered LDS tiles, one barrier per k-tile), but a k-tile is
// BKH = 64 contraction indices: the X tile is kept in LDS as it is in HBM (bf16, 128 B per row = the same bytes, the
// same 16-B-per-lane full-line loads and the same swizzled image as an fp32 tile of 32) and widened to fp32 only after
// the fragment read; the fp32 Y tile is 64 floats (256 B = one whole LDS bank row) per row.  A ds_read_b128 of X gives
// lane (li, h) the 8 contraction indices 8*(2s+h)..+7 of row li; the matching Y values are two ds_read_b128.
constexpr int BKH = 64;
// Y tile: row pitch = all 64 banks, so the 16 lanes of a read group (consecutive rows, same chunk) must land in 16
// different 16-B slots: XOR with the low 4 row bits.
__device__ __forceinline__ int ydx(int row, int chunk) { return row * BKH + ((chunk ^ (row & 15)) << 2); }

template <int R, int T, bool FAST, bool INTERIOR>
__device__ __forceinline__ void stage_load_y64(f32x4 (&v)[(R + T / 16 - 1) / (T / 16)], const float* __restrict__ Y,
                                               long ldy, int yrows, long cend, long c0, int tid) {
    constexpr int RP = T / 16, NP = (R + RP - 1) / RP;
    const int ch = tid & 15;
    const long c = c0 + ch * 4;
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const int rl = it * RP + (tid >> 4);
        if constexpr (FAST && INTERIOR) {
            // rows >= yrows (k < KP) read a clamped, valid row instead of being predicated: an MFMA output column
            // depends only on the matching B-operand lane, so they only pollute output columns >= k, never stored
            const int rc = rl < yrows ? rl : yrows - 1;
            if (R % RP == 0 || rl < R) v[it] = *reinterpret_cast<const f32x4*>(Y + (long)rc * ldy + c);
        } else {
            float d[4];
            load_vec<4, FAST>(d, Y + (long)rl * ldy, c, cend, rl < yrows && (R % RP == 0 || rl < R));
            v[it] = f32x4{d[0], d[1], d[2], d[3]};
        }
    }
}

template <int R, int T>
__device__ __forceinline__ void stage_store_y64(float* tile, const f32x4 (&v)[(R + T / 16 - 1) / (T / 16)], int tid) {
    constexpr int RP = T / 16, NP = (R + RP - 1) / RP;
    const int ch = tid & 15;
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const int r = it * RP + (tid >> 4);
        if (R % RP == 0 || r < R) *reinterpret_cast<f32x4*>(&tile[ydx(r, ch)]) = v[it];
    }
}

// X tile of R rows x 64 bf16, raw: thread t owns the 16-B chunk (t & 7) = elements 8*(t&7)..+7 of rows (t >> 3) + it*T/8
template <int R, int T, bool FAST, bool INTERIOR, bool NTL>
__device__ __forceinline__ void stage_load_xb(f32x4 (&v)[(R + T / 8 - 1) / (T / 8)], const bf16_t* __restrict__ X, long ldx,
                                              long nrows, long cend, long row0, long c0, int tid) {
    constexpr int RP = T / 8, NP = (R + RP - 1) / RP;
    const int ch = tid & 7;
    const long c = c0 + ch * 8;
#pragma unroll
    for (int it = 0; it < NP; ++it) {
        const int rl = it * RP + (tid >> 3);
        const long r = row0 + rl;
        if constexpr (FAST && INTERIOR) {
            if (R % RP == 0 || rl < R) {
                const f32x4* src = reinterpret_cast<const f32x4*>(X + r * ldx + c);
                v[it] = NTL ? __builtin_nontemporal_load(src) : *src;
            }
        } else {
            const bool ok = r < nrows && (R % RP == 0 || rl < R);
            const bf16_t* row = X + r * ldx;
            unsigned int w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned int lo = (ok && c + 2 * q < cend) ? row[c + 2 * q] : 0u;
                const unsigned int hi = (ok && c + 2 * q + 1 < cend) ? row[c + 2 * q + 1] : 0u;
                w[q] = lo | (hi << 16);
            }
            v[it] = f32x4{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
        }
    }
}

template <int KT, int MT, int NW, bool FAST, bool STAGGER, bool INTERIOR, bool NTX>
__device__ __forceinline__ void nt_mainloop_b16_(f32x16 (&acc)[MT][KT], const bf16_t* __restrict__ X, long ldx, long nrows,
                                                long row0, const float* __restrict__ Y, long ldy, int yrows, long cbeg,
                                                long cend, float* smem) {
    constexpr int BM = 32 * MT * NW, KP = 32 * KT, T = 64 * NW;
    constexpr int XT = BM * BK;              // floats (= 4-byte words) of the raw X tile: BM rows x 128 B
    constexpr int STAGE = XT + KP * BKH;     // [X tile raw | Y tile fp32]
    const int tid = threadIdx.x, lane = tid & 63, rg = tid >> 6, li = lane & 31, h = lane >> 5;
    f32x4 xv[(BM + T / 8 - 1) / (T / 8)], yv[(KP + T / 16 - 1) / (T / 16)];
    const long nk = (cend - cbeg + BKH - 1) / BKH;
    if (nk <= 0) return;
    const long kshift = STAGGER ? (long)((blockIdx.x * 37u) % (unsigned long)nk) : 0;   // see nt_mainloop_
    auto load_tile = [&](long kt) {
        kt += kshift;
        kt = kt >= nk ? kt - nk : kt;
        const long c0 = cbeg + kt * BKH;
        stage_load_xb<BM, T, FAST, INTERIOR, NTX>(xv, X, ldx, nrows, cend, row0, c0, tid);
        stage_load_y64<KP, T, FAST, INTERIOR>(yv, Y, ldy, yrows, cend, c0, tid);
    };
    auto store_tile = [&](float* stage) {
        stage_store<BM, T>(stage, xv, tid);
        stage_store_y64<KP, T>(stage + XT, yv, tid);
    };
    auto compute = [&](const float* xc) {
        const float* yc = xc + XT;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 a[MT], b0[KT], b1[KT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                a[mt] = *reinterpret_cast<const f32x4*>(&xc[lds_idx(rg * 32 * MT + mt * 32 + li, 2 * s + h)]);
#pragma unroll
            for (int jt = 0; jt < KT; ++jt) {
                b0[jt] = *reinterpret_cast<const f32x4*>(&yc[ydx(jt * 32 + li, 2 * (2 * s + h))]);
                b1[jt] = *reinterpret_cast<const f32x4*>(&yc[ydx(jt * 32 + li, 2 * (2 * s + h) + 1)]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const unsigned int w = __float_as_uint(a[mt][e >> 1]);
                    const float av = (e & 1) ? bf16_hi(w) : bf16_lo(w);
#pragma unroll
                    for (int jt = 0; jt < KT; ++jt)
                        acc[mt][jt] = MFMA32(av, e < 4 ? b0[jt][e & 3] : b1[jt][e & 3], acc[mt][jt]);
                }
        }
    };
    load_tile(0);
    store_tile(smem);
    __syncthreads();
    for (long kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) load_tile(kt + 1);
        compute(smem + cur * STAGE);
        if (more) store_tile(smem + (cur ^ 1) * STAGE);
        __syncthreads();
    }
}

template <int KT, int MT, int NW, bool FAST, bool STAGGER, bool NTX>
__device__ __forceinline__ void nt_mainloop_b16(f32x16 (&acc)[MT][KT], const bf16_t* __restrict__ X, long ldx, long nrows,
                                               long row0, const float* __restrict__ Y, long ldy, int yrows, long cbeg,
                                               long cend, float* smem) {
    constexpr int BM = 32 * MT * NW;
    const bool interior = FAST && row0 + BM <= nrows && (cend - cbeg) % BKH == 0;   // any yrows: see stage_load_y64
    if (interior) nt_mainloop_b16_<KT, MT, NW, FAST, STAGGER, true, NTX>(acc, X, ldx, nrows, row0, Y, ldy, yrows, cbeg, cend, smem);
    else nt_mainloop_b16_<KT, MT, NW, FAST, STAGGER, false, false>(acc, X, ldx, nrows, row0, Y, ldy, yrows, cbeg, cend, smem);
}

template <int KT, int MT, int NW, int KS, bool FAST, int MODE, int PF, typename TX = float>
__global__ __launch_bounds__(64 * NW) void nt_kernel(NtArgs p, BatchTab bt) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    rebase_args(p, bt);
    constexpr int NRG = NW / KS, BM = 32 * MT * NRG;
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int wave = (threadIdx.x >> 6) % NRG, ks = (threadIdx.x >> 6) / NRG;   // row group, contraction slice
    const long row0 = (long)blockIdx.x * BM;

    f32x16 acc[MT][KT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int jt = 0; jt < KT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][jt][r] = 0.f;

    if constexpr (MODE == NT_STORE || MODE == NT_FUSED_W) {
        const long cbeg = (long)blockIdx.y * p.cols_per_split;
        long cend = cbeg + p.cols_per_split;
        if (cend > p.ncols) cend = p.ncols;
        if constexpr (std::is_same<TX, bf16_t>::value) {
            static_assert(KS == 1, "bf16 X: one contraction slice");
            nt_mainloop_b16<KT, MT, NW, FAST, (PF == 5), (PF == 5)>(acc, static_cast<const bf16_t*>(p.X), p.ldx, p.nrows, row0, p.Y, p.ldy, p.yrows, cbeg, cend, smem);
        } else {
            nt_mainloop<KT, MT, NW, KS, FAST, (PF == 10 ? 3 : PF == 13 ? 4 : 1), (PF == 5 || PF >= 7), (PF >= 5), (PF == 7)>(acc, static_cast<const float*>(p.X), p.ldx, p.nrows, row0, p.Y, p.ldy, p.yrows, cbeg, cend, smem, p.yb);
        }
    }

    if constexpr (MODE == NT_STORE) {
        if (KS > 1 && ks != 0) return;       // the sums live in the slice-0 waves
        float* out = p.out + (long)blockIdx.y * p.split_stride;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int jt = 0; jt < KT; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long row = row0 + wave * 32 * MT + mt * 32 + crow(r, h);
                    const int col = jt * 32 + li;
                    if (p.store_all || (row < p.nrows && col < p.yrows)) out[row * p.ldo + col] = acc[mt][jt][r];
                }
    } else {
        // second product: acc2 = W[rows] . G  (G = H H^T is symmetric, so G[j][jj] serves as Y[j][c = jj])
        f32x16 acc2[MT][KT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int jt = 0; jt < KT; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[mt][jt][r] = 0.f;
        // W's alignment is independent of A's (an NMFk sweep visits k = 2, 3, 5 ...): block-uniform choice
        if (FAST && p.wfast) nt_mainloop<KT, MT, NW, KS, FAST>(acc2, p.W, p.ldw, p.nrows, row0, p.G, 32 * KT, 32 * KT, 0, p.k, smem);
        else nt_mainloop<KT, MT, NW, KS, false>(acc2, p.W, p.ldw, p.nrows, row0, p.G, 32 * KT, 32 * KT, 0, p.k, smem);
        if (KS > 1 && ks != 0) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int jt = 0; jt < KT; ++jt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long row = row0 + wave * 32 * MT + mt * 32 + crow(r, h);
                    const int col = jt * 32 + li;
                    if (row < p.nrows && col < p.k) {
                        const float ah = acc[mt][jt][r];
                        const float w = p.W[row * p.ldw + col];
                        const float q = ah / (acc2[mt][jt][r] + p.eps);   // dist_nmf.py:731-732
                        p.W[row * p.ldw + col] = w * q;
                    }
                }
    }
}


}  // namespace
