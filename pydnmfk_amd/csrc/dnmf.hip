// dnmf.hip -- MI355X (gfx950 / CDNA4) kernels + C ABI for the distributed-NMF multiplicative-update path.
//
// Written for gfx950 only: 64-wide wavefronts, fp32-input MFMA (v_mfma_f32_32x32x2_f32, exact fp32 = an
// fmaf chain), 160 KiB LDS per CU, 8 XCDs.  See DESIGN.md for the data layout and per-kernel rooflines and
// include/dnmf.h for the reference lines each entry point replaces.
//
// Two GEMM forms carry the whole path (k = NMF rank, padded to KP = 32*KT):
//   NT form  C[i][j] = sum_c X[i][c] * Y[j][c]   contraction index contiguous in both operands
//            -> A H^T (K2), H H^T (K1), W (H H^T) (K3).  X is streamed through LDS (the MFMA operand
//               layout puts the 32 rows of a tile across lanes, so a transpose is unavoidable).
//   TN form  C[j][c] = sum_i X[i][j] * Y[i][c]   contraction index is the row of both operands
//            -> W^T A (K6), W^T W (K5), (W^T W) H (K7).  Operands go global -> VGPR directly in MFMA
//               layout with 16-byte coalesced loads; no LDS, no barriers.
// MFMA 32x32x2 f32 operand maps (lane l, li = l & 31, h = l >> 5):
//   A-operand: A[i = li][kk = h]   B-operand: B[kk = h][j = li]
//   C/D: col = li, row = (reg & 3) + 8 * (reg >> 2) + 4 * h   (reg in [0,16))
// The contraction order inside a tile and the output row/column order inside a tile are permuted freely
// (sums are order-agnostic up to fp32 rounding; outputs are written to their true addresses).
#include <type_traits>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "dnmf.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// bf16 STORAGE of the data matrix A (BASELINE config 5): A is held as bfloat16 in HBM (half the bytes of the HBM-bound
// small-k regime), widened to fp32 in registers (exact: bf16 -> fp32 is a 16-bit shift) and fed to the same fp32 MFMAs.
// W, H and every intermediate stay fp32.
typedef unsigned short bf16_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf16_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }

namespace {

// ----------------------------------------------------------------------------------------------- errors
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// hipGetLastError is sticky per thread and shared with the host framework: clear before each launch sequence
inline void clear_hip_error() { (void)hipGetLastError(); }

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DNMF_EHIP, "%s: %s", what, hipGetErrorString(e));
    return DNMF_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
__host__ __device__ inline long cdiv(long a, long b) { return (a + b - 1) / b; }
inline long round_up(long a, long b) { return cdiv(a, b) * b; }

// C/D row of accumulator register `reg` for lane-half h
__device__ __forceinline__ int crow(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ----------------------------------------------------------------------------------------------- loads
// V contiguous floats starting at column `col` of a row; zero outside [0, ncols).
// FAST: col % V == 0, ncols % 4 == 0, row pointer 16-B aligned, so a vector is wholly in or out.
template <int V, bool FAST>
__device__ __forceinline__ void load_vec(float (&d)[V], const float* __restrict__ row, long col, long ncols, bool ok) {
    if constexpr (FAST) {
        if (ok && col < ncols) {
            if constexpr (V == 4) {
                f32x4 v = *reinterpret_cast<const f32x4*>(row + col);
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            } else if constexpr (V == 2) {
                f32x2 v = *reinterpret_cast<const f32x2*>(row + col);
                d[0] = v[0]; d[1] = v[1];
            } else {
                d[0] = row[col];
            }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) d[e] = 0.f;
        }
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] = (ok && col + e < ncols) ? row[col + e] : 0.f;
    }
}

template <int V, bool FAST>
__device__ __forceinline__ void store_vec(const float (&d)[V], float* __restrict__ row, long col, long ncols, bool ok) {
    if constexpr (FAST) {
        if (ok && col < ncols) {
            if constexpr (V == 4) {
                f32x4 v = {d[0], d[1], d[2], d[3]};
                *reinterpret_cast<f32x4*>(row + col) = v;
            } else if constexpr (V == 2) {
                f32x2 v = {d[0], d[1]};
                *reinterpret_cast<f32x2*>(row + col) = v;
            } else {
                row[col] = d[0];
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e)
            if (ok && col + e < ncols) row[col + e] = d[e];
    }
}

// unconditional vector load of V floats (address must be valid)
template <int V>
__device__ __forceinline__ void load_vec_raw(float (&d)[V], const float* __restrict__ p) {
    if constexpr (V == 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(p);
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    } else if constexpr (V == 2) {
        f32x2 v = *reinterpret_cast<const f32x2*>(p);
        d[0] = v[0]; d[1] = v[1];
    } else {
        d[0] = p[0];
    }
}

template <int V>
__device__ __forceinline__ void load_vec_raw_nt(float (&d)[V], const float* __restrict__ p) {
    if constexpr (V == 4) {
        f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    } else if constexpr (V == 2) {
        f32x2 v = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(p));
        d[0] = v[0]; d[1] = v[1];
    } else {
        d[0] = __builtin_nontemporal_load(p);
    }
}

// INTERIOR (compile time) = the caller has established, with ONE block/wave-uniform test, that every access of the tile is
// in bounds: plain vector accesses, no per-lane exec-masked branches.  (hipcc serialises exec-masked loads: it
// drains with vmcnt(0) at every branch join, so a tile of N predicated loads costs N memory latencies.)
template <int V, bool FAST, bool INTERIOR>
__device__ __forceinline__ void load_tile_vec(float (&d)[V], const float* __restrict__ row, long col, long ncols, bool ok) {
    if constexpr (FAST && INTERIOR) load_vec_raw<V>(d, row + col);
    else load_vec<V, FAST>(d, row, col, ncols, ok);
}

template <int V, bool FAST, bool INTERIOR>
__device__ __forceinline__ void store_tile_vec(const float (&d)[V], float* __restrict__ row, long col, long ncols, bool ok) {
    if constexpr (FAST && INTERIOR) {
        if constexpr (V == 4) *reinterpret_cast<f32x4*>(row + col) = f32x4{d[0], d[1], d[2], d[3]};
        else if constexpr (V == 2) *reinterpret_cast<f32x2*>(row + col) = f32x2{d[0], d[1]};
        else row[col] = d[0];
    } else {
        store_vec<V, FAST>(d, row, col, ncols, ok);
    }
}

// bf16 flavours of the element loaders (V in {1, 2, 4}: 2 / 4 / 8 bytes per lane)
template <int V>
__device__ __forceinline__ void load_vec_raw(float (&d)[V], const bf16_t* __restrict__ p) {
    if constexpr (V == 4) {
        const u32x2 w = *reinterpret_cast<const u32x2*>(p);
        d[0] = bf16_lo(w[0]); d[1] = bf16_hi(w[0]); d[2] = bf16_lo(w[1]); d[3] = bf16_hi(w[1]);
    } else if constexpr (V == 2) {
        const unsigned int w = *reinterpret_cast<const unsigned int*>(p);
        d[0] = bf16_lo(w); d[1] = bf16_hi(w);
    } else {
        d[0] = bf16_lo((unsigned int)p[0]);
    }
}

template <int V>
__device__ __forceinline__ void load_vec_raw_nt(float (&d)[V], const bf16_t* __restrict__ p) {
    if constexpr (V == 4) {
        const u32x2 w = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
        d[0] = bf16_lo(w[0]); d[1] = bf16_hi(w[0]); d[2] = bf16_lo(w[1]); d[3] = bf16_hi(w[1]);
    } else if constexpr (V == 2) {
        const unsigned int w = __builtin_nontemporal_load(reinterpret_cast<const unsigned int*>(p));
        d[0] = bf16_lo(w); d[1] = bf16_hi(w);
    } else {
        d[0] = bf16_lo((unsigned int)__builtin_nontemporal_load(p));
    }
}

// Raw<T, V>: V elements as they arrive from memory.  For bf16 the widening to fp32 is deferred to get(): a conversion
// right after the load would make the wave wait for the load at once and undo the software prefetch.
template <typename T, int V> struct Raw;
template <int V> struct Raw<float, V> {
    float v[V];
    __device__ __forceinline__ void load(const float* __restrict__ p) { load_vec_raw<V>(v, p); }
    __device__ __forceinline__ void load_nt(const float* __restrict__ p) { load_vec_raw_nt<V>(v, p); }
    __device__ __forceinline__ void get(float (&d)[V]) const {
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] = v[e];
    }
};
template <int V> struct Raw<bf16_t, V> {
    unsigned int w[(V + 1) / 2];
    template <bool NTL>
    __device__ __forceinline__ void load_(const bf16_t* __restrict__ p) {
        if constexpr (V == 4) {
            const u32x2 x = NTL ? __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p)) : *reinterpret_cast<const u32x2*>(p);
            w[0] = x[0]; w[1] = x[1];
        } else if constexpr (V == 2) {
            w[0] = NTL ? __builtin_nontemporal_load(reinterpret_cast<const unsigned int*>(p)) : *reinterpret_cast<const unsigned int*>(p);
        } else {
            w[0] = NTL ? __builtin_nontemporal_load(p) : p[0];
        }
    }
    __device__ __forceinline__ void load(const bf16_t* __restrict__ p) { load_<false>(p); }
    __device__ __forceinline__ void load_nt(const bf16_t* __restrict__ p) { load_<true>(p); }
    __device__ __forceinline__ void get(float (&d)[V]) const {
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] = (e & 1) ? bf16_hi(w[e >> 1]) : bf16_lo(w[e >> 1]);
    }
};

template <int V, bool FAST>
__device__ __forceinline__ void load_vec(float (&d)[V], const bf16_t* __restrict__ row, long col, long ncols, bool ok) {
    if constexpr (FAST) {
        if (ok && col < ncols) load_vec_raw<V>(d, row + col);
        else {
#pragma unroll
            for (int e = 0; e < V; ++e) d[e] = 0.f;
        }
    } else {
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] = (ok && col + e < ncols) ? bf16_lo((unsigned int)row[col + e]) : 0.f;
    }
}

template <int V, bool FAST, bool INTERIOR>
__device__ __forceinline__ void load_tile_vec(float (&d)[V], const bf16_t* __restrict__ row, long col, long ncols, bool ok) {
    if constexpr (FAST && INTERIOR) load_vec_raw<V>(d, row + col);
    else load_vec<V, FAST>(d, row, col, ncols, ok);
}

// =============================================================================================== NT form
constexpr int BK = 32;  // contraction tile (floats): 128-B LDS rows

// LDS tile = rows x 32 floats; the eight 16-B chunks of a row are XOR-swizzled with (row >> 1) & 7 so that
// a ds_read_b128 by lanes (row = li, chunk = 2s + h) is bank-conflict free (rows of one 16-lane group map
// to distinct 16-B slots of the 256-B bank row).
__device__ __forceinline__ int lds_idx(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 2); }

enum { NT_STORE = 0, NT_FUSED_W = 1 };

struct NtArgs {
    const void* X; long ldx; long nrows; long ncols;   // streamed operand (float, or bf16 bits: TX of nt_kernel); contraction over ncols
    const float* Y; long ldy; int yrows;               // small operand [yrows x ncols]
    long cols_per_split;                               // contraction range per blockIdx.y (multiple of BK)
    float* out; long ldo; long split_stride; int store_all;
    float* W; long ldw; const float* G; float eps; int k;   // NT_FUSED_W
    int wfast;                                              // NT_FUSED_W: rows of W are 16-byte aligned (k, ldw % 4 == 0)
};

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
// PF = prefetch distance in k-tiles: 1 = loads for tile t+1 are issued at the top of tile t; 2 = one more tile is kept
// in flight in registers (loads for t+2 issued at the top of tile t, written to LDS at the end of t+1), for shards
// with so few row tiles that a CU holds a single workgroup and nothing else hides the HBM latency.
template <int KT, int MT, int NW, int KS, bool FAST, int PF, bool STAGGER, bool INTERIOR, bool NTX = false, bool DMA = false, typename TX = float>
__device__ __forceinline__ void nt_mainloop_(f32x16 (&acc)[MT][KT], const TX* __restrict__ X, long ldx, long nrows,
                                            long row0, const float* __restrict__ Y, long ldy, int yrows, long cbeg,
                                            long cend, float* smem) {
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
                dma_tile<KP, NW, false>(stage + BM * BK, Y, ldy, 0, c0, wv, lane);
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
            stage_load<KP, T, FAST, INTERIOR>(yv, Y, ldy, yrows, cend, 0, c0, tid);
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
            stage_load<KP, T, FAST, INTERIOR>(yr, Y, ldy, yrows, cend, 0, c0, tid);
        };
        if constexpr (PF == 1) {
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
        } else {
            // two register sets: (xv, yv) and (xw, yw) alternate; each holds a tile for one whole compute phase
            f32x4 xw[(BM + T / 8 - 1) / (T / 8)], yw[(KP + T / 8 - 1) / (T / 8)];
            if (nk > 1) load_tile(xv, yv, 1);
            for (long kt = 0; kt < nk; kt += 2) {
                // even phase: tile kt in LDS stage 0, tile kt+1 in flight in (xv, yv); issue tile kt+2 into (xw, yw)
                if (kt + 2 < nk) load_tile(xw, yw, kt + 2);
                compute(smem);
                if (kt + 1 < nk) {
                    stage_store<BM, T>(smem + STAGE, xv, tid);
                    stage_store<KP, T>(smem + STAGE + BM * BK, yv, tid);
                }
                __syncthreads();
                if (kt + 1 >= nk) break;
                // odd phase: tile kt+1 in stage 1, tile kt+2 in flight in (xw, yw); issue tile kt+3 into (xv, yv)
                if (kt + 3 < nk) load_tile(xv, yv, kt + 3);
                compute(smem + STAGE);
                if (kt + 2 < nk) {
                    stage_store<BM, T>(smem, xw, tid);
                    stage_store<KP, T>(smem + BM * BK, yw, tid);
                }
                __syncthreads();
            }
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
                                               const float* __restrict__ Y, long ldy, long cbeg, long nk, float* smem) {
    constexpr int NRG = NW / KS, BM = 32 * MT * NRG, KP = 32 * KT, T = 64 * NW;
    constexpr int STAGE = (BM + KP) * BK, NS = BK / 8 / KS;   // NS = fragment groups per tile of ONE wave (slice ks)
    constexpr int NPX = (BM + T / 8 - 1) / (T / 8), NPY = (KP + T / 8 - 1) / (T / 8);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
    const int rg = wave % NRG, ks = wave / NRG;
    f32x4 x0[NPX], y0[NPY], x1[NPX], y1[NPY];
    const long kshift = STAGGER ? (long)((blockIdx.x * 37u) % (unsigned long)nk) : 0;
    auto load = [&](f32x4 (&xr)[NPX], f32x4 (&yr)[NPY], long kt) {
        kt = kt < nk ? kt : nk - 1;
        kt += kshift;
        kt = kt >= nk ? kt - nk : kt;
        const long c0 = cbeg + kt * BK;
        stage_load<BM, T, true, true, NTX>(xr, X, ldx, 0, 0, row0, c0, tid);
        stage_load<KP, T, true, true>(yr, Y, ldy, KP, 0, 0, c0, tid);
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

template <int KT, int MT, int NW, int KS, bool FAST, int PF = 1, bool STAGGER = false, bool NTX = false, bool DMA = false, typename TX = float>
__device__ __forceinline__ void nt_mainloop(f32x16 (&acc)[MT][KT], const TX* __restrict__ X, long ldx, long nrows,
                                            long row0, const float* __restrict__ Y, long ldy, int yrows, long cbeg,
                                            long cend, float* smem) {
    constexpr int BM = 32 * MT * (NW / KS), KP = 32 * KT;
    // block-uniform: every tile this workgroup stages is fully in bounds
    const bool interior = FAST && row0 + BM <= nrows && yrows >= KP && (cend - cbeg) % BK == 0;
    constexpr int PF1 = PF == 3 ? 1 : PF;
    if constexpr (PF == 3 && std::is_same<TX, float>::value) {
        if (interior) nt_mainloop_p2<KT, MT, NW, KS, STAGGER, NTX>(acc, X, ldx, row0, Y, ldy, cbeg, (cend - cbeg) / BK, smem);
        else nt_mainloop_<KT, MT, NW, KS, FAST, 1, STAGGER, false>(acc, X, ldx, nrows, row0, Y, ldy, yrows, cbeg, cend, smem);
    } else {
        if (interior) nt_mainloop_<KT, MT, NW, KS, FAST, PF1, STAGGER, true, NTX, DMA>(acc, X, ldx, nrows, row0, Y, ldy, yrows, cbeg, cend, smem);
        else nt_mainloop_<KT, MT, NW, KS, FAST, PF1, STAGGER, false>(acc, X, ldx, nrows, row0, Y, ldy, yrows, cbeg, cend, smem);
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
__global__ __launch_bounds__(64 * NW) void nt_kernel(NtArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
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
            nt_mainloop<KT, MT, NW, KS, FAST, (PF == 10 ? 3 : 1), (PF == 5 || PF >= 7), (PF >= 5), (PF == 7)>(acc, static_cast<const float*>(p.X), p.ldx, p.nrows, row0, p.Y, p.ldy, p.yrows, cbeg, cend, smem);
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

// =============================================================================================== TN form
enum { TN_PARTIAL = 0 };

struct TnArgs {
    const float* X; long ldx; int xcols;     // [nrows x xcols]  -> output rows j
    const void* Y; long ldy; long ycols;     // [nrows x ycols]  -> output cols c (float, or bf16 bits: TY of tn_kernel)
    long nrows; long rows_per_chunk; int nchunks; int ncolblk;
    float* P; long chunk_stride; long ldp;   // P[chunk][KP][ldp]
};

template <int KT, int NT, bool FAST, int U, typename TY>
__device__ __forceinline__ void tn_load(float (&a)[U][KT], float (&b)[U][NT], const float* __restrict__ X, long ldx,
                                        int xcols, const TY* __restrict__ Y, long ldy, long ycols, long col0,
                                        long r, long rend, int li, int h) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long row = r + 2 * u + h;
        const bool ok = row < rend;
        load_vec<KT, FAST>(a[u], X + row * ldx, (long)KT * li, xcols, ok);
        load_vec<NT, FAST>(b[u], Y + row * ldy, col0 + (long)NT * li, ycols, ok);
    }
}

template <int KT, int NT, int U>
__device__ __forceinline__ void tn_comp(f32x16 (&acc)[KT][NT], const float (&a)[U][KT], const float (&b)[U][NT]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int ke = 0; ke < KT; ++ke)
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) acc[ke][ne] = MFMA32(a[u][ke], b[u][ne], acc[ke][ne]);
}

// acc[ke][ne] (reg, lane) = C[j = KT*crow(reg,h) + ke][c = col0 + NT*li + ne], contraction over rows [rbeg, rend)
//
// FAST path = software pipeline over full batches of U row pairs, one batch ahead, with the two loads of the NEXT
// batch's row pair u issued right before the KT*NT MFMAs of THIS batch's row pair u (issue order pinned with
// sched_barrier).  Measured on MI355X (262144 x 8192, k = 64; tools/kbench.py): this interleave 2.34 ms; the same loads
// as one block of 8 ahead of the 32 MFMAs 3.5 ms (waves stall issuing VMEM while the matrix pipe idles: MFMA busy 51 %
// vs 88 %); exec-masked predicated loads (hipcc then drains with vmcnt(0)) 2.58 ms.  Loads are branch-free: the
// batch base is a wave-uniform pointer, the per-lane part (2u + h) * ld + column a loop-invariant 32-bit offset.
// Lanes whose output row j >= xcols or output column c >= ycols read a clamped (valid) column instead: an MFMA output
// row / column depends only on the matching A- / B-operand lane, so they only pollute outputs that are never stored.
template <int KT, int NT, bool FAST, bool NTY = false, typename TY = float>
__device__ __forceinline__ void tn_mainloop(f32x16 (&acc)[KT][NT], const float* __restrict__ X, long ldx, int xcols,
                                            const TY* __restrict__ Y, long ldy, long ycols, long col0, long rbeg,
                                            long rend, int li, int h) {
    constexpr int U = 4;  // row pairs per register batch
    float a0[U][KT], b0[U][NT], a1[U][KT];
    Raw<TY, NT> q0[U], q1[U];                // the streamed operand as loaded (bf16: widened right before its MFMAs)
    long r = rbeg;
    if constexpr (FAST) {
        const long nb = (rend - rbeg) / (2 * U);
        if (nb > 0 && 8 * ldx < 0x7fffffffL && 8 * ldy < 0x7fffffffL) {
            long xc = (long)KT * li, yc = col0 + (long)NT * li;
            xc = xc < xcols ? xc : xcols - KT;
            yc = yc < ycols ? yc : ycols - NT;
            int xo[U], yo[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xo[u] = (int)((2 * u + h) * ldx + xc);
                yo[u] = (int)((2 * u + h) * ldy + yc);
            }
            const long rlastb = rbeg + (nb - 1) * 2 * U;   // first row of the last full batch
            {
                const float* X0 = X + r * ldx; const TY* Y0 = Y + r * ldy;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    load_vec_raw<KT>(a0[u], X0 + xo[u]);
                    if constexpr (NTY) q0[u].load_nt(Y0 + yo[u]); else q0[u].load(Y0 + yo[u]);
                }
            }
            long b = 0;
            for (; b + 2 <= nb; b += 2) {
                const long r1 = r + 2 * U;
                long r2 = r + 4 * U;
                r2 = r2 < rlastb ? r2 : rlastb;              // prefetch past the end re-reads the last batch (unused)
                const float* X1 = X + r1 * ldx; const TY* Y1 = Y + r1 * ldy;
                const float* X2 = X + r2 * ldx; const TY* Y2 = Y + r2 * ldy;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    load_vec_raw<KT>(a1[u], X1 + xo[u]);
                    if constexpr (NTY) q1[u].load_nt(Y1 + yo[u]); else q1[u].load(Y1 + yo[u]);
                    __builtin_amdgcn_sched_barrier(0);
                    float bb[NT];
                    q0[u].get(bb);
#pragma unroll
                    for (int ke = 0; ke < KT; ++ke)
#pragma unroll
                        for (int ne = 0; ne < NT; ++ne) acc[ke][ne] = MFMA32(a0[u][ke], bb[ne], acc[ke][ne]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    load_vec_raw<KT>(a0[u], X2 + xo[u]);
                    if constexpr (NTY) q0[u].load_nt(Y2 + yo[u]); else q0[u].load(Y2 + yo[u]);
                    __builtin_amdgcn_sched_barrier(0);
                    float bb[NT];
                    q1[u].get(bb);
#pragma unroll
                    for (int ke = 0; ke < KT; ++ke)
#pragma unroll
                        for (int ne = 0; ne < NT; ++ne) acc[ke][ne] = MFMA32(a1[u][ke], bb[ne], acc[ke][ne]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                r += 4 * U;
            }
            if (b < nb) {
#pragma unroll
                for (int u = 0; u < U; ++u) q0[u].get(b0[u]);
                tn_comp<KT, NT, U>(acc, a0, b0);
                r += 2 * U;
            }
        }
    }
    // ragged tail of the FAST path and the whole generic path: predicated loads, zero fill
    for (; r < rend; r += 2 * U) {
        tn_load<KT, NT, FAST, U, TY>(a0, b0, X, ldx, xcols, Y, ldy, ycols, col0, r, rend, li, h);
        tn_comp<KT, NT, U>(acc, a0, b0);
    }
}

template <int KT, int NT, bool FAST, int MODE, bool NTY = false, typename TY = float>
__global__ __launch_bounds__(256, 2) void tn_kernel(TnArgs p) {
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    // wave-uniform quantities kept provably scalar (readfirstlane) so row bases live in SGPRs
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long gw = (long)blockIdx.x * (blockDim.x >> 6) + wid;
    const long chunk = gw / p.ncolblk;
    const long colblk = gw % p.ncolblk;
    if (chunk >= p.nchunks) return;
    const long col0 = colblk * 32 * NT;
    const long rbeg = chunk * p.rows_per_chunk;
    long rend = rbeg + p.rows_per_chunk;
    if (rend > p.nrows) rend = p.nrows;

    f32x16 acc[KT][NT];
#pragma unroll
    for (int ke = 0; ke < KT; ++ke)
#pragma unroll
        for (int ne = 0; ne < NT; ++ne)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ke][ne][r] = 0.f;

    tn_mainloop<KT, NT, FAST, NTY, TY>(acc, p.X, p.ldx, p.xcols, static_cast<const TY*>(p.Y), p.ldy, p.ycols, col0, rbeg, rend, li, h);

    if constexpr (MODE == TN_PARTIAL) {
        float* Pc = p.P + chunk * p.chunk_stride;
#pragma unroll
        for (int ke = 0; ke < KT; ++ke)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = KT * crow(r, h) + ke;
                float d[NT];
#pragma unroll
                for (int ne = 0; ne < NT; ++ne) d[ne] = acc[ke][ne][r];
                store_vec<NT, true>(d, Pc + (long)j * p.ldp, col0 + (long)NT * li, p.ldp, true);
            }
    }
}

// out[y][j][c] = sum_{s in slice y} P[s][j][c], j < rows, c < cols.  256 threads = 64 consecutive float4 outputs x 4
// split lanes; lane g sums splits g, g+4, ... of its slice in order, the four lane sums are combined in fixed order
// through LDS -> bitwise deterministic.  Everything else inside [rows_out x cols_out] is written as 0 (zero padding
// of the gram buffers).  gridDim.y > 1 = first stage of a two-stage reduction (out = scratch, y_stride apart).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ P, long stride, long ldp,
                                                              int nsplit, int splits_per_y, float* __restrict__ out,
                                                              long ldo, long y_stride, int rows, long cols,
                                                              int rows_out, long cols_out) {
    __shared__ f32x4 red[256];
    const long c4 = cdiv(cols_out, 4);
    const long total = (long)rows_out * c4;
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long idx = (long)blockIdx.x * 64 + o;
    const int s0 = blockIdx.y * splits_per_y;
    const int s1 = min(nsplit, s0 + splits_per_y);
    const int j = idx / c4;
    const long c = (idx % c4) * 4;
    const bool live = idx < total && j < rows && c < cols;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        const float* src = P + (long)j * ldp + c;
#pragma unroll 4
        for (int k = s0 + g; k < s1; k += 4) s += *reinterpret_cast<const f32x4*>(src + k * stride);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && idx < total) {
        s = ((red[o] + red[64 + o]) + red[128 + o]) + red[192 + o];
        float* dst = out + (long)blockIdx.y * y_stride + (long)j * ldo;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < cols_out) dst[c + e] = (live && c + e < cols) ? s[e] : 0.f;
    }
}

// =============================================================================================== small kernels
__global__ __launch_bounds__(256) void clamp_kernel(float* X, long rows, long cols, long ldx, float eps) {
    const long total = rows * cols;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / cols, c = idx % cols;
        float* p = X + r * ldx + c;
        *p = fmaxf(*p, eps);
    }
}

// W[i][j] = W[i][j] / (s[j] + eps)   |   H[j][c] = H[j][c] * s[j]
template <int OP>
__global__ __launch_bounds__(256) void scale_kernel(float* X, long rows, long cols, long ldx, const float* s, float eps) {
    const long total = rows * cols;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / cols, c = idx % cols;
        float* p = X + r * ldx + c;
        if (OP == 0) *p = *p / (s[c] + eps);
        else *p = *p * s[r];
    }
}

// KL eltwise: X[r][c] *= S[r][c] / (x[BYROW ? r : c] + eps)   (dist_nmf.py:828-830, 847-849)
template <bool BYROW>
__global__ __launch_bounds__(256) void kl_update_kernel(float* X, long rows, long cols, long ldx, const float* S,
                                                        long lds_, const float* x, float eps, int clamp) {
    const long total = rows * cols;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / cols, c = idx % cols;
        const float q = S[r * lds_ + c] / (x[BYROW ? r : c] + eps);
        float v = X[r * ldx + c] * q;
        if (clamp) v = fmaxf(v, eps);
        X[r * ldx + c] = v;
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ void block_atomic_sum(double v, double* out) {
    __shared__ double red[16];
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
        atomicAdd(out, s);
    }
}

// sum of squares of an m x n matrix; fp32 products, fp64 accumulation
template <bool FAST, typename TA = float>
__global__ __launch_bounds__(256) void sqnorm_kernel(const TA* __restrict__ A, long m, long n, long lda, double* out) {
    double acc = 0.0;
    if constexpr (FAST) {
        const long n4 = n / 4, total = m * n4;
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
            const long r = idx / n4, c = (idx % n4) * 4;
            float v[4];
            load_vec_raw<4>(v, A + r * lda + c);
            acc += (double)(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
        }
    } else {
        const long total = m * n;
        for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
            float v[1];
            load_vec_raw<1>(v, A + (idx / n) * lda + idx % n);
            acc += (double)(v[0] * v[0]);
        }
    }
    block_atomic_sum(acc, out);
}

// x[j] = sum_c H[j][c]  -- one workgroup per row
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ H, long n, long ldh, float* x) {
    const float* row = H + (long)blockIdx.x * ldh;
    double acc = 0.0;
    for (long c = threadIdx.x; c < n; c += blockDim.x) acc += (double)row[c];
    __shared__ double red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) x[blockIdx.x] = (float)(red[0] + red[1] + red[2] + red[3]);
}

// stage 1 of x[j] = sum_i W[i][j]: partial[blk][j] over a slab of rows (coalesced along j)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ W, long m, int k, long ldw,
                                                             long rows_per_blk, float* partial, int kp) {
    __shared__ float red[256];
    const int j = threadIdx.x % kp, g = threadIdx.x / kp, ng = 256 / kp;
    const long r0 = (long)blockIdx.x * rows_per_blk;
    long r1 = r0 + rows_per_blk;
    if (r1 > m) r1 = m;
    float acc = 0.f;
    if (j < k)
        for (long r = r0 + g; r < r1; r += ng) acc += W[r * ldw + j];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (g == 0) {
        for (int q = 1; q < ng; ++q) acc += red[q * kp + j];
        partial[(long)blockIdx.x * kp + j] = acc;
    }
}

__global__ void colsum_final_kernel(const float* partial, int nblk, int kp, int k, float* x) {
    const int j = threadIdx.x;
    if (j >= k) return;
    double acc = 0.0;
    for (int b = 0; b < nblk; ++b) acc += (double)partial[(long)b * kp + j];
    x[j] = (float)acc;
}

// =============================================================================================== MU update kernels
// The Frobenius multiplicative updates are HBM-bound element-wise passes with a k x k product inside:
//   H[j][c] *= S[j][c] / ((G H)[j][c] + eps)      (dist_nmf.py:750-751, G = W^T W, S = W^T A)
//   W[i][j] *= S[i][j] / ((W G)[i][j] + eps)      (dist_nmf.py:731-732, G = H H^T, S = A H^T)
// Both kernels load their whole tile of the factor AND of S up front (maximum memory-level parallelism), use the
// factor registers directly as the MFMA B operand, and -- by choosing which two contraction indices each MFMA pairs --
// make the register that fed step t the very value the epilogue needs at accumulator position t, so the factor is read
// from memory exactly once and nothing goes through LDS.  Algorithmic traffic: 12 bytes per factor element.

// H update: wave tile = KP rows x 32*NT columns, KT*NT == 4.  acc[ke][ne] (reg r, lane (li,h)) = (G H)[j][c] with
// j = KT*crow(r,h) + ke, c = col0 + NT*li + ne.  Step (r, ke) contracts the row pair jj(h) = KT*crow(r,h) + ke:
// B operand = hreg[r][ke][ne] = H[jj(h)][c] (exactly the epilogue's H value), A operand lane (li,h) = G[jj(h)][KT*li + ke'].
template <int KT, int NT, bool FAST, bool INTERIOR>
__device__ __forceinline__ void update_h_tile(float* __restrict__ H, int k, long n, long ldh, const float* __restrict__ Sm,
                                              long lds_, const float* gs, float eps, int clamp, long c, int li, int h) {
    constexpr int KP = 32 * KT;
    float hreg[16][KT][NT], sreg[16][KT][NT];
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int ke = 0; ke < KT; ++ke) {
            const int j = KT * crow(r, h) + ke;
            load_tile_vec<NT, FAST, INTERIOR>(hreg[r][ke], H + (long)j * ldh, c, n, j < k);
            load_tile_vec<NT, FAST, INTERIOR>(sreg[r][ke], Sm + (long)j * lds_, c, n, j < k);
        }
    f32x16 acc[KT][NT];
#pragma unroll
    for (int ke = 0; ke < KT; ++ke)
#pragma unroll
        for (int ne = 0; ne < NT; ++ne)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ke][ne][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int ke = 0; ke < KT; ++ke) {
            const int jj = KT * crow(r, h) + ke;           // this lane half's contraction row (jj < KP; G is zero padded)
            float a[KT];
            load_vec_raw<KT>(a, &gs[jj * KP + KT * li]);       // 32 lanes x KT floats contiguous: conflict free
#pragma unroll
            for (int k2 = 0; k2 < KT; ++k2)
#pragma unroll
                for (int ne = 0; ne < NT; ++ne) acc[k2][ne] = MFMA32(a[k2], hreg[r][ke][ne], acc[k2][ne]);
        }
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int ke = 0; ke < KT; ++ke) {
            const int j = KT * crow(r, h) + ke;
#pragma unroll
            for (int ne = 0; ne < NT; ++ne) {
                const float q = sreg[r][ke][ne] / (acc[ke][ne][r] + eps);
                float v = hreg[r][ke][ne] * q;
                if (clamp) v = fmaxf(v, eps);
                hreg[r][ke][ne] = v;
            }
            store_tile_vec<NT, FAST, INTERIOR>(hreg[r][ke], H + (long)j * ldh, c, n, j < k);
        }
}

template <int KT, int NT, bool FAST>
__global__ __launch_bounds__(256, 2) void update_h_kernel(float* __restrict__ H, int k, long n, long ldh,
                                                          const float* __restrict__ Sm, long lds_,
                                                          const float* __restrict__ G, float eps, int clamp) {
    constexpr int KP = 32 * KT;
    extern __shared__ __attribute__((aligned(16))) float gs[];   // G staged once per workgroup: rows jj, KP floats each
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long col0 = ((long)blockIdx.x * 4 + wid) * 32 * NT;
    const long c = col0 + (long)NT * li;
    for (int idx = threadIdx.x; idx < KP * KP / 4; idx += 256)
        *reinterpret_cast<f32x4*>(&gs[idx * 4]) = *reinterpret_cast<const f32x4*>(G + idx * 4);
    __syncthreads();
    if (col0 >= n) return;
    if (FAST && k == KP && col0 + 32 * NT <= n) update_h_tile<KT, NT, FAST, true>(H, k, n, ldh, Sm, lds_, gs, eps, clamp, c, li, h);
    else update_h_tile<KT, NT, FAST, false>(H, k, n, ldh, Sm, lds_, gs, eps, clamp, c, li, h);
}

// W update: wave tile = 32 rows x KP columns.  Lane (li,h) owns row i = row0 + li and keeps W[i][8s + 4h + e] in
// wreg[s][e] (a contiguous 32-row block of W is read with 16-B pieces).  out[jt] (reg r = 4g + e, lane (li,h)) =
// (W G)[i][j], j = 32 jt + 8g + 4h + e = exactly the index of wreg[4 jt + g][e]; B operand of step (s, e) = wreg[s][e],
// A operand lane (li,h) = G[32 jt + li][8s + 4h + e] (G symmetric).
template <int KT, bool FAST, bool INTERIOR>
__device__ __forceinline__ void update_w_tile(float* __restrict__ W, int k, long ldw, const float* __restrict__ Sm,
                                              long lds_, const float* gs, float eps, long row, bool rok, int li, int h) {
    constexpr int KP = 32 * KT, GP = KP + 4;
    float wreg[4 * KT][4], sreg[4 * KT][4];
#pragma unroll
    for (int s = 0; s < 4 * KT; ++s) {
        load_tile_vec<4, FAST, INTERIOR>(wreg[s], W + row * ldw, 8 * s + 4 * h, k, rok);
        load_tile_vec<4, FAST, INTERIOR>(sreg[s], Sm + row * lds_, 8 * s + 4 * h, k, rok);
    }
    f32x16 out[KT];
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[jt][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4 * KT; ++s)
#pragma unroll
        for (int jt = 0; jt < KT; ++jt) {
            float a[4];
            load_vec_raw<4>(a, &gs[(jt * 32 + li) * GP + 8 * s + 4 * h]);
#pragma unroll
            for (int e = 0; e < 4; ++e) out[jt] = MFMA32(a[e], wreg[s][e], out[jt]);
        }
#pragma unroll
    for (int jt = 0; jt < KT; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int s = 4 * jt + g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float q = sreg[s][e] / (out[jt][4 * g + e] + eps);
                wreg[s][e] = wreg[s][e] * q;
            }
            store_tile_vec<4, FAST, INTERIOR>(wreg[s], W + row * ldw, 8 * s + 4 * h, k, rok);
        }
}

template <int KT, bool FAST>
__global__ __launch_bounds__(256, 2) void update_w_kernel(float* __restrict__ W, long m, int k, long ldw,
                                                          const float* __restrict__ Sm, long lds_,
                                                          const float* __restrict__ G, float eps) {
    constexpr int KP = 32 * KT, GP = KP + 4;   // LDS row pitch: +16 B so that rows 0..15 land on distinct 16-B slots
    extern __shared__ __attribute__((aligned(16))) float gs[];
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long row0 = ((long)blockIdx.x * 4 + wid) * 32;
    const long row = row0 + li;
    for (int idx = threadIdx.x; idx < KP * KP / 4; idx += 256) {
        const int gr = idx / (KP / 4), gc = (idx % (KP / 4)) * 4;
        *reinterpret_cast<f32x4*>(&gs[gr * GP + gc]) = *reinterpret_cast<const f32x4*>(G + gr * KP + gc);
    }
    __syncthreads();
    if (row0 >= m) return;
    if (FAST && k == KP && row0 + 32 <= m) update_w_tile<KT, FAST, true>(W, k, ldw, Sm, lds_, gs, eps, row, true, li, h);
    else update_w_tile<KT, FAST, false>(W, k, ldw, Sm, lds_, gs, eps, row, row < m, li, h);
}

// =============================================================================================== HALS sweeps
// Frobenius HALS (dist_nmf.py:873-934, :411-470) reuses the MU contractions (A H^T, H H^T, W^T A, W^T W) and replaces
// the multiply-divide by a column-sequential sweep.
//
// W sweep, one launch per column kk (the global 2-norm of column kk must be known before column kk+1 is touched;
// with p_r > 1 the host allreduces the 8-byte sum of squares between launches, exactly where the reference calls
// utils.norm, dist_nmf.py:889):
//   first the pending normalisation of column kk-1 is applied (W[i][kk-1] /= ss, ss = sqrt(*prev_ss2), skipped when 0),
//   t = W[i][kk] * G[kk][kk] + AH[i][kk] - sum_j W[i][j] G[j][kk];  W[i][kk] = max(t, eps);  *ss2_out += W[i][kk]^2
// One lane per row; a row of W is k contiguous floats.
__global__ __launch_bounds__(256) void hals_w_col_kernel(float* __restrict__ W, long m, int k, long ldw,
                                                        const float* __restrict__ AH, long ldah,
                                                        const float* __restrict__ G, int kp, int kk,
                                                        const double* __restrict__ prev_ss2, float eps,
                                                        double* __restrict__ ss2_out) {
    __shared__ float gcol[DNMF_MAX_K];
    for (int j = threadIdx.x; j < k; j += blockDim.x) gcol[j] = G[(long)j * kp + kk];
    __syncthreads();
    float inv_den = 0.f;   // ss of the previous column (0 = no pending normalisation)
    if (kk > 0 && prev_ss2) inv_den = (float)sqrt(*prev_ss2);
    double sq = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long)gridDim.x * blockDim.x) {
        float* row = W + i * ldw;
        if (kk > 0 && inv_den > 0.f) row[kk - 1] = row[kk - 1] / inv_den;
        float dot = 0.f;
        for (int j = 0; j < k; ++j) dot = fmaf(row[j], gcol[j], dot);
        const float t = row[kk] * gcol[kk] + AH[i * ldah + kk] - dot;
        const float w = fmaxf(t, eps);
        row[kk] = w;
        sq += (double)w * (double)w;
    }
    block_atomic_sum(sq, ss2_out);
}

// final normalisation of one column: W[i][col] /= sqrt(*ss2) (skipped when 0)
__global__ __launch_bounds__(256) void hals_w_scale_kernel(float* __restrict__ W, long m, long ldw, int col,
                                                          const double* __restrict__ ss2) {
    const float ss = (float)sqrt(*ss2);
    if (!(ss > 0.f)) return;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long)gridDim.x * blockDim.x)
        W[i * ldw + col] = W[i * ldw + col] / ss;
}

// H sweep: columns are independent, rows are sequential (row kk uses the already updated rows < kk):
//   H[kk][c] = max(H[kk][c] + AtW[kk][c] - sum_j G[kk][j] H[j][c], eps)            (dist_nmf.py:905-909)
// One lane per column with the whole column of H in registers; G (= W^T W, zero padded) is broadcast from LDS.
template <int KP>
__global__ __launch_bounds__(256) void hals_h_kernel(float* __restrict__ H, int k, long n, long ldh,
                                                    const float* __restrict__ AtW, long ldatw,
                                                    const float* __restrict__ G, float eps) {
    extern __shared__ __attribute__((aligned(16))) float gs[];
    for (int idx = threadIdx.x; idx < KP * KP / 4; idx += 256)
        *reinterpret_cast<f32x4*>(&gs[idx * 4]) = *reinterpret_cast<const f32x4*>(G + idx * 4);
    __syncthreads();
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    float hc[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) hc[j] = j < k ? H[(long)j * ldh + c] : 0.f;
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
        if (kk < k) {
            float dot = 0.f;
#pragma unroll
            for (int j4 = 0; j4 < KP; j4 += 4) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(&gs[kk * KP + j4]);
                dot = fmaf(g[0], hc[j4], dot);
                dot = fmaf(g[1], hc[j4 + 1], dot);
                dot = fmaf(g[2], hc[j4 + 2], dot);
                dot = fmaf(g[3], hc[j4 + 3], dot);
            }
            const float t = hc[kk] + AtW[(long)kk * ldatw + c] - dot;
            hc[kk] = fmaxf(t, eps);
        }
    }
#pragma unroll
    for (int j = 0; j < KP; ++j)
        if (j < k) H[(long)j * ldh + c] = hc[j];
}

// Same sweep for KP = 128 with the column state in LDS instead of 128 registers per lane (runtime loops, 64 lanes per
// workgroup: hs[j][lane], G rows broadcast from LDS).
__global__ __launch_bounds__(64) void hals_h_kernel_lds(float* __restrict__ H, int k, long n, long ldh,
                                                       const float* __restrict__ AtW, long ldatw,
                                                       const float* __restrict__ G, int kp, float eps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* gs = sm;                 // kp * kp
    float* hs = sm + kp * kp;       // kp * 64
    for (int idx = threadIdx.x; idx < kp * kp; idx += 64) gs[idx] = G[idx];
    const long c = (long)blockIdx.x * 64 + threadIdx.x;
    const bool live = c < n;
    for (int j = 0; j < k; ++j) hs[j * 64 + threadIdx.x] = live ? H[(long)j * ldh + c] : 0.f;
    __syncthreads();
    if (!live) return;
    for (int kk = 0; kk < k; ++kk) {
        float dot = 0.f;
        for (int j = 0; j < k; ++j) dot = fmaf(gs[kk * kp + j], hs[j * 64 + threadIdx.x], dot);
        const float t = hs[kk * 64 + threadIdx.x] + AtW[(long)kk * ldatw + c] - dot;
        hs[kk * 64 + threadIdx.x] = fmaxf(t, eps);
    }
    for (int j = 0; j < k; ++j) H[(long)j * ldh + c] = hs[j * 64 + threadIdx.x];
}

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
};

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
__global__ __launch_bounds__(256) void resid_kernel(NnArgs p) {
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
        for (int r = 0; r < 16; ++r) acc[ne][r] = 0.f;
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
        for (int ne = 0; ne < NT; ++ne) acc[ne][r] = areg[r][ne] / (acc[ne][r] + p.eps);  // U (dist_nmf.py:806)
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

template <int KT, int NT, bool FAST>
__global__ __launch_bounds__(256, KT == 2 ? 2 : 1) void kl_wtu_kernel(NnArgs p, long rowblks_per_chunk) {
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
    // Measured: the branch-free (INTERIOR) form of this block is 8-20 % SLOWER here (its 16 A loads then issue as one
    // VMEM block ahead of the MFMAs, cf. tn_mainloop); the predicated loads spread out.  Kept predicated until the
    // block is software pipelined across row blocks.
    for (long rb = chunk * rowblks_per_chunk; rb < rb1; ++rb)
        kl_wtu_block<KT, NT, FAST, false>(out, p, smem, rb * 32, col0, li, h);
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
    const long arow = (long)blockIdx.x * 128 + wave * 32 + li;
    const bool rok = arow < p.m;
    const long cbeg = (long)blockIdx.y * cols_per_split;
    long cend = cbeg + cols_per_split;
    if (cend > p.n) cend = p.n;
    const long nt = (cend - cbeg + BK - 1) / BK;

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
        if (hrows_in && cbeg + BK <= cend) stage_load<KP, T, FAST, true>(hst, p.H, p.ldh, p.k, cend, 0, cbeg, tid);
        else stage_load<KP, T, FAST, false>(hst, p.H, p.ldh, p.k, cend, 0, cbeg, tid);
        stage_store<KP, T>(smem, hst, tid);
        if (INTERIOR && cbeg + BK <= cend) {
#pragma unroll
            for (int g = 0; g < 4; ++g) load_tile_vec<4, FAST, INTERIOR>(a_cur[g], p.A + arow * p.lda, cbeg + 8 * g + 4 * h, cend, rok);
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) load_vec<4, FAST>(a_cur[g], p.A + arow * p.lda, cbeg + 8 * g + 4 * h, cend, rok);
        }
    }
    __syncthreads();
    for (long t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < nt;
        const long c1 = cbeg + (t + 1) * BK;
        const float* Hs = smem + cur * STAGE;
        f32x16 st;  // S^T tile: rows c, lanes i
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4 * KT; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int jj = 8 * s + 4 * h + e;
                const float hv = Hs[lds_idx(jj, li >> 2) + (li & 3)];
                st = MFMA32(hv, wreg[s][e], st);
            }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) st[4 * g + e] = a_cur[g][e] / (st[4 * g + e] + p.eps);   // U^T (dist_nmf.py:806)
        // the A registers are free now: fetch the next tile's pieces (and the next H tile) under the second product
        if (more) {
            if (hrows_in && c1 + BK <= cend) stage_load<KP, T, FAST, true>(hst, p.H, p.ldh, p.k, cend, 0, c1, tid);
            else stage_load<KP, T, FAST, false>(hst, p.H, p.ldh, p.k, cend, 0, c1, tid);
            if (INTERIOR && c1 + BK <= cend) {
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
                                                        long split_stride, long cols_per_split, int out_cols) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // block-uniform: all 128 rows in bounds and no rank padding -> branch-free W / A / output accesses
    const bool interior = FAST && p.k == 32 * KT && ((long)blockIdx.x + 1) * 128 <= p.m && out_cols >= 32 * KT;
    if (interior) kl_uht_body<KT, FAST, true>(p, out_base, ldo, split_stride, cols_per_split, out_cols, smem);
    else kl_uht_body<KT, FAST, false>(p, out_base, ldo, split_stride, cols_per_split, out_cols, smem);
}

// =============================================================================================== host side
int kt_of(int k) {
    if (k < 1 || k > DNMF_MAX_K) return -1;
    return k <= 32 ? 1 : (k <= 64 ? 2 : 4);
}

hipStream_t S(void* s) {
    clear_hip_error();
    return reinterpret_cast<hipStream_t>(s);
}

template <typename K>
void allow_lds(K kernel, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int KT, int MT, int NW, int KS, bool FAST, int MODE, int PF, typename TX>
int launch_nt_pf(const NtArgs& a, int nsplit, hipStream_t st) {
    constexpr int BM = 32 * MT * (NW / KS);
    constexpr size_t lds32 = 2ul * (BM + 32 * KT) * BK * sizeof(float);                     // fp32 tiles (also the W.G loop)
    constexpr size_t lds16 = 2ul * (BM * BK + 32 * KT * BKH) * sizeof(float);               // bf16 X: [X raw | Y 64 wide]
    constexpr size_t lds = std::is_same<TX, bf16_t>::value ? (lds16 > lds32 ? lds16 : lds32) : lds32;
    static bool once = false;
    if (!once) { allow_lds(nt_kernel<KT, MT, NW, KS, FAST, MODE, PF, TX>, lds); once = true; }
    hipLaunchKernelGGL((nt_kernel<KT, MT, NW, KS, FAST, MODE, PF, TX>), dim3((unsigned)cdiv(a.nrows, BM), (unsigned)nsplit),
                       dim3(64 * NW), lds, st, a);
    return check_launch("nt_kernel");
}

template <int KT, int MT, int NW, int KS, bool FAST, int MODE, typename TX>
int launch_nt_inst(const NtArgs& a, int nsplit, hipStream_t st) {
    // PF code (switch DNMF_NT_PF for A/B runs) -- what the interior tiles of the streamed operand do:
    //   1  loads of k-tile t+1 issued at the top of tile t, tiles in order (also the generic / edge path)
    //   5  1 + every workgroup starts at a different k-tile (rotated order) + nontemporal loads of A
    //   7  5 with LDS-DMA staging (global_load_lds: no staging VGPRs, no ds_write)
    //  10  5 with TWO k-tiles in flight and a branch-free loop (nt_mainloop_p2) -- the default
    // Measured on MI355X (tools/kbench.py, k = 64, n = 8192; ms at 262144 / 65536 / 32768 rows):
    //   1: 2.54 / 0.82 / -     5: 2.42-2.49 / 0.76 / 0.44     7: +-2 % of 5     10: 2.43-2.47 / 0.76 / 0.40
    // HBM reads per launch at 262144 rows (PMC): 8.67 GB for 1, 10.99 GB with the rotation alone (the streamed A evicts
    // H from L2), 8.47 GB with the nontemporal hint on top.  What did NOT help (measured, not kept): a prefetch
    // distance of 2 with conditional loads (hipcc then drains vmcnt(0) at every join: 0-18 % slower), fragment reads
    // one group ahead in a second register set (-0..4 %), 8-wave workgroups with two contraction slices (+-3 %),
    // 64-row tiles, capping workgroups per CU through the LDS request (+-5 %).  With A cache-resident the same
    // kernel reaches 103 / 118 / 128 TFLOP/s at 32768 / 65536 / 262144 rows against 82-87 / 91 / 113 from HBM, while
    // the TN form loses only 2-8 %: the remaining gap is the memory system under this access shape (128 B per row
    // per visit, 128 rows apart by the 32 KiB row pitch), not the instruction schedule.
    static const int pf = getenv("DNMF_NT_PF") ? atoi(getenv("DNMF_NT_PF")) : 10;
    if constexpr (FAST && KS == 1) if (MODE == NT_FUSED_W || !a.store_all) {
        if (pf == 5) return launch_nt_pf<KT, MT, NW, KS, FAST, MODE, 5, TX>(a, nsplit, st);
        if constexpr (std::is_same<TX, float>::value) {
            if (pf == 7) return launch_nt_pf<KT, MT, NW, KS, FAST, MODE, 7, TX>(a, nsplit, st);
            if (pf == 10) return launch_nt_pf<KT, MT, NW, KS, FAST, MODE, 10, TX>(a, nsplit, st);
        } else {
            if (pf == 10) return launch_nt_pf<KT, MT, NW, KS, FAST, MODE, 5, TX>(a, nsplit, st);   // bf16 X: one 64-wide tile in flight
        }
    }
    return launch_nt_pf<KT, MT, NW, KS, FAST, MODE, 1, TX>(a, nsplit, st);
}

// Tile configuration per padded rank (KT = KP/32).  The kernel template also supports 8-wave workgroups with two contraction slices per row group (KS = 2: same
// tile, twice the waves per SIMD); measured within +-3 % of the 4-wave form at every shard size and not instantiated.
template <int MODE, typename TX = float>
int launch_nt(int kt, bool fast, const NtArgs& a, int nsplit, hipStream_t st) {
#define NT_CASE(KT_, MT_, NW_, KS_)                                                               \
    return fast ? launch_nt_inst<KT_, MT_, NW_, KS_, true, MODE, TX>(a, nsplit, st)                \
                : launch_nt_inst<KT_, MT_, NW_, KS_, false, MODE, TX>(a, nsplit, st);
    // 128-row tiles for every rank.  (k <= 32 used 256-row tiles, MT = 2, in an earlier version: equal at 262144 rows,
    // 1.8x slower at 32768 rows where it left half the CUs without a workgroup.)
    if (kt == 1) { NT_CASE(1, 1, 4, 1) }
    if (kt == 2) { NT_CASE(2, 1, 4, 1) }
    if (kt == 4) { NT_CASE(4, 1, 4, 1) }
#undef NT_CASE
    return fail(DNMF_EINVAL, "unsupported k tile %d", kt);
}

inline int nt_rows_per_tile(int) { return 128; }
inline int tn_nt(int kt) { return kt == 4 ? 2 : 4; }  // column sets per wave in TN form
inline int kl_nt(int kt) { return kt == 1 ? 4 : 2; }  // column sets per wave in kl_wtu (three live tiles: out, S/U, A)

template <int MODE, typename TY = float>
int launch_tn(int kt, bool fast, const TnArgs& a, hipStream_t st) {
    const long waves = (long)a.nchunks * a.ncolblk;
    const dim3 grid((unsigned)cdiv(waves, 4)), block(256);
    // nontemporal loads of the streamed operand (A): +2 % at 262144 rows, +7 % at 65536 rows, slightly less HBM
    // traffic (the reused W rows stay in L2).  DNMF_TN_NT=0 switches them off for A/B runs.
    static const bool nty = !(getenv("DNMF_TN_NT") && atoi(getenv("DNMF_TN_NT")) == 0);
#define TN_CASE(KT_, NT_)                                                                                \
    if (kt == KT_) {                                                                                     \
        if (fast && nty) hipLaunchKernelGGL((tn_kernel<KT_, NT_, true, MODE, true, TY>), grid, block, 0, st, a); \
        else if (fast) hipLaunchKernelGGL((tn_kernel<KT_, NT_, true, MODE, false, TY>), grid, block, 0, st, a);  \
        else hipLaunchKernelGGL((tn_kernel<KT_, NT_, false, MODE, false, TY>), grid, block, 0, st, a);          \
        return check_launch("tn_kernel");                                                                \
    }
    TN_CASE(1, 4)
    TN_CASE(2, 4)
    TN_CASE(4, 2)
#undef TN_CASE
    return fail(DNMF_EINVAL, "unsupported k tile %d", kt);
}

// two-stage when there are many partials per output (gram of a tall W): slices of 32 splits, then one more pass
constexpr int REDUCE_SLICE = 32;
inline int reduce_slices(int nsplit) { return nsplit > 2 * REDUCE_SLICE ? (int)cdiv(nsplit, REDUCE_SLICE) : 1; }
inline size_t reduce_scratch_bytes(int nsplit, int rows_out, long cols_out) {
    const int y = reduce_slices(nsplit);
    return y > 1 ? (size_t)y * rows_out * round_up(cols_out, 4) * sizeof(float) : 0;
}

// `scratch` must hold reduce_scratch_bytes(nsplit, rows_out, cols_out)
int launch_reduce(const float* P, long stride, long ldp, int nsplit, float* out, long ldo, int rows, long cols,
                  int rows_out, long cols_out, float* scratch, hipStream_t st) {
    const long total = (long)rows_out * cdiv(cols_out, 4);
    const unsigned gx = (unsigned)cdiv(total, 64);
    const int ny = reduce_slices(nsplit);
    if (ny == 1) {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, 1), dim3(256), 0, st, P, stride, ldp, nsplit, nsplit, out, ldo,
                           0L, rows, cols, rows_out, cols_out);
    } else {
        const long ld2 = round_up(cols_out, 4), ys = (long)rows_out * ld2;
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, ny), dim3(256), 0, st, P, stride, ldp, nsplit, REDUCE_SLICE,
                           scratch, ld2, ys, rows, cols, rows_out, cols_out);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, 1), dim3(256), 0, st, (const float*)scratch, ys, ld2, ny, ny, out,
                           ldo, 0L, rows_out, cols_out, rows_out, cols_out);
    }
    return check_launch("reduce_partials");
}

// ---- chunking heuristics (shared by the ws-size query and the launches)
struct TnPlan { int ncolblk; int nchunks; long rows_per_chunk; long ldp; long chunk_stride; };

TnPlan plan_tn(long nrows, long ycols, int kt, int nt) {
    TnPlan p;
    p.ncolblk = (int)cdiv(ycols, 32 * nt);
    static const long target_waves = getenv("DNMF_TN_WAVES") ? atol(getenv("DNMF_TN_WAVES")) : 2048;  // one resident round: 256 CUs x 2 waves/SIMD (tunable for experiments)
    long nchunks = std::max<long>(1, target_waves / p.ncolblk);
    nchunks = std::min<long>(nchunks, std::max<long>(1, cdiv(nrows, 256)));
    p.rows_per_chunk = round_up(cdiv(nrows, nchunks), 16);
    p.nchunks = (int)cdiv(nrows, p.rows_per_chunk);
    p.ldp = (long)p.ncolblk * 32 * nt;
    p.chunk_stride = p.ldp * 32 * kt;
    return p;
}

struct SplitPlan { int nsplit; long cols_per_split; };

SplitPlan plan_gram_nt(long n) {
    SplitPlan s;
    s.cols_per_split = 256;
    if (n > 256 * 512) s.cols_per_split = round_up(cdiv(n, 512), BK);
    s.nsplit = (int)std::max<long>(1, cdiv(n, s.cols_per_split));
    return s;
}

size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

struct WsLayout {
    size_t g_off, s_off, x_off, part_off, total;  // G [KP*KP] | S = AtW / AH / UHT / WTU | x [KP] | partials
};

size_t partial_bytes(long m, long n, int k) {
    const int kt = kt_of(k), kp = 32 * kt;
    size_t b = 0;
    {   // wta / kl_wtu: A [m x n]
        TnPlan p = plan_tn(m, n, kt, tn_nt(kt));
        b = std::max(b, (size_t)p.nchunks * p.chunk_stride * sizeof(float) + reduce_scratch_bytes(p.nchunks, k, n));
        TnPlan q = plan_tn(m, n, kt, kl_nt(kt));
        const long klc = cdiv(cdiv(m, 32), std::max<long>(1, q.rows_per_chunk / 32));
        b = std::max(b, (size_t)klc * q.ldp * kp * sizeof(float) + reduce_scratch_bytes((int)klc, k, n));
    }
    {   // gram W^T W: Y = W [m x k]
        TnPlan p = plan_tn(m, kp, kt, kt == 4 ? 2 : kt);
        b = std::max(b, (size_t)p.nchunks * p.chunk_stride * sizeof(float) + reduce_scratch_bytes(p.nchunks, kp, kp));
    }
    {   // gram H H^T
        SplitPlan s = plan_gram_nt(n);
        b = std::max(b, (size_t)s.nsplit * nt_rows_per_tile(kt) * kp * sizeof(float) + reduce_scratch_bytes(s.nsplit, kp, kp));
    }
    {   // kl_uht column-split slabs
        const long rowtiles = cdiv(m, 128);
        long ns = std::min<long>(std::max<long>(1, cdiv(1536, rowtiles)), std::max<long>(1, n / 256));
        const long cps = round_up(cdiv(n, ns), BK);
        const int nsp = (int)cdiv(n, cps);
        if (nsp > 1) b = std::max(b, (size_t)nsp * m * kp * sizeof(float) + reduce_scratch_bytes(nsp, (int)m, k));
    }
    b = std::max(b, (size_t)cdiv(m, 1024) * kp * sizeof(float));  // colsum partials
    return b;
}

WsLayout ws_layout(long m, long n, int k) {
    const int kp = 32 * kt_of(k);
    WsLayout w;
    w.g_off = 0;
    w.s_off = align256((size_t)kp * kp * sizeof(float));
    const size_t s_elems = std::max((size_t)k * round_up(n, 4), (size_t)m * round_up(k, 4));
    w.x_off = w.s_off + align256(s_elems * sizeof(float));
    w.part_off = w.x_off + align256((size_t)kp * sizeof(float));
    w.total = w.part_off + align256(partial_bytes(m, n, k));
    return w;
}

}  // namespace

// =============================================================================================== C ABI
extern "C" {

const char* dnmf_last_error(void) { return g_err; }
int dnmf_version(void) { return 100; }
int dnmf_kp(int k) { const int kt = kt_of(k); return kt < 0 ? -1 : 32 * kt; }

size_t dnmf_ws_bytes(long m, long n, int k) {
    if (kt_of(k) < 0 || m < 1 || n < 1) return 0;
    return ws_layout(m, n, k).total;
}

#define REQUIRE(cond, ...) \
    do { if (!(cond)) return fail(DNMF_EINVAL, __VA_ARGS__); } while (0)

int dnmf_gram_hht(const float* H, int k, long n, long ldh, float* G, void* ws, size_t ws_bytes, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && H && G && ws && n >= 1 && ldh >= n, "gram_hht: bad arguments (k=%d n=%ld)", k, n);
    const int kp = 32 * kt;
    const SplitPlan sp = plan_gram_nt(n);
    const long tile_rows = nt_rows_per_tile(kt);
    const size_t pbytes = (size_t)sp.nsplit * tile_rows * kp * sizeof(float);
    const size_t need = pbytes + reduce_scratch_bytes(sp.nsplit, kp, kp);
    if (ws_bytes < need) return fail(DNMF_EWS, "gram_hht: workspace %zu < %zu", ws_bytes, need);
    NtArgs a{};
    a.X = H; a.ldx = ldh; a.nrows = k; a.ncols = n;
    a.Y = H; a.ldy = ldh; a.yrows = k;
    a.cols_per_split = sp.cols_per_split;
    a.out = (float*)ws; a.ldo = kp; a.split_stride = tile_rows * kp; a.store_all = 1;
    const bool fast = aligned16(H) && ldh % 4 == 0 && n % 4 == 0;
    int rc = launch_nt<NT_STORE>(kt, fast, a, sp.nsplit, S(stream));
    if (rc) return rc;
    return launch_reduce((const float*)ws, tile_rows * kp, kp, sp.nsplit, G, kp, k, k, kp, kp,
                         (float*)((char*)ws + pbytes), S(stream));
}

int dnmf_gram_wtw(const float* W, long m, int k, long ldw, float* G, void* ws, size_t ws_bytes, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && W && G && ws && m >= 1 && ldw >= k, "gram_wtw: bad arguments (k=%d m=%ld)", k, m);
    const int kp = 32 * kt;
    // TN form with X = Y = W; NT column sets per wave (KT = 4 uses 2 column blocks of 64 to bound registers)
    const int nt = kt == 4 ? 2 : kt;
    TnPlan p = plan_tn(m, kp, kt, nt);
    const size_t pbytes = (size_t)p.nchunks * p.chunk_stride * sizeof(float);
    const size_t need = pbytes + reduce_scratch_bytes(p.nchunks, kp, kp);
    if (ws_bytes < need) return fail(DNMF_EWS, "gram_wtw: workspace %zu < %zu", ws_bytes, need);
    TnArgs a{};
    a.X = W; a.ldx = ldw; a.xcols = k; a.Y = W; a.ldy = ldw; a.ycols = k;
    a.nrows = m; a.rows_per_chunk = p.rows_per_chunk; a.nchunks = p.nchunks; a.ncolblk = p.ncolblk;
    a.P = (float*)ws; a.chunk_stride = p.chunk_stride; a.ldp = p.ldp;
    const bool fast = aligned16(W) && ldw % 4 == 0 && k % 4 == 0;
    const long waves = (long)a.nchunks * a.ncolblk;
    const dim3 grid((unsigned)cdiv(waves, 4)), block(256);
    hipStream_t st = S(stream);
#define GRAM_CASE(KT_, NT_)                                                                          \
    if (kt == KT_) {                                                                                 \
        if (fast) hipLaunchKernelGGL((tn_kernel<KT_, NT_, true, TN_PARTIAL>), grid, block, 0, st, a); \
        else hipLaunchKernelGGL((tn_kernel<KT_, NT_, false, TN_PARTIAL>), grid, block, 0, st, a);    \
    }
    GRAM_CASE(1, 1) GRAM_CASE(2, 2) GRAM_CASE(4, 2)
#undef GRAM_CASE
    int rc = check_launch("gram_wtw");
    if (rc) return rc;
    return launch_reduce((const float*)ws, p.chunk_stride, p.ldp, p.nchunks, G, kp, k, k, kp, kp,
                         (float*)((char*)ws + pbytes), st);
}

}  // extern "C"  (typed implementations shared by the fp32 and the bf16-A entry points)
namespace {
// alignment the vector path needs from A: 16 B for fp32 rows, 8 B for bf16 rows (4 elements per lane either way)
template <typename TA> bool a_aligned(const TA* A) { return ((uintptr_t)A % (4 * sizeof(TA))) == 0; }
// the NT form reads 16 B per lane from A whatever its type: bf16 rows need lda % 8 == 0 and a 16-byte aligned base
template <typename TA> bool a_rows16(const TA* A, long lda) { return aligned16(A) && (lda * sizeof(TA)) % 16 == 0; }

template <typename TA>
int aht_impl(const TA* A, long m, long n, long lda, const float* H, int k, long ldh, float* AH, long ldah,
             void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && H && AH && m >= 1 && n >= 1 && lda >= n && ldh >= n && ldah >= k, "aht: bad arguments");
    NtArgs a{};
    a.X = A; a.ldx = lda; a.nrows = m; a.ncols = n;
    a.Y = H; a.ldy = ldh; a.yrows = k;
    a.cols_per_split = round_up(n, BK);
    a.out = AH; a.ldo = ldah; a.split_stride = 0; a.store_all = 0;
    const bool fast = a_rows16(A, lda) && aligned16(H) && ldh % 4 == 0 && n % 4 == 0;
    return launch_nt<NT_STORE, TA>(kt, fast, a, 1, S(stream));
}

template <typename TA>
int aht_update_w_impl(const TA* A, long m, long n, long lda, const float* H, int k, long ldh, const float* G,
                      float* W, long ldw, float eps, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && H && G && W && m >= 1 && n >= 1 && (lda >= n || lda == 0) && ldh >= n && ldw >= k, "aht_update_w: bad arguments");
    NtArgs a{};
    a.X = A; a.ldx = lda; a.nrows = m; a.ncols = n;
    a.Y = H; a.ldy = ldh; a.yrows = k;
    a.cols_per_split = round_up(n, BK);
    a.W = W; a.ldw = ldw; a.G = G; a.eps = eps; a.k = k;
    const bool fast = a_rows16(A, lda) && aligned16(H) && ldh % 4 == 0 && n % 4 == 0;
    a.wfast = aligned16(W) && ldw % 4 == 0 && k % 4 == 0;
    return launch_nt<NT_FUSED_W, TA>(kt, fast, a, 1, S(stream));
}
}  // namespace
extern "C" {

int dnmf_aht(const float* A, long m, long n, long lda, const float* H, int k, long ldh, float* AH, long ldah,
             void* stream) {
    return aht_impl<float>(A, m, n, lda, H, k, ldh, AH, ldah, stream);
}
int dnmf_aht_bf16a(const void* A, long m, long n, long lda, const float* H, int k, long ldh, float* AH, long ldah,
                   void* stream) {
    return aht_impl<bf16_t>((const bf16_t*)A, m, n, lda, H, k, ldh, AH, ldah, stream);
}
int dnmf_aht_update_w(const float* A, long m, long n, long lda, const float* H, int k, long ldh, const float* G,
                      float* W, long ldw, float eps, void* stream) {
    return aht_update_w_impl<float>(A, m, n, lda, H, k, ldh, G, W, ldw, eps, stream);
}
int dnmf_aht_update_w_bf16a(const void* A, long m, long n, long lda, const float* H, int k, long ldh, const float* G,
                            float* W, long ldw, float eps, void* stream) {
    return aht_update_w_impl<bf16_t>((const bf16_t*)A, m, n, lda, H, k, ldh, G, W, ldw, eps, stream);
}

int dnmf_mu_update_w(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, float eps,
                     void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && W && AH && G && m >= 1 && ldw >= k && ldah >= k, "mu_update_w: bad arguments");
    const bool fast = aligned16(W) && aligned16(AH) && ldw % 4 == 0 && ldah % 4 == 0 && k % 4 == 0;
    const dim3 grid((unsigned)cdiv(cdiv(m, 32), 4)), block(256);
    hipStream_t st = S(stream);
#define UW_CASE(KT_)                                                                                              \
    if (kt == KT_) {                                                                                              \
        const size_t lds = (size_t)(32 * KT_) * (32 * KT_ + 4) * sizeof(float);                                    \
        static bool once = false;                                                                                 \
        if (!once) { allow_lds(update_w_kernel<KT_, true>, lds); allow_lds(update_w_kernel<KT_, false>, lds); once = true; } \
        if (fast) hipLaunchKernelGGL((update_w_kernel<KT_, true>), grid, block, lds, st, W, m, k, ldw, AH, ldah, G, eps); \
        else hipLaunchKernelGGL((update_w_kernel<KT_, false>), grid, block, lds, st, W, m, k, ldw, AH, ldah, G, eps);    \
    }
    UW_CASE(1) UW_CASE(2) UW_CASE(4)
#undef UW_CASE
    return check_launch("mu_update_w");
}

}  // extern "C"
namespace {
template <typename TA>
int wta_impl(const TA* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw,
             void* ws, size_t ws_bytes, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && AtW && ws && m >= 1 && n >= 1 && (lda >= n || lda == 0) && ldw >= k && ldatw >= n, "wta: bad arguments");
    const int nt = tn_nt(kt);
    TnPlan p = plan_tn(m, n, kt, nt);
    const size_t pbytes = (size_t)p.nchunks * p.chunk_stride * sizeof(float);
    const size_t need = pbytes + reduce_scratch_bytes(p.nchunks, k, n);
    if (ws_bytes < need) return fail(DNMF_EWS, "wta: workspace %zu < %zu", ws_bytes, need);
    TnArgs a{};
    a.X = W; a.ldx = ldw; a.xcols = k; a.Y = A; a.ldy = lda; a.ycols = n;
    a.nrows = m; a.rows_per_chunk = p.rows_per_chunk; a.nchunks = p.nchunks; a.ncolblk = p.ncolblk;
    a.P = (float*)ws; a.chunk_stride = p.chunk_stride; a.ldp = p.ldp;
    // the streamed A is read 4 elements per lane, W only KT per lane (1 float for k <= 32): W's alignment need is KT-wide
    const bool fast = a_aligned(A) && lda % 4 == 0 && n % 4 == 0 &&
                      ((uintptr_t)W % (4 * kt)) == 0 && ldw % kt == 0 && k % kt == 0;
    int rc = launch_tn<TN_PARTIAL, TA>(kt, fast, a, S(stream));
    if (rc) return rc;
    return launch_reduce((const float*)ws, p.chunk_stride, p.ldp, p.nchunks, AtW, ldatw, k, n, k, n,
                         (float*)((char*)ws + pbytes), S(stream));
}
}  // namespace
extern "C" {

int dnmf_wta(const float* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw,
             void* ws, size_t ws_bytes, void* stream) {
    return wta_impl<float>(A, m, n, lda, W, k, ldw, AtW, ldatw, ws, ws_bytes, stream);
}
int dnmf_wta_bf16a(const void* A, long m, long n, long lda, const float* W, int k, long ldw, float* AtW, long ldatw,
                   void* ws, size_t ws_bytes, void* stream) {
    return wta_impl<bf16_t>((const bf16_t*)A, m, n, lda, W, k, ldw, AtW, ldatw, ws, ws_bytes, stream);
}

int dnmf_mu_update_h(float* H, int k, long n, long ldh, const float* AtW, long ldatw, const float* G, float eps,
                     int clamp, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && H && AtW && G && n >= 1 && ldh >= n && ldatw >= n, "mu_update_h: bad arguments");
    const int nt = 4 / kt;                                  // KT * NT == 4: three 64-register tiles per wave
    const bool fast = aligned16(H) && aligned16(AtW) && ldh % 4 == 0 && ldatw % 4 == 0 && n % 4 == 0;
    const dim3 grid((unsigned)cdiv(cdiv(n, 32 * nt), 4)), block(256);
    hipStream_t st = S(stream);
#define UH_CASE(KT_, NT_)                                                                                         \
    if (kt == KT_) {                                                                                              \
        const size_t lds = (size_t)(32 * KT_) * (32 * KT_) * sizeof(float);                                        \
        if (fast) hipLaunchKernelGGL((update_h_kernel<KT_, NT_, true>), grid, block, lds, st, H, k, n, ldh, AtW, ldatw, G, eps, clamp); \
        else hipLaunchKernelGGL((update_h_kernel<KT_, NT_, false>), grid, block, lds, st, H, k, n, ldh, AtW, ldatw, G, eps, clamp);    \
    }
    UH_CASE(1, 4) UH_CASE(2, 2) UH_CASE(4, 1)
#undef UH_CASE
    return check_launch("mu_update_h");
}

int dnmf_clamp_min(float* X, long rows, long cols, long ldx, float eps, void* stream) {
    REQUIRE(X && rows >= 1 && cols >= 1 && ldx >= cols, "clamp_min: bad arguments");
    const unsigned grid = (unsigned)std::min<long>(cdiv(rows * cols, 256), 8192);
    hipLaunchKernelGGL(clamp_kernel, dim3(grid), dim3(256), 0, S(stream), X, rows, cols, ldx, eps);
    return check_launch("clamp");
}

int dnmf_scale_cols_div(float* W, long m, int k, long ldw, const float* s, float eps, void* stream) {
    REQUIRE(W && s && m >= 1 && k >= 1 && ldw >= k, "scale_cols_div: bad arguments");
    const unsigned grid = (unsigned)std::min<long>(cdiv(m * k, 256), 8192);
    hipLaunchKernelGGL(scale_kernel<0>, dim3(grid), dim3(256), 0, S(stream), W, m, (long)k, ldw, s, eps);
    return check_launch("scale_cols_div");
}

int dnmf_scale_rows_mul(float* H, int k, long n, long ldh, const float* s, void* stream) {
    REQUIRE(H && s && n >= 1 && k >= 1 && ldh >= n, "scale_rows_mul: bad arguments");
    const unsigned grid = (unsigned)std::min<long>(cdiv((long)k * n, 256), 8192);
    hipLaunchKernelGGL(scale_kernel<1>, dim3(grid), dim3(256), 0, S(stream), H, (long)k, n, ldh, s, 0.f);
    return check_launch("scale_rows_mul");
}

}  // extern "C"
namespace {
template <typename TA>
int sqnorm_impl(const TA* A, long m, long n, long lda, double* out, void* stream) {
    REQUIRE(A && out && m >= 1 && n >= 1 && lda >= n, "sqnorm: bad arguments");
    hipStream_t st = S(stream);
    if (hipMemsetAsync(out, 0, sizeof(double), st) != hipSuccess) return fail(DNMF_EHIP, "sqnorm: memset failed");
    const bool fast = a_aligned(A) && lda % 4 == 0 && n % 4 == 0;
    const long work = fast ? m * (n / 4) : m * n;
    const unsigned grid = (unsigned)std::min<long>(cdiv(work, 256), 2048);
    if (fast) hipLaunchKernelGGL((sqnorm_kernel<true, TA>), dim3(grid), dim3(256), 0, st, A, m, n, lda, out);
    else hipLaunchKernelGGL((sqnorm_kernel<false, TA>), dim3(grid), dim3(256), 0, st, A, m, n, lda, out);
    return check_launch("sqnorm");
}
}  // namespace
extern "C" {

int dnmf_sqnorm(const float* A, long m, long n, long lda, double* out, void* stream) {
    return sqnorm_impl<float>(A, m, n, lda, out, stream);
}
int dnmf_sqnorm_bf16a(const void* A, long m, long n, long lda, double* out, void* stream) {
    return sqnorm_impl<bf16_t>((const bf16_t*)A, m, n, lda, out, stream);
}

static NnArgs nn_args(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                      int k, float eps) {
    NnArgs a{};
    a.A = A; a.lda = lda; a.m = m; a.n = n; a.W = W; a.ldw = ldw; a.H = H; a.ldh = ldh; a.k = k; a.eps = eps;
    a.nrowblk = cdiv(m, 32); a.ncolblk = (int)cdiv(n, 128);
    return a;
}

static bool nn_fast(const float* A, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k) {
    return aligned16(A) && aligned16(W) && aligned16(H) && lda % 4 == 0 && n % 4 == 0 && ldw % 4 == 0 && k % 4 == 0 &&
           ldh % 4 == 0;
}

}  // extern "C"
namespace {
template <typename TA>
int resid_sqnorm_impl(const TA* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                      int k, double* out, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && out && m >= 1 && n >= 1 && lda >= n && ldw >= k && ldh >= n, "resid_sqnorm: bad arguments");
    hipStream_t st = S(stream);
    if (hipMemsetAsync(out, 0, sizeof(double), st) != hipSuccess) return fail(DNMF_EHIP, "resid_sqnorm: memset failed");
    NnArgs a = nn_args(reinterpret_cast<const float*>(A), m, n, lda, W, ldw, H, ldh, k, 0.f);
    a.out = out;
    const bool fast = a_aligned(A) && nn_fast(W, n, lda, W, ldw, H, ldh, k);
    const dim3 grid((unsigned)cdiv(a.nrowblk * a.ncolblk, 4)), block(256);
#define RS_CASE(KT_)                                                                   \
    if (kt == KT_) {                                                                   \
        if (fast) hipLaunchKernelGGL((resid_kernel<KT_, true, TA>), grid, block, 0, st, a); \
        else hipLaunchKernelGGL((resid_kernel<KT_, false, TA>), grid, block, 0, st, a);    \
    }
    RS_CASE(1) RS_CASE(2) RS_CASE(4)
#undef RS_CASE
    return check_launch("resid_sqnorm");
}
}  // namespace
extern "C" {

int dnmf_resid_sqnorm(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                      int k, double* out, void* stream) {
    return resid_sqnorm_impl<float>(A, m, n, lda, W, ldw, H, ldh, k, out, stream);
}
int dnmf_resid_sqnorm_bf16a(const void* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh,
                            int k, double* out, void* stream) {
    return resid_sqnorm_impl<bf16_t>((const bf16_t*)A, m, n, lda, W, ldw, H, ldh, k, out, stream);
}

struct UhtPlan { int nsplit; long cols_per_split; };

static UhtPlan plan_uht(long m, long n) {
    UhtPlan u;
    const long rowtiles = cdiv(m, 128);
    long ns = std::max<long>(1, cdiv(1536, rowtiles));           // aim at >= 1536 workgroups (2 resident per CU)
    ns = std::min<long>(ns, std::max<long>(1, n / 256));         // at least 8 column tiles per split
    u.cols_per_split = round_up(cdiv(n, ns), BK);
    u.nsplit = (int)cdiv(n, u.cols_per_split);
    return u;
}

int dnmf_kl_uht(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k,
                float eps, float* UHT, long ldo, void* ws, size_t ws_bytes, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && UHT && m >= 1 && n >= 1 && lda >= n && ldw >= k && ldh >= n && ldo >= k, "kl_uht: bad arguments");
    const int kp = 32 * kt;
    NnArgs a = nn_args(A, m, n, lda, W, ldw, H, ldh, k, eps);
    const UhtPlan u = plan_uht(m, n);
    const size_t pbytes = u.nsplit > 1 ? (size_t)u.nsplit * m * kp * sizeof(float) : 0;
    const size_t need = pbytes + reduce_scratch_bytes(u.nsplit, (int)m, k);
    if (u.nsplit > 1 && (!ws || ws_bytes < need)) return fail(DNMF_EWS, "kl_uht: workspace %zu < %zu", ws_bytes, need);
    const bool split = u.nsplit > 1;
    float* out = split ? (float*)ws : UHT;
    const long ldout = split ? kp : ldo;
    const int out_cols = split ? kp : k;
    const bool fast = nn_fast(A, n, lda, W, ldw, H, ldh, k) && aligned16(out) && ldout % 4 == 0;
    const dim3 grid((unsigned)cdiv(m, 128), (unsigned)u.nsplit), block(256);
    const size_t lds = 2ul * kp * BK * sizeof(float);
    hipStream_t st = S(stream);
#define UH_CASE(KT_)                                                                                                  \
    if (kt == KT_) {                                                                                                  \
        if (fast) hipLaunchKernelGGL((kl_uht_kernel<KT_, true>), grid, block, lds, st, a, out, ldout, (long)m * kp,    \
                                     u.cols_per_split, out_cols);                                                     \
        else hipLaunchKernelGGL((kl_uht_kernel<KT_, false>), grid, block, lds, st, a, out, ldout, (long)m * kp,        \
                                u.cols_per_split, out_cols);                                                          \
    }
    UH_CASE(1) UH_CASE(2) UH_CASE(4)
#undef UH_CASE
    int rc = check_launch("kl_uht");
    if (rc || !split) return rc;
    return launch_reduce((const float*)ws, (long)m * kp, kp, u.nsplit, UHT, ldo, (int)m, k, (int)m, k,
                         (float*)((char*)ws + pbytes), st);
}

int dnmf_kl_wtu(const float* A, long m, long n, long lda, const float* W, long ldw, const float* H, long ldh, int k,
                float eps, float* WTU, long ldo, void* ws, size_t ws_bytes, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && WTU && ws && m >= 1 && n >= 1 && lda >= n && ldw >= k && ldh >= n && ldo >= n, "kl_wtu: bad arguments");
    const int kp = 32 * kt;
    NnArgs a = nn_args(A, m, n, lda, W, ldw, H, ldh, k, eps);
    const int nt = kl_nt(kt);
    TnPlan p = plan_tn(m, n, kt, nt);
    a.ncolblk = p.ncolblk;
    const long rowblks_per_chunk = std::max<long>(1, p.rows_per_chunk / 32);
    const long nchunks = cdiv(a.nrowblk, rowblks_per_chunk);
    a.P = (float*)ws; a.ldp = p.ldp; a.chunk_stride = p.ldp * kp;
    const size_t pbytes = (size_t)nchunks * a.chunk_stride * sizeof(float);
    const size_t need = pbytes + reduce_scratch_bytes((int)nchunks, k, n);
    if (ws_bytes < need) return fail(DNMF_EWS, "kl_wtu: workspace %zu < %zu", ws_bytes, need);
    const bool fast = nn_fast(A, n, lda, W, ldw, H, ldh, k);
    const dim3 grid((unsigned)(cdiv(nchunks, 4) * a.ncolblk)), block(256);   // 4 row chunks (waves) per workgroup
    const size_t lds = (size_t)kp * 32 * nt * sizeof(float);                   // the H block of the workgroup's columns
    hipStream_t st = S(stream);
#define WU_CASE(KT_, NT_)                                                                                         \
    if (kt == KT_) {                                                                                              \
        if (fast) hipLaunchKernelGGL((kl_wtu_kernel<KT_, NT_, true>), grid, block, lds, st, a, rowblks_per_chunk); \
        else hipLaunchKernelGGL((kl_wtu_kernel<KT_, NT_, false>), grid, block, lds, st, a, rowblks_per_chunk);    \
    }
    WU_CASE(1, 4) WU_CASE(2, 2) WU_CASE(4, 2)
#undef WU_CASE
    int rc = check_launch("kl_wtu");
    if (rc) return rc;
    return launch_reduce((const float*)ws, a.chunk_stride, a.ldp, (int)nchunks, WTU, ldo, k, n, k, n,
                         (float*)((char*)ws + pbytes), st);
}

int dnmf_rowsum(const float* H, int k, long n, long ldh, float* x, void* stream) {
    REQUIRE(H && x && k >= 1 && n >= 1 && ldh >= n, "rowsum: bad arguments");
    hipLaunchKernelGGL(rowsum_kernel, dim3(k), dim3(256), 0, S(stream), H, n, ldh, x);
    return check_launch("rowsum");
}

int dnmf_colsum(const float* W, long m, int k, long ldw, float* x, void* ws, size_t ws_bytes, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && W && x && ws && m >= 1 && ldw >= k, "colsum: bad arguments");
    const int kp = 32 * kt;
    const long rows_per_blk = 1024;
    const int nblk = (int)cdiv(m, rows_per_blk);
    if (ws_bytes < (size_t)nblk * kp * sizeof(float)) return fail(DNMF_EWS, "colsum: workspace too small");
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, S(stream), W, m, k, ldw, rows_per_blk, (float*)ws, kp);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(1), dim3(kp), 0, S(stream), (const float*)ws, nblk, kp, k, x);
    return check_launch("colsum");
}

int dnmf_kl_update_w(float* W, long m, int k, long ldw, const float* Sm, long lds_, const float* x, float eps,
                     void* stream) {
    REQUIRE(W && Sm && x && m >= 1 && k >= 1 && ldw >= k && lds_ >= k, "kl_update_w: bad arguments");
    const unsigned grid = (unsigned)std::min<long>(cdiv(m * k, 256), 8192);
    hipLaunchKernelGGL(kl_update_kernel<false>, dim3(grid), dim3(256), 0, S(stream), W, m, (long)k, ldw, Sm, lds_, x, eps, 0);
    return check_launch("kl_update_w");
}

int dnmf_kl_update_h(float* H, int k, long n, long ldh, const float* Sm, long lds_, const float* x, float eps,
                     int clamp, void* stream) {
    REQUIRE(H && Sm && x && n >= 1 && k >= 1 && ldh >= n && lds_ >= n, "kl_update_h: bad arguments");
    const unsigned grid = (unsigned)std::min<long>(cdiv((long)k * n, 256), 8192);
    hipLaunchKernelGGL(kl_update_kernel<true>, dim3(grid), dim3(256), 0, S(stream), H, (long)k, n, ldh, Sm, lds_, x, eps, clamp);
    return check_launch("kl_update_h");
}

static int hals_w_col_launch(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, int kk,
                             const double* prev_ss2, float eps, double* ss2_out, bool zero, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && W && AH && G && ss2_out && m >= 1 && ldw >= k && ldah >= k && kk >= 0 && kk < k, "hals_w_col: bad arguments");
    hipStream_t st = S(stream);
    if (zero && hipMemsetAsync(ss2_out, 0, sizeof(double), st) != hipSuccess) return fail(DNMF_EHIP, "hals_w_col: memset failed");
    const unsigned grid = (unsigned)std::min<long>(cdiv(m, 256), 2048);
    hipLaunchKernelGGL(hals_w_col_kernel, dim3(grid), dim3(256), 0, st, W, m, k, ldw, AH, ldah, G, 32 * kt, kk, prev_ss2,
                       eps, ss2_out);
    return check_launch("hals_w_col");
}

int dnmf_hals_w_col(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, int kk,
                    const double* prev_ss2, float eps, double* ss2_out, void* stream) {
    return hals_w_col_launch(W, m, k, ldw, AH, ldah, G, kk, prev_ss2, eps, ss2_out, true, stream);
}

int dnmf_hals_w_scale(float* W, long m, long ldw, int col, const double* ss2, void* stream) {
    REQUIRE(W && ss2 && m >= 1 && col >= 0 && ldw > col, "hals_w_scale: bad arguments");
    const unsigned grid = (unsigned)std::min<long>(cdiv(m, 256), 2048);
    hipLaunchKernelGGL(hals_w_scale_kernel, dim3(grid), dim3(256), 0, S(stream), W, m, ldw, col, ss2);
    return check_launch("hals_w_scale");
}

int dnmf_hals_update_w(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, float eps,
                       double* ss2, void* stream) {
    REQUIRE(ss2 != nullptr && k >= 1, "hals_update_w: ss2 scratch (k doubles) required");
    if (hipMemsetAsync(ss2, 0, (size_t)k * sizeof(double), S(stream)) != hipSuccess) return fail(DNMF_EHIP, "hals_update_w: memset failed");
    for (int kk = 0; kk < k; ++kk) {   // one launch per column: the column norm is a grid-wide dependency
        int rc = hals_w_col_launch(W, m, k, ldw, AH, ldah, G, kk, kk ? ss2 + kk - 1 : nullptr, eps, ss2 + kk, false, stream);
        if (rc) return rc;
    }
    return dnmf_hals_w_scale(W, m, ldw, k - 1, ss2 + k - 1, stream);
}

int dnmf_hals_update_h(float* H, int k, long n, long ldh, const float* AtW, long ldatw, const float* G, float eps,
                       void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && H && AtW && G && n >= 1 && ldh >= n && ldatw >= n, "hals_update_h: bad arguments");
    const dim3 grid((unsigned)cdiv(n, 256)), block(256);
    hipStream_t st = S(stream);
#define HH_CASE(KT_)                                                                                              \
    if (kt == KT_) {                                                                                              \
        const size_t lds = (size_t)(32 * KT_) * (32 * KT_) * sizeof(float);                                        \
        hipLaunchKernelGGL((hals_h_kernel<32 * KT_>), grid, block, lds, st, H, k, n, ldh, AtW, ldatw, G, eps);     \
    }
    HH_CASE(1) HH_CASE(2)
#undef HH_CASE
    if (kt == 4) {
        const int kp = 128;
        const size_t lds = (size_t)(kp * kp + kp * 64) * sizeof(float);   // 96 KiB
        static bool once = false;
        if (!once) { allow_lds(hals_h_kernel_lds, lds); once = true; }
        hipLaunchKernelGGL(hals_h_kernel_lds, dim3((unsigned)cdiv(n, 64)), dim3(64), lds, st, H, k, n, ldh, AtW, ldatw, G, kp, eps);
    }
    return check_launch("hals_update_h");
}

}  // extern "C"
namespace {
template <typename TA>
int mu_fro_step_impl(const TA* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k,
                     float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && ws && m >= 1 && n >= 1, "mu_fro_step: bad arguments");
    const WsLayout L = ws_layout(m, n, k);
    if (ws_bytes < L.total) return fail(DNMF_EWS, "mu_fro_step: workspace %zu < %zu", ws_bytes, L.total);
    char* base = (char*)ws;
    float* G = (float*)(base + L.g_off);
    float* Sb = (float*)(base + L.s_off);
    void* part = base + L.part_off;
    const size_t part_bytes = L.total - L.part_off;
    int rc;
    if (w_update) {                                                                   // dist_nmf.py:716-732
        if ((rc = dnmf_gram_hht(H, k, n, ldh, G, part, part_bytes, stream))) return rc;
        if ((rc = aht_update_w_impl<TA>(A, m, n, lda, H, k, ldh, G, W, ldw, eps, stream))) return rc;
    }
    const long ldatw = round_up(n, 4);                                                // dist_nmf.py:736-751
    if ((rc = dnmf_gram_wtw(W, m, k, ldw, G, part, part_bytes, stream))) return rc;
    if ((rc = wta_impl<TA>(A, m, n, lda, W, k, ldw, Sb, ldatw, part, part_bytes, stream))) return rc;
    if ((rc = dnmf_mu_update_h(H, k, n, ldh, Sb, ldatw, G, eps, clamp, stream))) return rc;
    if (clamp) return dnmf_clamp_min(W, m, k, ldw, eps, stream);                      // pyDNMF.py:155-157
    return DNMF_OK;
}
}  // namespace
extern "C" {

int dnmf_mu_fro_step(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k,
                     float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream) {
    return mu_fro_step_impl<float>(A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, clamp, ws, ws_bytes, stream);
}
int dnmf_mu_fro_step_bf16a(const void* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k,
                           float eps, int w_update, int clamp, void* ws, size_t ws_bytes, void* stream) {
    return mu_fro_step_impl<bf16_t>((const bf16_t*)A, m, n, lda, W, ldw, H, ldh, k, eps, w_update, clamp, ws, ws_bytes, stream);
}

int dnmf_mu_kl_step(const float* A, long m, long n, long lda, float* W, long ldw, float* H, long ldh, int k, float eps,
                    int w_update, int clamp, void* ws, size_t ws_bytes, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && ws && m >= 1 && n >= 1, "mu_kl_step: bad arguments");
    const WsLayout L = ws_layout(m, n, k);
    if (ws_bytes < L.total) return fail(DNMF_EWS, "mu_kl_step: workspace %zu < %zu", ws_bytes, L.total);
    char* base = (char*)ws;
    float* Sb = (float*)(base + L.s_off);
    float* x = (float*)(base + L.x_off);
    void* part = base + L.part_off;
    const size_t part_bytes = L.total - L.part_off;
    int rc;
    if (w_update) {                                                                   // dist_nmf.py:813-830
        const long ldu = round_up(k, 4);
        if ((rc = dnmf_rowsum(H, k, n, ldh, x, stream))) return rc;
        if ((rc = dnmf_kl_uht(A, m, n, lda, W, ldw, H, ldh, k, eps, Sb, ldu, part, part_bytes, stream))) return rc;
        if ((rc = dnmf_kl_update_w(W, m, k, ldw, Sb, ldu, x, eps, stream))) return rc;
    }
    const long ldo = round_up(n, 4);                                                  // dist_nmf.py:832-849
    if ((rc = dnmf_colsum(W, m, k, ldw, x, part, part_bytes, stream))) return rc;
    if ((rc = dnmf_kl_wtu(A, m, n, lda, W, ldw, H, ldh, k, eps, Sb, ldo, part, part_bytes, stream))) return rc;
    if ((rc = dnmf_kl_update_h(H, k, n, ldh, Sb, ldo, x, eps, clamp, stream))) return rc;
    if (clamp) return dnmf_clamp_min(W, m, k, ldw, eps, stream);
    return DNMF_OK;
}

}  // extern "C"
