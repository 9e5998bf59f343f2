// dnmf_common.h -- types, error plumbing and the element loaders shared by every kernel family.
// Part of libdnmf_hip.so (included by every translation unit csrc/*.hip).
#pragma once
#include <type_traits>
#include <utility>
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

// ----------------------------------------------------------------------------------------------- errors
// One thread-local message buffer for the whole library (csrc/dnmf.hip owns it; dnmf_last_error returns it); the other
// translation units reach it through this library-internal accessor.
constexpr int DNMF_ERRBUF = 512;
__attribute__((visibility("hidden"))) char* dnmf_errbuf_();

// MUBUF intrinsics (see "buffer addressing" below): external declarations bound to the LLVM intrinsics, no definition
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ float buf_ld_f32(i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ f32x2 buf_ld_f32x2(i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2f32");
__device__ f32x4 buf_ld_f32x4(i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void buf_st_f32(float v, i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
__device__ void buf_st_f32x2(f32x2 v, i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.v2f32");
__device__ void buf_st_f32x4(f32x4 v, i32x4 rsrc, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4f32");

// ----------------------------------------------------------------------------------------------- batched launches
// B independent problems of ONE shape in one launch: blockIdx.z = problem (the perturbations of an NMFk sweep,
// include/dnmf.h "whole fits").  The host lays everything out for problem 0 exactly as for a single problem -- same plans,
// same grids, same workspace offsets -- and hands every kernel this table: up to four operand families (the data blocks,
// the W factors, the H factors, the workspaces), each the byte extent [lo, hi) of problem 0's member and the byte distance to
// the next problem's.  A kernel's first statement moves each of its pointers into its own problem (`rebase`): whatever
// family the pointer falls in, + blockIdx.z x that family's stride.  Everything after that line is the single-problem code,
// so problem z of a batched launch executes the instructions of a single launch on the same operands -- bit-identical results.
// n = 0 (and gridDim.z = 1) outside a batched fit: the rebase then adds zero.  Strides are multiples of 16 bytes, so the
// host's alignment decisions for problem 0 hold for every problem.
struct BatchFam { unsigned long lo, hi; long stride; };
struct BatchTab { int n; int z0; BatchFam f[4]; };      // z0: first problem of this launch (a batch cut into several launches)
// the calling thread's batch state (csrc/dnmf.hip owns it; set by the *_fit entry points around their launch sequence)
struct BatchCtx { int B; BatchTab tab; };
__attribute__((visibility("hidden"))) BatchCtx* dnmf_batch_();
// process-wide switch of the kernels that need co-residency (csrc/dnmf.hip owns it; dnmf_set_persistent)
__attribute__((visibility("hidden"))) int dnmf_persistent_on_();
__attribute__((visibility("hidden"))) void dnmf_persistent_set_(int on);

__device__ __forceinline__ long batch_off(const void* p, const BatchTab& bt) {
    const unsigned long a = (unsigned long)p;
    long d = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < bt.n && a >= bt.f[i].lo && a < bt.f[i].hi) d = bt.f[i].stride;
    return d * (long)(blockIdx.z + bt.z0);
}
// pointer arithmetic, not integer arithmetic: the kernel-argument pointer keeps its global address space through it
template <typename T>
__device__ __forceinline__ void rebase(T*& p, const BatchTab& bt) { p = (T*)((char*)p + batch_off((const void*)p, bt)); }
// the same for a __restrict__-qualified kernel parameter (a reference to it would drop the qualifier)
#define REBASE(p) (p) = (decltype(p))((char*)(p) + batch_off((const void*)(p), bt))

// every kernel of the batched paths is launched through this: grid.z = problems, the table as the last argument
#define DNMF_LAUNCH(kernel, grid, block, lds, st, ...)                                           \
    do {                                                                                         \
        const BatchCtx* bc_ = dnmf_batch_();                                                     \
        dim3 g_ = (grid);                                                                        \
        g_.z = (unsigned)bc_->B;                                                                 \
        hipLaunchKernelGGL(kernel, g_, block, lds, st, __VA_ARGS__, bc_->tab);                   \
    } while (0)

// persistent HALS W sweep (csrc/dnmf_hals.h); the workspace query (csrc/dnmf.hip) sizes its slot slab from these
constexpr int HALS_WG = 512;                 // threads per workgroup = rows per workgroup
constexpr int HALS_MAX_WG = 1024;            // slots per column (2 per polling thread at most)
// The same sweep ACROSS ranks (W's rows spread over the ranks of a row grid: the column norms are global, utils.py:388-391): every
// workgroup publishes its column partial into its slot of EVERY rank's slab (slabs live in the IPC-exported regions of
// csrc/dnmf_comm.hip), polls its own rank's slab and sums all ranks' slots in slot order.  P = 0: the local sweep.
struct HalsPeers {
    int P, rank;
    int first;                               // this rank's first slot = workgroups of the ranks before it
    int total;                               // slots per column = workgroups of all ranks (<= HALS_MAX_WG)
    unsigned long long patience;             // ticks of the 100 MHz wall clock a poll may see no progress
    unsigned long long* slab[DNMF_DIRECT_MAX_RANKS];   // every rank's slab of this sweep's parity, as mapped into this process
};

namespace {

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dnmf_errbuf_(), DNMF_ERRBUF, fmt, ap);
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

// Experiment switches exist only in the tuning build (-DDNMF_TUNING -> tools/_build/libdnmf_hip_tune.so, tools/README.md);
// the shipped library has the defaults compiled in and never reads the environment.
#ifdef DNMF_TUNING
inline long tune(const char* name, long dflt) { const char* v = getenv(name); return v ? atol(v) : dflt; }
#else
constexpr long tune(const char*, long dflt) { return dflt; }
#endif

// host twins of `rebase`: problem z's copy of a pointer laid out for problem 0, and a memset of every problem's copy
inline void* batch_ptr(const void* p, int z) {
    const BatchCtx* bc = dnmf_batch_();
    const unsigned long a = (unsigned long)p;
    long d = 0;
    for (int i = 0; i < bc->tab.n; ++i)
        if (a >= bc->tab.f[i].lo && a < bc->tab.f[i].hi) d = bc->tab.f[i].stride;
    return (char*)p + d * z;
}
inline hipError_t batch_memset(void* p, int value, size_t bytes, hipStream_t st) {
    const int B = dnmf_batch_()->B;
    if (B == 1) return hipMemsetAsync(p, value, bytes, st);
    // the problems' copies are `stride` apart: ONE two-dimensional fill (B rows of `bytes`) instead of B fills -- the slot slab of
    // the persistent HALS sweep is reset before every sweep, and at the NMFk sweep shape 20 fills per sweep were a third of the
    // GPU time of a batched fit (profiles/r05_c5s_*)
    const long stride = (char*)batch_ptr(p, 1) - (char*)p;
    if (stride >= (long)bytes) return hipMemset2DAsync(p, (size_t)stride, value, bytes, (size_t)B, st);
    for (int z = 0; z < B; ++z) {
        const hipError_t e = hipMemsetAsync(batch_ptr(p, z), value, bytes, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
__host__ __device__ inline long cdiv(long a, long b) { return (a + b - 1) / b; }
inline long round_up(long a, long b) { return cdiv(a, b) * b; }

// ----------------------------------------------------------------------------------------------- buffer addressing
// MUBUF loads / stores: address = descriptor base (4 SGPRs) + one 32-bit VGPR offset + one SGPR offset + immediate.  A
// tile whose rows differ only by a wave-uniform offset then needs ONE address register per lane for all its loads and
// stores, where flat/global addressing makes hipcc keep a 64-bit VGPR pair per row alive from the load to the store (the
// update kernels spilled because of exactly that).  The descriptor claims 2 GiB from its base and every valid offset
// (VGPR + SGPR + immediate) stays below that; a lane is switched off by giving it the offset BUF_OOB (>= the claimed size:
// the hardware returns 0 for such a load and drops such a store), so edge tiles need no exec-masked branches.
// RULE: a store of more than 8 bytes must not use the SGPR offset (pass 0 and add constants to the VGPR offset, they
// fold into the immediate): with an SGPR offset the gfx950 hardware may still be reading the data registers when the
// next VALU write hits them, and hipcc pads that hazard only for the immediate form (see update_w_seq_tile).
// HAZARD 2 (round 4, measured): an MFMA reads its C operand DURING its passes (64 cycles for 32x32x2), and neither the hardware
// nor hipcc's hazard recognizer protects those registers against the write-back of an LDS / global LOAD issued after it (only
// VALU writes are padded).  When the C operand is a register block that dies with that MFMA -- eps as the initial value of an
// accumulator chain: `acc = MFMA(a, b, epsv)` -- the allocator may hand the block to the next ds_read / buffer_load, and
// data that returns within the MFMA's passes corrupts the upper accumulator rows.  Kernels that start a chain from eps therefore
// pin the order [loads of the step] sched_barrier [the step's MFMAs, at least two on that chain] sched_barrier [next loads]:
// the second MFMA cannot issue before the first has finished reading C (dnmf_kluht.h, kl_wtu_chunk_pipe, kl_uht_body);
// loops without barriers start from zero (an inline constant, no registers) and add eps in the epilogue.
constexpr int BUF_OOB = (int)0x80000000u;

// descriptor for `base` (wave-uniform): raw buffer (stride 0), 2 GiB window, gfx9 data format word
__device__ __forceinline__ i32x4 buf_rsrc(const void* base) {
    const unsigned long a = (unsigned long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)(a & 0xffffffffu));
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
    r[2] = (int)0x80000000u;
    r[3] = 0x00020000;
    return r;
}

// AUX = the instruction's cache-policy bits (0 = default, 2 = nontemporal / streaming)
template <int V, int AUX = 0>
__device__ __forceinline__ void buf_load(float (&d)[V], i32x4 rsrc, int voff, int soff) {
    if constexpr (V == 4) { const f32x4 v = buf_ld_f32x4(rsrc, voff, soff, AUX); d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3]; }
    else if constexpr (V == 2) { const f32x2 v = buf_ld_f32x2(rsrc, voff, soff, AUX); d[0] = v[0]; d[1] = v[1]; }
    else d[0] = buf_ld_f32(rsrc, voff, soff, AUX);
}

template <int V, int AUX = 0>
__device__ __forceinline__ void buf_store(const float (&d)[V], i32x4 rsrc, int voff, int soff) {
    if constexpr (V == 4) buf_st_f32x4(f32x4{d[0], d[1], d[2], d[3]}, rsrc, voff, soff, AUX);
    else if constexpr (V == 2) buf_st_f32x2(f32x2{d[0], d[1]}, rsrc, voff, soff, AUX);
    else buf_st_f32(d[0], rsrc, voff, soff, AUX);
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E).  `#pragma unroll` is a request hipcc drops for large
// bodies ("loop not unrolled"), and an array indexed by a runtime loop variable then lives in scratch memory instead of
// registers; this form cannot be left rolled.
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// C/D row of accumulator register `reg` for lane-half h
__device__ __forceinline__ int crow(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// a / d for 0 < d < inf away from the ends of the exponent range (d = S + eps here): reciprocal, one Newton step, quotient
// and one residual correction -- the division sequence hipcc emits (v_div_scale / v_div_fmas / v_div_fixup) without its
// range scaling: 6 VALU instructions instead of 10, same result wherever no intermediate underflows or overflows.
__device__ __forceinline__ float div_pos(float a, float d) {
    float r = __builtin_amdgcn_rcpf(d);
    r = fmaf(fmaf(-d, r, 1.0f), r, r);
    float q = a * r;
    return fmaf(fmaf(-d, q, a), r, q);
}

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
    if constexpr (V == 8) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 w = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int q = 0; q < 4; ++q) { d[2 * q] = bf16_lo(w[q]); d[2 * q + 1] = bf16_hi(w[q]); }
    } else if constexpr (V == 4) {
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
        if constexpr (V == 8) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 x = NTL ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)) : *reinterpret_cast<const u32x4*>(p);
            w[0] = x[0]; w[1] = x[1]; w[2] = x[2]; w[3] = x[3];
        } else if constexpr (V == 4) {
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

}  // namespace
