// dnmf_comm.hip -- the grid exchanges INSIDE the library: RCCL communicators for the p_r x p_c grid (dist_comm.py:16-56) and
// whole 1D steps that enqueue  kernels -> ncclAllReduce -> kernels  on the caller's stream with no host code in between
// (dist_nmf.py:663-771 Frobenius, :776-869 KL).  A host that is not PyTorch (the reference's mpi4py driver)
// gets the packed single allreduce and the overlapped H phase through these entry points; the PyTorch host can use them
// instead of torch.distributed (params.exchange = 'native').
//
// RCCL is bound at run time (dlopen): the library loads and every compute entry point works on a host without RCCL; the
// first dnmf_comm_* call looks for an RCCL that is ALREADY in the process (PyTorch's own copy: two RCCL / HIP runtime
// instances in one process do not share device state) and only then loads librccl.so.1 by name.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "dnmf_common.h"
#include "dnmf_host.h"

// csrc/dnmf_hals.hip: the persistent W sweep with the column norms summed over the ranks of `pe` (nullptr: applicability check only)
__attribute__((visibility("hidden"))) int dnmf_hals_sweep_w_peers_(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G,
                                                                   float eps, void* ws, size_t ws_bytes, const HalsPeers* pe, int* nwg,
                                                                   void* stream);

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, ncclConfig_t*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;                  // optional
    const char* origin = "";
};

Rccl* rccl() {
    // bound once (function-local static: initialisation is serialised by the language, callers on any thread see the result)
    static Rccl* const inst = []() -> Rccl* {
        static Rccl r;
        const char* resident[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : resident)
            if (!r.handle && (r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) r.origin = "already in the process";
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if (!r.handle && (r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) r.origin = n;
        if (!r.handle) return nullptr;
#define BIND(field, sym) \
        r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym)); \
        if (!r.field) return nullptr;
        BIND(GetUniqueId, "ncclGetUniqueId") BIND(CommInitRank, "ncclCommInitRank") BIND(CommDestroy, "ncclCommDestroy")
        BIND(AllReduce, "ncclAllReduce") BIND(AllGather, "ncclAllGather")
        BIND(ReduceScatter, "ncclReduceScatter") BIND(GroupStart, "ncclGroupStart") BIND(GroupEnd, "ncclGroupEnd")
        BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
        // optional symbols (round 4): ncclCommSplit is needed by 2D grids only -- a 1D grid must not lose the in-library exchange
        // to an RCCL without it; ncclGetVersion feeds dnmf_comm_rccl_version
        r.CommSplit = reinterpret_cast<decltype(r.CommSplit)>(dlsym(r.handle, "ncclCommSplit"));
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(r.handle, "ncclGetVersion"));
        return &r;
    }();
    return inst;
}

constexpr int MAX_CHUNKS = 8;

}  // namespace

struct dnmf_comm {
    ncclComm_t world = nullptr, row = nullptr, col = nullptr;   // row: the p_r ranks of one grid column; col: the p_c ranks of one grid row
    int nranks = 1, rank = 0, p_r = 1, p_c = 1;
    int overlap_chunks = 1;
    int always = 0;                                              // testing: a one-rank communicator still issues its RCCL calls
    int null_exchange = 0;                                       // measurement: the steps skip their RCCL calls (wrong results on > 1 rank)
    dnmf_collective_fn hook = nullptr;                           // host-supplied collectives instead of RCCL (dnmf_comm_create_hosted)
    void* hook_user = nullptr;
    hipStream_t xstream = nullptr;                               // exchanges of the overlapped H phase run here
    hipEvent_t ready[MAX_CHUNKS] = {}, done[MAX_CHUNKS] = {};
    // direct (two-shot) allreduce over IPC peer buffers (dnmf_comm_direct_*): every rank's region as mapped into this process
    int direct_on = 0;
    size_t direct_cap = 0;                                       // floats per message
    char* direct_peer[DNMF_DIRECT_MAX_RANKS] = {};               // [rank] = the own region
    unsigned long long direct_seq = 0, direct_small_seq = 0;
    unsigned long long direct_patience = 30ull * 100000000ull;   // 30 s of the 100 MHz wall clock (dnmf_comm_set_direct_timeout)
    int emulated = 0;                                            // measurement: ONE member of a p_r x p_c grid, every collective on a one-rank RCCL communicator
    // cross-rank persistent HALS W sweep over the peer regions (hals_sweep_exchanged): the ranks' agreement for the last shape seen
    unsigned long long hals_seq = 0;                             // sweeps issued: its parity selects the slot slab
    long xs_m = -1; int xs_k = -1, xs_ok = 0, xs_first = 0, xs_total = 0;
    // WHEN the ranks agree again is a function of things every rank sees alike -- the fit counter (dnmf_comm_fit_begin: PyNMF calls it on
    // every rank at the start of every fit), the rank k, the process-wide persistent switch -- never of a rank's LOCAL row count: with
    // an asymmetric split two fits of different global m can change the row count on some ranks only, and an agreement
    // that only those ranks enter is a collective mismatch
    int xs_epoch = 0, xs_seen_epoch = -1, xs_pers = -1;
    float* agree_buf = nullptr;                                  // DNMF_DIRECT_MAX_RANKS device floats of the comm's own (the caller's workspace may be absent)
};

namespace {

int nccl_fail(const char* what, ncclResult_t e) {
    Rccl* r = rccl();
    return fail(DNMF_ECOMM, "%s: RCCL error %d (%s)", what, (int)e, r ? r->GetErrorString(e) : "?");
}
#define NCCL_OK(call, what) do { ncclResult_t e_ = (call); if (e_ != ncclSuccess) return nccl_fail(what, e_); } while (0)
#define HIP_OK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(DNMF_EHIP, "%s: %s", what, hipGetErrorString(e_)); } while (0)

inline size_t pad64(size_t x) { return (x + 63) / 64 * 64; }

// The three collectives of the path over a GROUP of the grid: 0 = all ranks, 1 = cart_1d_row (the p_r ranks of a grid column),
// 2 = cart_1d_column (the p_c ranks of a grid row) -- dist_comm.py:16-56.  A group of one member is the identity (unless `always`
// asks a one-rank communicator to issue its calls anyway); a hosted communicator hands every call to the host's function.
enum { G_WORLD = 0, G_ROW = 1, G_COL = 2 };
enum { OP_ALLREDUCE = DNMF_ALLREDUCE, OP_ALLGATHER = DNMF_ALLGATHER, OP_REDUCE_SCATTER = DNMF_REDUCE_SCATTER };

int group_size(const dnmf_comm* cm, int g) { return g == G_WORLD ? cm->nranks : g == G_ROW ? cm->p_r : cm->p_c; }
ncclComm_t group_comm(const dnmf_comm* cm, int g) {
    if (g == G_WORLD) return cm->world;
    if (g == G_ROW) return cm->row ? cm->row : (cm->p_c == 1 ? cm->world : nullptr);   // 1D grids: the long axis IS the world
    return cm->col ? cm->col : (cm->p_r == 1 ? cm->world : nullptr);
}
// 0: run the collective on *out; 1: identity (nothing to exchange); < 0: error
int resolve(dnmf_comm* cm, int g, ncclComm_t* out, const char* what) {
    if (cm->null_exchange) return 1;
    if (group_size(cm, g) == 1 && (!cm->always || cm->hook)) return 1;   // (a hosted communicator never sees a group of one)
    if (cm->hook) { *out = nullptr; return 0; }
    if (cm->emulated) { *out = cm->world; return 0; }            // every group's call is issued for real, on the one-rank communicator
    ncclComm_t c = group_comm(cm, g);
    if (!c && group_size(cm, g) == 1) c = cm->world;              // `always` on a one-rank communicator
    if (!c) return fail(DNMF_EINVAL, "%s: sub-communicator %d missing", what, g);
    *out = c;
    return 0;
}

// ---------------------------------------------------------------------------------------------- direct two-shot allreduce
// SURVEY section 5 / VERDICT r03: the packed [W^T A | W^T W] message of config 3 is 2 MiB -- latency-bound for a ring over
// point-to-point xGMI links.  The direct form reads PEER memory instead: every rank owns a region of uncached device memory
// exported with hipIpcGetMemHandle and mapped by all the others;
//   region = [flag1[P] | flag2[P] | status]  [send, parity 0 | send, parity 1]  [reduced, parity 0 | reduced, parity 1]  [HALS slot slab x 2]
// and one allreduce is, on the caller's stream:  copy the message into send[parity]  ->  signal + wait (everybody has written)  ->
// REDUCE: rank r sums chunk r of all P send buffers in rank order (one owner per element: every rank ends with the same bits,
// whatever the timing) into reduced[parity]  ->  signal + wait  ->  GATHER every chunk from its owner.  Four small launches; the
// stream order between them is the grid-wide barrier, so nothing needs to be co-resident, and each wait is ONE workgroup
// spinning on flags the peers store with system scope.  Alternating parity makes a trailing barrier unnecessary: a rank reaches
// call s + 2 only after every peer has signalled in call s + 1, i.e. has finished reading the buffers of call s.
// Remote data are read with system-scope loads (no stale lines from the previous step: the same addresses carry new data every
// call).  A wait that sees no progress for the configured time (30 s by default) sets the region's status word (dnmf_comm_direct_status) and gives up.
// NOT measured on more than one GPU (this pool has none): bench.py treats it as a third arm of its warm-up A/B and only after
// its result has matched RCCL's on the warm-up step.
constexpr unsigned long long DIRECT_MAGIC = 0x444e4d4644495231ull;      // "DNMFDIR1"
constexpr size_t DIRECT_HDR = 8192;       // flag1 @0, flag2 @1024, status @2048, small-message flags @3072, small-message slots @4096
// behind the four message buffers: the two slot slabs (by sweep parity) of the cross-rank persistent HALS W sweep, 1 MiB each
constexpr size_t HALS_SLAB_BYTES = (size_t)DNMF_TUNED_MAX_K * HALS_MAX_WG * sizeof(unsigned long long);
inline size_t direct_hals_off(size_t cap) { return (DIRECT_HDR + 4 * cap * sizeof(float) + 255) & ~size_t(255); }
struct DirectArgs {
    char* peer[DNMF_DIRECT_MAX_RANKS];
    int P, rank, par;
    size_t cap;                      // floats per send / reduced buffer
    size_t count, chunk;             // message length, floats per owner
    unsigned long long seq;
    unsigned long long patience;     // ticks of the 100 MHz wall clock a wait may see no progress (dnmf_comm_set_direct_timeout)
};
__device__ __forceinline__ unsigned long long* d_flags(char* region, int which) { return reinterpret_cast<unsigned long long*>(region + 1024 * which); }
__device__ __forceinline__ float* d_send(char* region, size_t cap, int par) { return reinterpret_cast<float*>(region + DIRECT_HDR) + (size_t)par * cap; }
__device__ __forceinline__ float* d_red(char* region, size_t cap, int par) { return reinterpret_cast<float*>(region + DIRECT_HDR) + (size_t)(2 + par) * cap; }

__global__ __launch_bounds__(64) void direct_signal_wait_kernel(DirectArgs a, int which) {
    const int q = threadIdx.x;
    __threadfence_system();                                        // (everything this stream wrote before is visible first)
    if (q < a.P) __hip_atomic_store(d_flags(a.peer[q], which) + a.rank, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (q < a.P) {
        const unsigned long long* mine = d_flags(a.peer[a.rank], which) + q;
        const unsigned long long t0 = wall_clock64();             // (100 MHz, independent of the shader clock)
        while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > a.patience) {                // a peer is gone -- say so instead of hanging the GPU
                __hip_atomic_store(d_flags(a.peer[a.rank], 2), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __threadfence_system();
}

__device__ __forceinline__ f32x2 load_sys2(const float* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return f32x2{__uint_as_float((unsigned)(v & 0xffffffffu)), __uint_as_float((unsigned)(v >> 32))};
}

// this rank's chunk of the sum, in rank order, into its reduced buffer (pairs of floats: count and chunk are even)
__global__ __launch_bounds__(256) void direct_reduce_kernel(DirectArgs a) {
    const size_t beg = (size_t)a.rank * a.chunk, end = std::min(beg + a.chunk, a.count);
    float* red = d_red(a.peer[a.rank], a.cap, a.par);
    for (size_t i = beg + 2 * ((size_t)blockIdx.x * 256 + threadIdx.x); i < end; i += 2 * (size_t)gridDim.x * 256) {
        f32x2 s = {0.f, 0.f};
        for (int q = 0; q < a.P; ++q) {
            const f32x2 v = load_sys2(d_send(a.peer[q], a.cap, a.par) + i);
            s[0] += v[0]; s[1] += v[1];
        }
        *reinterpret_cast<f32x2*>(red + i) = s;
    }
}

__global__ __launch_bounds__(256) void direct_gather_kernel(DirectArgs a, float* __restrict__ out) {
    for (size_t i = 2 * ((size_t)blockIdx.x * 256 + threadIdx.x); i < a.count; i += 2 * (size_t)gridDim.x * 256) {
        const int owner = (int)(i / a.chunk);
        const f32x2 v = load_sys2(d_red(a.peer[owner], a.cap, a.par) + i);
        *reinterpret_cast<f32x2*>(out + i) = v;
    }
}

int direct_allreduce(dnmf_comm* cm, float* buf, size_t count, hipStream_t st) {
    REQUIRE(cm->direct_peer[cm->rank] && count <= cm->direct_cap, "allreduce(direct): not set up for %zu floats", count);
    REQUIRE(count % 2 == 0 && ((uintptr_t)buf & 7) == 0, "allreduce(direct): the message must be an even number of floats, 8-byte aligned");
    DirectArgs a{};
    for (int q = 0; q < cm->nranks; ++q) a.peer[q] = cm->direct_peer[q];
    a.P = cm->nranks; a.rank = cm->rank; a.cap = cm->direct_cap; a.count = count;
    a.chunk = (cdiv((long)count, cm->nranks) + 1) / 2 * 2;
    a.seq = ++cm->direct_seq;
    a.par = (int)(a.seq & 1);
    a.patience = cm->direct_patience;
    float* send = reinterpret_cast<float*>(cm->direct_peer[cm->rank] + DIRECT_HDR) + (size_t)a.par * a.cap;
    HIP_OK(hipMemcpyAsync(send, buf, count * sizeof(float), hipMemcpyDeviceToDevice, st), "allreduce(direct): copy");
    const unsigned g1 = (unsigned)std::max<long>(1, std::min<long>(64, cdiv((long)a.chunk, 512)));
    const unsigned g2 = (unsigned)std::max<long>(1, std::min<long>(256, cdiv((long)count, 512)));
    hipLaunchKernelGGL(direct_signal_wait_kernel, dim3(1), dim3(64), 0, st, a, 0);
    hipLaunchKernelGGL(direct_reduce_kernel, dim3(g1), dim3(256), 0, st, a);
    hipLaunchKernelGGL(direct_signal_wait_kernel, dim3(1), dim3(64), 0, st, a, 1);
    hipLaunchKernelGGL(direct_gather_kernel, dim3(g2), dim3(256), 0, st, a, buf);
    return check_launch("allreduce(direct)");
}

// ONE-shot form for a few doubles (the column norms of the HALS W sweep, utils.py:388-391: k DEPENDENT 8-byte allreduces per
// iteration when W's rows are spread over ranks -- pure latency): a single launch of one wave.  Thread q pushes this rank's
// values into slot [parity][rank] of peer q's region and then raises its flag there; when all P flags of the own region have
// arrived the values are summed in rank order -- identical bits everywhere.  Same parity argument as above.
constexpr int DIRECT_SMALL_MAX = 8;
__global__ __launch_bounds__(64) void direct_small_f64_kernel(DirectArgs a, double* __restrict__ buf, int count) {
    const int q = threadIdx.x;
    double mine[DIRECT_SMALL_MAX];
#pragma unroll
    for (int i = 0; i < DIRECT_SMALL_MAX; ++i) mine[i] = i < count ? buf[i] : 0.0;
    __threadfence_system();
    if (q < a.P) {
        double* slot = reinterpret_cast<double*>(a.peer[q] + 4096) + ((size_t)a.par * DNMF_DIRECT_MAX_RANKS + a.rank) * DIRECT_SMALL_MAX;
        for (int i = 0; i < count; ++i)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot) + i, (unsigned long long)__double_as_longlong(mine[i]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        __threadfence_system();
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.peer[q] + 3072) + a.rank, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long* flag = reinterpret_cast<const unsigned long long*>(a.peer[a.rank] + 3072) + q;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > a.patience) {
                __hip_atomic_store(d_flags(a.peer[a.rank], 2), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __syncthreads();
    if (q < count) {
        const unsigned long long* slots = reinterpret_cast<const unsigned long long*>(a.peer[a.rank] + 4096) + (size_t)a.par * DNMF_DIRECT_MAX_RANKS * DIRECT_SMALL_MAX;
        double s = 0.0;
        for (int r = 0; r < a.P; ++r)
            s += __longlong_as_double((long long)__hip_atomic_load(slots + (size_t)r * DIRECT_SMALL_MAX + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        buf[q] = s;
    }
}

int direct_allreduce_small_f64(dnmf_comm* cm, double* buf, size_t count, hipStream_t st) {
    DirectArgs a{};
    for (int q = 0; q < cm->nranks; ++q) a.peer[q] = cm->direct_peer[q];
    a.P = cm->nranks; a.rank = cm->rank; a.cap = cm->direct_cap; a.count = count;
    a.seq = ++cm->direct_small_seq;
    a.par = (int)(a.seq & 1);
    a.patience = cm->direct_patience;
    hipLaunchKernelGGL(direct_small_f64_kernel, dim3(1), dim3(64), 0, st, a, buf, (int)count);
    return check_launch("allreduce(direct, small)");
}

// in-place SUM over the group
int allreduce_f32(dnmf_comm* cm, int g, float* buf, size_t count, hipStream_t st) {
    if (cm->direct_on && g == G_WORLD && cm->nranks > 1 && !cm->null_exchange && count <= cm->direct_cap && count % 2 == 0 &&
        ((uintptr_t)buf & 7) == 0)
        return direct_allreduce(cm, buf, count, st);
    ncclComm_t c;
    const int r = resolve(cm, g, &c, "allreduce");
    if (r) return r < 0 ? r : DNMF_OK;
    if (cm->hook) return cm->hook(cm->hook_user, OP_ALLREDUCE, g, buf, buf, count, st) ? fail(DNMF_ECOMM, "allreduce: the host collective failed") : DNMF_OK;
    NCCL_OK(rccl()->AllReduce(buf, buf, count, ncclFloat32, ncclSum, c, st), "allreduce");
    return DNMF_OK;
}

// in-place SUM of doubles (the column norms of the HALS W sweep, utils.py:388-391)
int allreduce_f64(dnmf_comm* cm, int g, double* buf, size_t count, hipStream_t st) {
    if (cm->direct_on && g == G_WORLD && cm->nranks > 1 && !cm->null_exchange && count >= 1 && count <= DIRECT_SMALL_MAX)
        return direct_allreduce_small_f64(cm, buf, count, st);
    ncclComm_t c;
    const int r = resolve(cm, g, &c, "allreduce");
    if (r) return r < 0 ? r : DNMF_OK;
    if (cm->hook)
        return cm->hook(cm->hook_user, DNMF_ALLREDUCE_F64, g, buf, buf, count, st)
                   ? fail(DNMF_ECOMM, "allreduce: the host collective failed") : DNMF_OK;
    NCCL_OK(rccl()->AllReduce(buf, buf, count, ncclFloat64, ncclSum, c, st), "allreduce");
    return DNMF_OK;
}

// The W sweep of HALS with cross-rank column norms (dist_nmf.py:884-891 with utils.norm :388-391): column kernel, 8-byte
// allreduce, next column (which first applies the pending normalisation), final scale.  ss2 = k device doubles.
// Do ALL ranks take the one-launch cross-rank sweep for this shape?  Decided once per (rows, rank k) by the ranks together: every rank
// contributes its workgroup count (0: the persistent sweep does not apply here -- too many rows to keep resident, workspace too
// small) through a world allreduce; the sweep is on when every rank can and the slots fit a column of the slab.  The counts also
// give each rank its first slot.  (Host-synchronous, once per shape.)
int hals_xsweep_agree(dnmf_comm* cm, float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, float eps, void* ws,
                      size_t ws_bytes, hipStream_t st) {
    cm->xs_m = m; cm->xs_k = k; cm->xs_ok = 0; cm->xs_seen_epoch = cm->xs_epoch; cm->xs_pers = dnmf_persistent_on_();
    int nwg = 0;
    const bool mine = ws && ws_bytes >= 64 * sizeof(float) &&
                      dnmf_hals_sweep_w_peers_(W, m, k, ldw, AH, ldah, G, eps, ws, ws_bytes, nullptr, &nwg, (void*)st) == 0;
    float host[DNMF_DIRECT_MAX_RANKS] = {};
    host[cm->rank] = mine ? (float)nwg : -1.0e6f;                  // (a rank that cannot poisons the sum of its own slot)
    // the comm's own buffer: a rank that cannot take the sweep (no workspace, too small a one) still joins the allreduce
    if (!cm->agree_buf) HIP_OK(hipMalloc((void**)&cm->agree_buf, sizeof(host)), "hals sweep: agreement buffer");
    float* dev = cm->agree_buf;
    HIP_OK(hipMemcpyAsync(dev, host, sizeof(host), hipMemcpyHostToDevice, st), "hals sweep: agreement");
    HIP_OK(hipStreamSynchronize(st), "hals sweep: agreement");
    if (int rc = allreduce_f32(cm, G_WORLD, dev, DNMF_DIRECT_MAX_RANKS, st)) return rc;
    HIP_OK(hipMemcpyAsync(host, dev, sizeof(host), hipMemcpyDeviceToHost, st), "hals sweep: agreement");
    HIP_OK(hipStreamSynchronize(st), "hals sweep: agreement");
    int total = 0, first = 0;
    bool all = true;
    for (int q = 0; q < cm->nranks; ++q) {
        if (!(host[q] >= 1.0f)) all = false;
        if (q < cm->rank) first += (int)host[q];
        total += (int)host[q];
    }
    cm->xs_ok = all && total <= HALS_MAX_WG;
    cm->xs_first = first; cm->xs_total = total;
    return DNMF_OK;
}

int hals_sweep_exchanged(dnmf_comm* cm, float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, float eps,
                         double* ss2, void* ws, size_t ws_bytes, void* stream, int column_sweep = 0) {
    hipStream_t st = S(stream);
    // ONE persistent launch when the direct peer regions are up (dnmf_comm_set_direct) and every rank can keep its rows resident:
    // the column partials cross the ranks through the slot slabs in the exported regions (csrc/dnmf_hals.h, HalsPeers)
    // (`column_sweep`: the caller asked for the column launches -- every rank passes the same value, params.hals_sweep = 'columns')
    if (!column_sweep && cm->direct_on && !cm->null_exchange && k <= DNMF_TUNED_MAX_K && cm->nranks <= DNMF_DIRECT_MAX_RANKS) {
        if (cm->xs_seen_epoch != cm->xs_epoch || cm->xs_k != k || cm->xs_pers != dnmf_persistent_on_()) {
            if (int rc = hals_xsweep_agree(cm, W, m, k, ldw, AH, ldah, G, eps, ws, ws_bytes, st)) return rc;
        } else if (cm->xs_m != m)                                   // a rank cannot re-open the agreement on its own
            return fail(DNMF_EINVAL, "hals sweep: the row count of rank %d changed (%ld -> %ld) inside one fit: call dnmf_comm_fit_begin on every rank "
                                     "before the first step on new data", cm->rank, cm->xs_m, m);
        if (cm->xs_ok) {
            HalsPeers pe{};
            pe.P = cm->nranks; pe.rank = cm->rank; pe.first = cm->xs_first; pe.total = cm->xs_total; pe.patience = cm->direct_patience;
            const size_t off = direct_hals_off(cm->direct_cap) + (size_t)(cm->hals_seq & 1) * HALS_SLAB_BYTES;
            for (int q = 0; q < cm->nranks; ++q) pe.slab[q] = reinterpret_cast<unsigned long long*>(cm->direct_peer[q] + off);
            ++cm->hals_seq;
            const int rc = dnmf_hals_sweep_w_peers_(W, m, k, ldw, AH, ldah, G, eps, ws, ws_bytes, &pe, nullptr, stream);
            if (rc != 1) return rc;
            return fail(DNMF_EINVAL, "hals sweep: the persistent sweep the ranks agreed on does not apply on rank %d", cm->rank);
        }
    }
    HIP_OK(hipMemsetAsync(ss2, 0, (size_t)k * sizeof(double), st), "hals sweep: memset");
    int rc;
    for (int kk = 0; kk < k; ++kk) {
        if ((rc = dnmf_hals_w_col(W, m, k, ldw, AH, ldah, G, kk, kk > 0 ? ss2 + kk - 1 : nullptr, eps, ss2 + kk, stream))) return rc;
        if ((rc = allreduce_f64(cm, G_WORLD, ss2 + kk, 1, st))) return rc;
    }
    return dnmf_hals_w_scale(W, m, ldw, k - 1, ss2 + k - 1, stream);
}
// ... and with local norms: the persistent sweep (or its column form on request)
int hals_sweep_local(float* W, long m, int k, long ldw, const float* AH, long ldah, const float* G, float eps, int column_sweep,
                     double* ss2, void* ws, size_t kws, void* stream) {
    if (column_sweep) {
        HIP_OK(hipMemsetAsync(ss2, 0, (size_t)k * sizeof(double), S(stream)), "hals sweep: memset");
        return dnmf_hals_update_w(W, m, k, ldw, AH, ldah, G, eps, ss2, stream);
    }
    return dnmf_hals_sweep_w(W, m, k, ldw, AH, ldah, G, eps, ws, kws, stream);
}

// equal blocks of `count` floats: recv[q * count ...] = member q's send (MPI Allgather, dist_nmf.py:163-165, :195-197)
int allgather_f32(dnmf_comm* cm, int g, const float* send, float* recv, size_t count, hipStream_t st) {
    ncclComm_t c;
    const int r = resolve(cm, g, &c, "allgather");
    if (r < 0) return r;
    if (r) {                                                       // one member (or a stubbed exchange): its own block
        HIP_OK(hipMemcpyAsync(recv, send, count * sizeof(float), hipMemcpyDeviceToDevice, st), "allgather: copy");
        return DNMF_OK;
    }
    if (cm->hook) return cm->hook(cm->hook_user, OP_ALLGATHER, g, send, recv, count, st) ? fail(DNMF_ECOMM, "allgather: the host collective failed") : DNMF_OK;
    NCCL_OK(rccl()->AllGather(send, recv, count, ncclFloat32, c, st), "allgather");
    if (cm->emulated)      // (the other members' blocks: copies of this one, so that the step runs on finite numbers)
        for (int q = 1; q < group_size(cm, g); ++q)
            HIP_OK(hipMemcpyAsync(recv + (size_t)q * count, recv, count * sizeof(float), hipMemcpyDeviceToDevice, st), "allgather: copy");
    return DNMF_OK;
}

// SUM reduce-scatter of members x count floats: recv = this member's block of the sum (MPI Reduce_scatter, :169, :202)
int reduce_scatter_f32(dnmf_comm* cm, int g, const float* send, float* recv, size_t count, int member, hipStream_t st) {
    ncclComm_t c;
    const int r = resolve(cm, g, &c, "reduce_scatter");
    if (r < 0) return r;
    if (r) {
        const size_t at = group_size(cm, g) == 1 ? 0 : (size_t)member * count;
        HIP_OK(hipMemcpyAsync(recv, send + at, count * sizeof(float), hipMemcpyDeviceToDevice, st), "reduce_scatter: copy");
        return DNMF_OK;
    }
    if (cm->hook) return cm->hook(cm->hook_user, OP_REDUCE_SCATTER, g, send, recv, count, st) ? fail(DNMF_ECOMM, "reduce_scatter: the host collective failed") : DNMF_OK;
    NCCL_OK(rccl()->ReduceScatter(cm->emulated ? send + (size_t)member * count : send, recv, count, ncclFloat32, ncclSum, c, st), "reduce_scatter");
    return DNMF_OK;
}

// ---- 2D grids: slices of a block over the members of a sub-communicator follow the reference's partition rule (utils.py:36-41):
// the first total % p members hold one item more
struct Split {
    int p; long base, rem;
    long count(int q) const { return base + (q < rem ? 1 : 0); }
    long start(int q) const { return (long)q * base + std::min<long>(q, rem); }
    long maxc() const { return base + (rem > 0 ? 1 : 0); }
    bool equal() const { return rem == 0; }
};
inline Split split_of(long total, int p) { return Split{p, total / p, total % p}; }

// workspace of the 2D steps: kernel scratch | G | the exchange buffers (sized for the padded, ragged form)
struct Ws2d { size_t g_off, hs_off, hj_off, v_off, vp_off, sw_off, wp_off, wg_off, wi_off, y_off, yb_off, sh_off, x_off, total; };
Ws2d ws2d_layout(long m_l, long n_l, int k, int p_r, int p_c) {
    const int kp = 32 * kt_of(k);
    const Split ws = split_of(m_l, p_c), hs = split_of(n_l, p_r);
    const long mw = ws.maxc(), nh = hs.maxc();
    size_t kws = std::max(dnmf_ws_bytes(m_l, n_l, k), dnmf_ws_bytes(m_l, std::max<long>(1, hs.base), k));
    kws = std::max(kws, dnmf_ws_bytes(ws.maxc(), k, k));            // (the HALS W sweep on the rank's slice)
    if (hs.equal() && hs.base % 32 == 0) kws = std::max(kws, dnmf_ws_bytes_hblocks(m_l, n_l, k, hs.base));
    Ws2d w;
    size_t o = align256(kws);
    auto take = [&](size_t floats) { const size_t at = o; o += align256(floats * sizeof(float)); return at; };
    w.g_off = take((size_t)kp * kp);
    w.hs_off = take((size_t)p_r * k * nh + (size_t)k * nh);     // gathered H slices [p_r][k nh] (+ this rank's padded send block)
    w.hj_off = take((size_t)k * n_l);                           // H_j assembled (ragged / narrow slices)
    w.v_off = take((size_t)m_l * k);                            // A H^T or U H^T of the block
    w.vp_off = take((size_t)p_c * mw * k);                      // ... laid out at the pitch of the largest W slice
    w.sw_off = take((size_t)mw * k);                            // this rank's slice of the reduced product
    w.wp_off = take((size_t)mw * k);                            // this rank's W slice, padded
    w.wg_off = take((size_t)p_c * mw * k);                      // gathered W slices
    w.wi_off = take((size_t)m_l * k);                           // W_i assembled
    w.y_off = take((size_t)k * n_l);                            // full-width H-phase product
    w.yb_off = take((size_t)p_r * k * nh);                      // ... as member blocks
    w.sh_off = take((size_t)k * nh);
    w.x_off = take(128 + 2 * DNMF_MAX_K);                          // factor sums; k doubles for the HALS norms
    w.total = o;
    return w;
}

int check_2d(const char* what, const dnmf_comm* c, long m_l, long n_l, long m_w, long n_h, long ldw, long ldh, int k) {
    if (c->p_r * c->p_c != c->nranks) return fail(DNMF_EINVAL, "%s: communicator grid %d x %d", what, c->p_r, c->p_c);
    const int i = c->rank / c->p_c, j = c->rank % c->p_c;
    const Split ws = split_of(m_l, c->p_c), hs = split_of(n_l, c->p_r);
    if (m_l < c->p_c || n_l < c->p_r || m_w != ws.count(j) || n_h != hs.count(i) || ldw != k || ldh != n_h)
        return fail(DNMF_EINVAL, "%s: factor slices off the partition rule or strided (A %ld x %ld on %d x %d at (%d, %d): W slice %ld x %d "
                    "ld %ld, expected %ld rows; H slice %d x %ld ld %ld, expected %ld columns): use the host choreography",
                    what, m_l, n_l, c->p_r, c->p_c, i, j, m_w, k, ldw, ws.count(j), k, n_h, ldh, hs.count(i));
    return DNMF_OK;
}

#define COPY2D(dst, dpitch, src, spitch, width, height, what)                                                                   \
    HIP_OK(hipMemcpy2DAsync((dst), (size_t)(dpitch) * sizeof(float), (src), (size_t)(spitch) * sizeof(float),                   \
                            (size_t)(width) * sizeof(float), (size_t)(height), hipMemcpyDeviceToDevice, st), what)
#define COPY1D(dst, src, count, what) \
    HIP_OK(hipMemcpyAsync((dst), (src), (size_t)(count) * sizeof(float), hipMemcpyDeviceToDevice, st), what)

// The exchanges of a 2D step, shared by the Frobenius and the KL form.  How H_j reaches the kernels mirrors gather_H /
// _product_scattered_to_H of pydnmfk_amd/dist_nmf.py exactly (the same kernels on the same operand shapes: same bits):
//   blocked  equal slices of whole 32-column tiles, p_r > 1: the allgather's receive buffer [p_r][k][n_h] IS the operand
//   else     H_j (k x n_l) is assembled from the (padded) slices
//   sliced   equal slices of whole 16-byte vectors: the H-phase product is formed slice by slice into the reduce-scatter's
//            send buffer; else one full-width product, cut into member blocks at the pitch of the largest slice
struct Grid2d {
    dnmf_comm* c; hipStream_t st; int k, i, j; long m_l, n_l, m_w, n_h; Split ws, hs; bool blocked, sliced;
    char* base; Ws2d L;
    float* at(size_t off) const { return (float*)(base + off); }

    // H_j from the row group: returns the operand pointer (*hb = leading dimension, or the block width when blocked)
    int gather_h(const float* H, const float** Hop, long* hb) {
        float* Hs = at(L.hs_off);
        const long nh = hs.maxc();
        int rc;
        if (c->p_r == 1) { *Hop = H; *hb = n_l; return DNMF_OK; }                                        // (one member: its slice is H_j)
        if (hs.equal()) {
            if ((rc = allgather_f32(c, G_ROW, H, Hs, (size_t)k * n_h, st))) return rc;
            if (blocked) { *Hop = Hs; *hb = n_h; return DNMF_OK; }
        } else {                                                                                          // ragged: blocks padded to the largest
            float* mine = Hs + (size_t)c->p_r * k * nh;
            COPY1D(mine, H, (size_t)k * n_h, "gather_h: pad");
            if ((rc = allgather_f32(c, G_ROW, mine, Hs, (size_t)k * nh, st))) return rc;
        }
        float* Hj = at(L.hj_off);
        const size_t pitch = hs.equal() ? (size_t)k * n_h : (size_t)k * nh;
        for (int q = 0; q < c->p_r; ++q)
            COPY2D(Hj + hs.start(q), n_l, Hs + q * pitch, hs.count(q), hs.count(q), k, "gather_h: assemble");
        *Hop = Hj; *hb = n_l;
        return DNMF_OK;
    }
    // W_i (m_l x k) from the column group
    int gather_w(const float* W, const float** Wi_out) {
        int rc;
        if (c->p_c == 1) { *Wi_out = W; return DNMF_OK; }
        float* Wi = at(L.wi_off);
        if (ws.equal()) {
            if ((rc = allgather_f32(c, G_COL, W, Wi, (size_t)m_w * k, st))) return rc;
        } else {
            float *Wp = at(L.wp_off), *Wg = at(L.wg_off);
            const size_t pitch = (size_t)ws.maxc() * k;
            COPY1D(Wp, W, (size_t)m_w * k, "gather_w: pad");
            if ((rc = allgather_f32(c, G_COL, Wp, Wg, pitch, st))) return rc;
            for (int q = 0; q < c->p_c; ++q) COPY1D(Wi + ws.start(q) * k, Wg + q * pitch, (size_t)ws.count(q) * k, "gather_w: assemble");
        }
        *Wi_out = Wi;
        return DNMF_OK;
    }
    // reduce-scatter of V (m_l x k) over the column group -> this rank's m_w x k slice
    int scatter_to_w(const float* V, const float** out) {
        int rc;
        if (c->p_c == 1) { *out = V; return DNMF_OK; }
        float* Sw = at(L.sw_off);
        if (ws.equal()) {
            if ((rc = reduce_scatter_f32(c, G_COL, V, Sw, (size_t)m_w * k, j, st))) return rc;
        } else {
            float* Vp = at(L.vp_off);
            const size_t pitch = (size_t)ws.maxc() * k;
            for (int q = 0; q < c->p_c; ++q) COPY1D(Vp + q * pitch, V + ws.start(q) * k, (size_t)ws.count(q) * k, "scatter_to_w: pad");
            if ((rc = reduce_scatter_f32(c, G_COL, Vp, Sw, pitch, j, st))) return rc;
        }
        *out = Sw;
        return DNMF_OK;
    }
    // reduce-scatter of the H-phase product over the row group -> this rank's k x n_h slice.  sliced: Yb holds the member
    // blocks already; else Y (k x n_l) is cut into blocks at the pitch of the largest slice first.
    int scatter_to_h(const float* Y, const float** out) {
        int rc;
        float *Yb = at(L.yb_off), *Sh = at(L.sh_off);
        if (c->p_r == 1) { *out = sliced ? Yb : Y; return DNMF_OK; }
        size_t pitch = (size_t)k * n_h;
        if (!sliced) {
            pitch = (size_t)k * hs.maxc();
            for (int q = 0; q < c->p_r; ++q)
                COPY2D(Yb + q * pitch, hs.count(q), Y + hs.start(q), n_l, hs.count(q), k, "scatter_to_h: blocks");
        }
        if ((rc = reduce_scatter_f32(c, G_ROW, Yb, Sh, pitch, i, st))) return rc;
        *out = Sh;
        return DNMF_OK;
    }
};

// the entry points that touch A, for fp32 and for bf16-stored A (params.precision = 'bfloat16': Frobenius mu / hals)
template <typename TA> struct AOps;
template <> struct AOps<float> {
    static constexpr bool f32 = true;
    static int aht(const float* A, long m, long n, long lda, const float* H, int k, long ldh, float* o, long ldo, void* s) { return dnmf_aht(A, m, n, lda, H, k, ldh, o, ldo, s); }
    static int wta(const float* A, long m, long n, long lda, const float* W, int k, long ldw, float* o, long ldo, void* ws, size_t wb, void* s) { return dnmf_wta(A, m, n, lda, W, k, ldw, o, ldo, ws, wb, s); }
    static int wta_gram(const float* A, long m, long n, long lda, const float* W, int k, long ldw, float* o, long ldo, float* G, void* ws, size_t wb, void* s) { return dnmf_wta_gram(A, m, n, lda, W, k, ldw, o, ldo, G, ws, wb, s); }
    static int ahtw(const float* A, long m, long n, long lda, const float* H, int k, long ldh, const float* G, float* W, long ldw, float eps, void* s) { return dnmf_aht_update_w(A, m, n, lda, H, k, ldh, G, W, ldw, eps, s); }
};
template <> struct AOps<bf16_t> {
    static constexpr bool f32 = false;
    static int aht(const bf16_t* A, long m, long n, long lda, const float* H, int k, long ldh, float* o, long ldo, void* s) { return dnmf_aht_bf16a(A, m, n, lda, H, k, ldh, o, ldo, s); }
    static int wta(const bf16_t* A, long m, long n, long lda, const float* W, int k, long ldw, float* o, long ldo, void* ws, size_t wb, void* s) { return dnmf_wta_bf16a(A, m, n, lda, W, k, ldw, o, ldo, ws, wb, s); }
    static int wta_gram(const bf16_t* A, long m, long n, long lda, const float* W, int k, long ldw, float* o, long ldo, float* G, void* ws, size_t wb, void* s) { return dnmf_wta_gram_bf16a(A, m, n, lda, W, k, ldw, o, ldo, G, ws, wb, s); }
    static int ahtw(const bf16_t* A, long m, long n, long lda, const float* H, int k, long ldh, const float* G, float* W, long ldw, float eps, void* s) { return dnmf_aht_update_w_bf16a(A, m, n, lda, H, k, ldh, G, W, ldw, eps, s); }
};

int grid2d_init(Grid2d& g, const char* what, dnmf_comm* c, long m_l, long n_l, long m_w, long n_h, long ldw, long ldh, int k,
                void* ws, size_t ws_bytes, void* stream, bool f32 = true) {
    int rc;
    if ((rc = check_2d(what, c, m_l, n_l, m_w, n_h, ldw, ldh, k))) return rc;
    g.c = c; g.st = S(stream); g.k = k; g.i = c->rank / c->p_c; g.j = c->rank % c->p_c;
    g.m_l = m_l; g.n_l = n_l; g.m_w = m_w; g.n_h = n_h;
    g.ws = split_of(m_l, c->p_c); g.hs = split_of(n_l, c->p_r);
    g.blocked = f32 && c->p_r > 1 && g.hs.equal() && n_h % 32 == 0;   // (bf16-stored A has no column-block form of A H^T)
    g.sliced = g.hs.equal() && n_h % 4 == 0;
    g.L = ws2d_layout(m_l, n_l, k, c->p_r, c->p_c);
    if (ws_bytes < g.L.total) return fail(DNMF_EWS, "%s: workspace %zu < %zu", what, ws_bytes, g.L.total);
    g.base = (char*)ws;
    return DNMF_OK;
}

// workspace of the 1D steps: [kernel scratch of dnmf_ws_bytes | G (KP x KP) | packed exchange buffer]
struct Ws1d { size_t g_off, x_off, total; };
Ws1d ws1d_layout(long m_l, long n_l, int k) {
    const int kp = 32 * kt_of(k);
    Ws1d w;
    size_t kws = std::max(dnmf_ws_bytes(m_l, n_l, k), dnmf_ws_bytes(m_l, k, k));   // kernel scratch: the whole block (and the HALS sweep's), every chunk width of the
    for (int nch = 2; nch <= MAX_CHUNKS; ++nch) {                 // overlapped H phase (the chunking of W^T A depends on the width)
        const long cw = (cdiv(n_l, nch) + 63) / 64 * 64;
        if (cw >= n_l) continue;
        kws = std::max(kws, dnmf_ws_bytes(m_l, cw, k));
        if (n_l % cw) kws = std::max(kws, dnmf_ws_bytes(m_l, n_l % cw, k));
    }
    w.g_off = align256(kws);
    w.x_off = w.g_off + align256((size_t)kp * kp * sizeof(float));
    const size_t big = std::max((size_t)k * n_l, (size_t)m_l * k);
    // one packed message [product | pad | KP x KP] or up to MAX_CHUNKS chunk messages, each padded to 64 floats
    w.total = w.x_off + align256((pad64(big) + (size_t)MAX_CHUNKS * 64 + (size_t)kp * kp + 128 + 2 * DNMF_MAX_K) * sizeof(float));   // (+ k doubles: HALS norms)
    return w;
}

}  // namespace

extern "C" {

int dnmf_comm_unique_id(void* id_out) {
    REQUIRE(id_out, "comm_unique_id: null pointer");
    Rccl* r = rccl();
    if (!r) return fail(DNMF_ECOMM, "comm_unique_id: no RCCL library found (librccl.so.1)");
    ncclUniqueId id;
    NCCL_OK(r->GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_out, &id, DNMF_UNIQUE_ID_BYTES);
    return DNMF_OK;
}

int dnmf_comm_create(const void* unique_id, int nranks, int rank, int p_r, int p_c, dnmf_comm_t** out) {
    REQUIRE(unique_id && out && nranks >= 1 && rank >= 0 && rank < nranks && p_r >= 1 && p_c >= 1 && p_r * p_c == nranks,
            "comm_create: bad arguments (nranks %d, rank %d, grid %d x %d)", nranks, rank, p_r, p_c);
    static_assert(DNMF_UNIQUE_ID_BYTES == sizeof(ncclUniqueId), "unique id size");
    Rccl* r = rccl();
    if (!r) return fail(DNMF_ECOMM, "comm_create: no RCCL library found (librccl.so.1)");
    dnmf_comm* c = new dnmf_comm();
    c->nranks = nranks; c->rank = rank; c->p_r = p_r; c->p_c = p_c;
    if (p_r > 1 && p_c > 1 && !r->CommSplit) {
        delete c;
        return fail(DNMF_ECOMM, "comm_create: this RCCL (%s) has no ncclCommSplit: a %d x %d grid needs it (1D grids do not)", r->origin, p_r, p_c);
    }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t e = r->CommInitRank(&c->world, nranks, id, rank);
    if (e != ncclSuccess) { delete c; return nccl_fail("ncclCommInitRank", e); }
    if (p_r > 1 && p_c > 1) {            // rank = i * p_c + j (dist_comm.py:22, reorder = False)
        const int i = rank / p_c, j = rank % p_c;
        e = r->CommSplit(c->world, j, i, &c->row, nullptr);                    // same grid column j, ordered by i (dist_comm.py:25-37)
        if (e == ncclSuccess) e = r->CommSplit(c->world, i, j, &c->col, nullptr);   // same grid row i, ordered by j (dist_comm.py:39-51)
        if (e != ncclSuccess) { dnmf_comm_destroy(c); return nccl_fail("ncclCommSplit", e); }
    }
    if (hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking) != hipSuccess) { dnmf_comm_destroy(c); return fail(DNMF_EHIP, "comm_create: stream"); }
    for (int q = 0; q < MAX_CHUNKS; ++q)
        if (hipEventCreateWithFlags(&c->ready[q], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->done[q], hipEventDisableTiming) != hipSuccess) { dnmf_comm_destroy(c); return fail(DNMF_EHIP, "comm_create: events"); }
    *out = c;
    return DNMF_OK;
}

// MEASUREMENT ONLY (bench.py --emulate-ranks on a one-GPU box): member `rank` of a p_r x p_c grid whose collectives -- world, row
// and column groups alike -- are issued for real on a ONE-rank RCCL communicator: launch and stream-ordering costs are real, there
// is no wire, and the buffers have the sizes of the real grid (the other members' allgather blocks are copies of this member's).
// The step entry points then run exactly as on the grid; their results are not a factorisation of anything.
int dnmf_comm_create_emulated(int p_r, int p_c, int rank, dnmf_comm_t** out) {
    REQUIRE(out && p_r >= 1 && p_c >= 1 && rank >= 0 && rank < p_r * p_c, "comm_create_emulated: bad arguments (grid %d x %d, member %d)", p_r, p_c, rank);
    Rccl* r = rccl();
    if (!r) return fail(DNMF_ECOMM, "comm_create_emulated: no RCCL library found (librccl.so.1)");
    dnmf_comm* c = new dnmf_comm();
    c->nranks = p_r * p_c; c->rank = rank; c->p_r = p_r; c->p_c = p_c; c->emulated = 1;
    ncclUniqueId id;
    ncclResult_t e = r->GetUniqueId(&id);
    if (e == ncclSuccess) e = r->CommInitRank(&c->world, 1, id, 0);
    if (e != ncclSuccess) { delete c; return nccl_fail("comm_create_emulated", e); }
    if (hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking) != hipSuccess) { dnmf_comm_destroy(c); return fail(DNMF_EHIP, "comm_create_emulated: stream"); }
    for (int q = 0; q < MAX_CHUNKS; ++q)
        if (hipEventCreateWithFlags(&c->ready[q], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->done[q], hipEventDisableTiming) != hipSuccess) { dnmf_comm_destroy(c); return fail(DNMF_EHIP, "comm_create_emulated: events"); }
    *out = c;
    return DNMF_OK;
}

int dnmf_comm_rccl_version(int* version, char* origin, size_t origin_bytes) {
    Rccl* r = rccl();
    if (!r) return fail(DNMF_ECOMM, "comm_rccl_version: no RCCL library found (librccl.so.1)");
    int v = 0;
    if (r->GetVersion && r->GetVersion(&v) != ncclSuccess) v = 0;
    if (version) *version = v;
    if (origin && origin_bytes) snprintf(origin, origin_bytes, "%s", r->origin);
    return DNMF_OK;
}

int dnmf_comm_create_hosted(int nranks, int rank, int p_r, int p_c, dnmf_collective_fn fn, void* user, dnmf_comm_t** out) {
    REQUIRE(fn && out && nranks >= 1 && rank >= 0 && rank < nranks && p_r >= 1 && p_c >= 1 && p_r * p_c == nranks,
            "comm_create_hosted: bad arguments (nranks %d, rank %d, grid %d x %d)", nranks, rank, p_r, p_c);
    dnmf_comm* c = new dnmf_comm();
    c->nranks = nranks; c->rank = rank; c->p_r = p_r; c->p_c = p_c;
    c->hook = fn; c->hook_user = user;
    if (hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking) != hipSuccess) { dnmf_comm_destroy(c); return fail(DNMF_EHIP, "comm_create_hosted: stream"); }
    for (int q = 0; q < MAX_CHUNKS; ++q)
        if (hipEventCreateWithFlags(&c->ready[q], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->done[q], hipEventDisableTiming) != hipSuccess) { dnmf_comm_destroy(c); return fail(DNMF_EHIP, "comm_create_hosted: events"); }
    *out = c;
    return DNMF_OK;
}

int dnmf_comm_direct_init(dnmf_comm_t* c, size_t max_floats, void* handle_out) {
    REQUIRE(c && handle_out && max_floats >= 2 && c->nranks <= DNMF_DIRECT_MAX_RANKS, "comm_direct_init: bad arguments (%d ranks, at most %d)",
            c ? c->nranks : 0, DNMF_DIRECT_MAX_RANKS);
    REQUIRE(!c->direct_peer[c->rank], "comm_direct_init: already set up");
    static_assert(sizeof(hipIpcMemHandle_t) + 2 * sizeof(unsigned long long) <= DNMF_DIRECT_HANDLE_BYTES, "IPC handle size");
    const size_t cap = (max_floats + 63) / 64 * 64;
    const size_t bytes = direct_hals_off(cap) + 2 * HALS_SLAB_BYTES;
    void* region = nullptr;
    // UNCACHED (fine-grained) device memory: flags and data are read by peers while kernels of this GPU run.  No fallback to an
    // ordinary allocation: its coherence across agents is not what the protocol was validated on -- the rank fails here, the
    // host's agreement round (NativeComm.enable_direct) keeps every rank on the communicator's own allreduce.
    hipError_t e = hipExtMallocWithFlags(&region, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(DNMF_EHIP, "comm_direct_init: uncached allocation of %zu bytes: %s", bytes, hipGetErrorString(e)); }
    if ((e = hipMemset(region, 0, bytes)) != hipSuccess ||                                         // flags, buffers; the HALS slot slabs: "empty"
        (e = hipMemset((char*)region + direct_hals_off(cap), 0xff, 2 * HALS_SLAB_BYTES)) != hipSuccess) {
        (void)hipFree(region);
        return fail(DNMF_EHIP, "comm_direct_init: memset: %s", hipGetErrorString(e));
    }
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, region);
    if (e != hipSuccess) { (void)hipFree(region); return fail(DNMF_EHIP, "comm_direct_init: hipIpcGetMemHandle: %s", hipGetErrorString(e)); }
    // the exchanged blob: [IPC handle | capacity in floats | magic] -- peers compute offsets into EVERY region from one capacity,
    // so dnmf_comm_direct_connect requires all of them equal
    memset(handle_out, 0, DNMF_DIRECT_HANDLE_BYTES);
    memcpy(handle_out, &h, sizeof(h));
    const unsigned long long tail[2] = {(unsigned long long)cap, DIRECT_MAGIC};
    memcpy((char*)handle_out + sizeof(h), tail, sizeof(tail));
    c->direct_peer[c->rank] = (char*)region;
    c->direct_cap = cap;
    return DNMF_OK;
}

int dnmf_comm_direct_connect(dnmf_comm_t* c, const void* handles) {
    REQUIRE(c && handles && c->direct_peer[c->rank], "comm_direct_connect: call dnmf_comm_direct_init first");
    for (int q = 0; q < c->nranks; ++q) {       // every rank sees every blob: a mismatch anywhere fails the connect everywhere
        unsigned long long tail[2];
        memcpy(tail, (const char*)handles + (size_t)q * DNMF_DIRECT_HANDLE_BYTES + sizeof(hipIpcMemHandle_t), sizeof(tail));
        REQUIRE(tail[1] == DIRECT_MAGIC, "comm_direct_connect: rank %d did not initialise its region", q);
        REQUIRE(tail[0] == (unsigned long long)c->direct_cap, "comm_direct_connect: rank %d sized its region for %llu floats, this rank for %zu "
                "(every rank must pass the same max_floats to dnmf_comm_direct_init)", q, tail[0], c->direct_cap);
    }
    for (int q = 0; q < c->nranks; ++q) {
        if (q == c->rank || c->direct_peer[q]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)q * DNMF_DIRECT_HANDLE_BYTES, sizeof(h));
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return fail(DNMF_EHIP, "comm_direct_connect: hipIpcOpenMemHandle(rank %d): %s", q, hipGetErrorString(e));
        c->direct_peer[q] = (char*)p;
    }
    return DNMF_OK;
}

// undo dnmf_comm_direct_init / _connect (a set-up that failed on some rank, or the end of the communicator): unmap the peers,
// free the own region.  The caller makes sure no peer still reads it (NativeComm.close runs a host barrier first).
static void direct_teardown(dnmf_comm* c) {
    c->direct_on = 0;
    for (int q = 0; q < c->nranks && q < DNMF_DIRECT_MAX_RANKS; ++q) {
        if (!c->direct_peer[q]) continue;
        if (q == c->rank) (void)hipFree(c->direct_peer[q]);
        else (void)hipIpcCloseMemHandle(c->direct_peer[q]);
        c->direct_peer[q] = nullptr;
    }
    c->direct_cap = 0;
}

int dnmf_comm_direct_teardown(dnmf_comm_t* c) {
    REQUIRE(c, "comm_direct_teardown: null communicator");
    (void)hipDeviceSynchronize();
    direct_teardown(c);
    return DNMF_OK;
}

int dnmf_comm_set_direct_timeout(dnmf_comm_t* c, double seconds) {
    REQUIRE(c && seconds >= 0.001 && seconds <= 86400.0, "comm_set_direct_timeout: 0.001 .. 86400 seconds");
    c->direct_patience = (unsigned long long)(seconds * 1e8);
    return DNMF_OK;
}

int dnmf_comm_hals_xsweeps(dnmf_comm_t* c, unsigned long long* count) {
    REQUIRE(c && count, "comm_hals_xsweeps: null pointer");
    *count = c->hals_seq;
    return DNMF_OK;
}

int dnmf_comm_set_direct(dnmf_comm_t* c, int on) {
    REQUIRE(c, "comm_set_direct: null communicator");
    if (on) for (int q = 0; q < c->nranks; ++q) REQUIRE(c->direct_peer[q], "comm_set_direct: rank %d is not connected", q);
    c->direct_on = on ? 1 : 0;
    return DNMF_OK;
}

int dnmf_comm_direct_status(dnmf_comm_t* c, int* timed_out) {
    REQUIRE(c && timed_out && c->direct_peer[c->rank], "comm_direct_status: not set up");
    unsigned long long w = 0;
    HIP_OK(hipMemcpy(&w, c->direct_peer[c->rank] + 2048, sizeof(w), hipMemcpyDeviceToHost), "comm_direct_status: copy");
    *timed_out = w != 0;
    return DNMF_OK;
}

int dnmf_comm_allreduce_direct(dnmf_comm_t* c, float* buf, size_t count, void* stream) {
    REQUIRE(c && buf, "comm_allreduce_direct: bad arguments");
    for (int q = 0; q < c->nranks; ++q) REQUIRE(c->direct_peer[q], "comm_allreduce_direct: rank %d is not connected", q);
    if (c->nranks == 1) return DNMF_OK;
    return direct_allreduce(c, buf, count, S(stream));
}

int dnmf_comm_allreduce_direct_f64(dnmf_comm_t* c, double* buf, size_t count, void* stream) {
    REQUIRE(c && buf && count >= 1 && count <= DIRECT_SMALL_MAX, "comm_allreduce_direct_f64: 1 <= count <= %d doubles", DIRECT_SMALL_MAX);
    for (int q = 0; q < c->nranks; ++q) REQUIRE(c->direct_peer[q], "comm_allreduce_direct_f64: rank %d is not connected", q);
    if (c->nranks == 1) return DNMF_OK;
    return direct_allreduce_small_f64(c, buf, count, S(stream));
}

int dnmf_comm_destroy(dnmf_comm_t* c) {
    if (!c) return DNMF_OK;
    direct_teardown(c);
    Rccl* r = rccl();
    for (int q = 0; q < MAX_CHUNKS; ++q) {
        if (c->ready[q]) (void)hipEventDestroy(c->ready[q]);
        if (c->done[q]) (void)hipEventDestroy(c->done[q]);
    }
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    if (c->agree_buf) (void)hipFree(c->agree_buf);
    if (r) {
        if (c->row) r->CommDestroy(c->row);
        if (c->col) r->CommDestroy(c->col);
        if (c->world) r->CommDestroy(c->world);
    }
    delete c;
    return DNMF_OK;
}

int dnmf_comm_fit_begin(dnmf_comm_t* c) {
    REQUIRE(c, "comm_fit_begin: null communicator");
    ++c->xs_epoch;
    return DNMF_OK;
}

int dnmf_comm_set_overlap_chunks(dnmf_comm_t* c, int chunks) {
    REQUIRE(c && chunks >= 1 && chunks <= MAX_CHUNKS, "comm_set_overlap_chunks: 1 <= chunks <= %d", MAX_CHUNKS);
    c->overlap_chunks = chunks;
    return DNMF_OK;
}

int dnmf_comm_set_always_exchange(dnmf_comm_t* c, int on) {
    REQUIRE(c, "comm_set_always_exchange: null communicator");
    c->always = on != 0;
    return DNMF_OK;
}

int dnmf_comm_set_null_exchange(dnmf_comm_t* c, int on) {
    REQUIRE(c, "comm_set_null_exchange: null communicator");
    c->null_exchange = on != 0;
    return DNMF_OK;
}

int dnmf_comm_info(const dnmf_comm_t* c, int* nranks, int* rank, int* p_r, int* p_c) {
    REQUIRE(c, "comm_info: null communicator");
    if (nranks) *nranks = c->nranks;
    if (rank) *rank = c->rank;
    if (p_r) *p_r = c->p_r;
    if (p_c) *p_c = c->p_c;
    return DNMF_OK;
}

int dnmf_comm_allreduce(dnmf_comm_t* c, float* buf, size_t count, int group, void* stream) {
    REQUIRE(c && buf && group >= 0 && group <= 2, "comm_allreduce: bad arguments");
    return allreduce_f32(c, group, buf, count, S(stream));
}

size_t dnmf_ws_bytes_1d(long m_l, long n_l, int k) {
    if (kt_of(k) < 0 || m_l < 1 || n_l < 1) return 0;
    return ws1d_layout(m_l, n_l, k).total;
}

// One MU / Frobenius step of a rank of a 1D grid (dist_nmf.py:716-771 with the exchanges of :681,:707).  p_c == 1: A and W
// are row blocks, H is replicated -- the W phase is the fused local kernel, the H phase ONE allreduce of the packed
// [W^T A (k x n_l) | pad | W^T W (KP x KP)] message (or `overlap_chunks` column chunks: W^T A of chunk c+1 is computed on
// the caller's stream while chunk c is reduced on the communicator's stream; chunk 0 carries W^T W).  p_r == 1: the mirror
// image (A and H column blocks, W replicated, [A H^T | H H^T] reduced).
}  // extern "C"
namespace {
template <typename TA>
int mu_fro_step_1d_impl(const TA* A, long m_l, long n_l, long lda, float* W, long ldw, float* H, long ldh, int k,
                        float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* c, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && ws && c && m_l >= 1 && n_l >= 1, "mu_fro_step_1d: bad arguments");
    REQUIRE(c->p_r == 1 || c->p_c == 1, "mu_fro_step_1d: a %d x %d grid is 2D", c->p_r, c->p_c);
    const int kp = 32 * kt;
    const Ws1d L = ws1d_layout(m_l, n_l, k);
    if (ws_bytes < L.total) return fail(DNMF_EWS, "mu_fro_step_1d: workspace %zu < %zu", ws_bytes, L.total);
    char* base = (char*)ws;
    const size_t kws = L.g_off;                                   // kernel scratch: [0, g_off)
    float* G = (float*)(base + L.g_off);
    float* X = (float*)(base + L.x_off);
    hipStream_t st = S(stream);
    int rc;
    if (w_update) {                                               // Fro_MU_update_W :716-732
        if (c->p_c == 1) {
            if ((rc = dnmf_gram_hht(H, k, n_l, ldh, G, ws, kws, stream))) return rc;
            if ((rc = AOps<TA>::ahtw(A, m_l, n_l, lda, H, k, ldh, G, W, ldw, eps, stream))) return rc;
        } else {
            const size_t off = pad64((size_t)m_l * k);
            float* AH = X; float* Gx = X + off;
            if ((rc = dnmf_gram_hht(H, k, n_l, ldh, Gx, ws, kws, stream))) return rc;
            if ((rc = AOps<TA>::aht(A, m_l, n_l, lda, H, k, ldh, AH, k, stream))) return rc;
            if ((rc = allreduce_f32(c, G_WORLD, X, off + (size_t)kp * kp, st))) return rc;
            if ((rc = dnmf_mu_update_w(W, m_l, k, ldw, AH, k, Gx, eps, stream))) return rc;
        }
    }
    const bool xr = c->p_r != 1 || c->always;                                 // which axis exchanges (always: a 1 x 1 grid acts as a row grid)
    int nch = (c->p_c == 1 && xr) ? c->overlap_chunks : 1;                     // Fro_MU_update_H :736-751
    if (nch > 1 && n_l / 64 < nch) nch = (int)std::max<long>(1, n_l / 64);
    if (nch <= 1) {
        const size_t off = pad64((size_t)k * n_l);
        float* AtW = X; float* Gx = X + off;
        if ((rc = AOps<TA>::wta_gram(A, m_l, n_l, lda, W, k, ldw, AtW, n_l, Gx, ws, kws, stream))) return rc;
        if (xr && (rc = allreduce_f32(c, G_WORLD, X, off + (size_t)kp * kp, st))) return rc;
        if ((rc = dnmf_mu_update_h(H, k, n_l, ldh, AtW, n_l, Gx, eps, clamp, stream))) return rc;
    } else {
        const long cw = (cdiv(n_l, nch) + 63) / 64 * 64;          // chunk width: ceil(n_l / nch) rounded up to 64 columns
        float* Gx = X + pad64((size_t)k * std::min(cw, n_l));
        if ((rc = dnmf_gram_wtw(W, m_l, k, ldw, Gx, ws, kws, stream))) return rc;
        size_t off = 0;
        float* chunk_ptr[MAX_CHUNKS]; long c0s[MAX_CHUNKS], c1s[MAX_CHUNKS];
        int nq = 0;
        for (long c0 = 0; c0 < n_l; c0 += cw, ++nq) {
            const long c1 = std::min(n_l, c0 + cw);
            const size_t ne = pad64((size_t)k * (c1 - c0));
            float* AtW = X + off;
            if ((rc = AOps<TA>::wta(A + c0, m_l, c1 - c0, lda, W, k, ldw, AtW, c1 - c0, ws, kws, stream))) return rc;
            const size_t span = ne + (nq == 0 ? (size_t)kp * kp : 0);          // chunk 0: [W^T A chunk | W^T W] in one message
            HIP_OK(hipEventRecord(c->ready[nq], st), "mu_fro_step_1d: event record");
            HIP_OK(hipStreamWaitEvent(c->xstream, c->ready[nq], 0), "mu_fro_step_1d: stream wait");
            if ((rc = allreduce_f32(c, G_WORLD, X + off, span, c->xstream))) return rc;
            HIP_OK(hipEventRecord(c->done[nq], c->xstream), "mu_fro_step_1d: event record");
            chunk_ptr[nq] = AtW; c0s[nq] = c0; c1s[nq] = c1;
            off += span;
        }
        for (int q = 0; q < nq; ++q) {
            HIP_OK(hipStreamWaitEvent(st, c->done[q], 0), "mu_fro_step_1d: stream wait");
            if ((rc = dnmf_mu_update_h(H + c0s[q], k, c1s[q] - c0s[q], ldh, chunk_ptr[q], c1s[q] - c0s[q], Gx, eps, clamp, stream))) return rc;
        }
    }
    if (clamp) return dnmf_clamp_min(W, m_l, k, ldw, eps, stream);
    return DNMF_OK;
}
}  // namespace
extern "C" {
int dnmf_mu_fro_step_1d(const float* A, long m_l, long n_l, long lda, float* W, long ldw, float* H, long ldh, int k,
                        float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* c, void* stream) {
    return mu_fro_step_1d_impl<float>(A, m_l, n_l, lda, W, ldw, H, ldh, k, eps, w_update, clamp, ws, ws_bytes, c, stream);
}
int dnmf_mu_fro_step_1d_bf16a(const void* A, long m_l, long n_l, long lda, float* W, long ldw, float* H, long ldh, int k,
                              float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* c, void* stream) {
    return mu_fro_step_1d_impl<bf16_t>((const bf16_t*)A, m_l, n_l, lda, W, ldw, H, ldh, k, eps, w_update, clamp, ws, ws_bytes, c, stream);
}

// One MU / KL step of a rank of a 1D grid (dist_nmf.py:813-869; exchanges :797,:707): [U H^T | rowsum(H)] is reduced when
// W is replicated (p_r == 1), [W^T U | colsum(W)] when H is (p_c == 1).
int dnmf_mu_kl_step_1d(const float* A, long m_l, long n_l, long lda, float* W, long ldw, float* H, long ldh, int k,
                       float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* c, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && ws && c && m_l >= 1 && n_l >= 1, "mu_kl_step_1d: bad arguments");
    REQUIRE(c->p_r == 1 || c->p_c == 1, "mu_kl_step_1d: a %d x %d grid is 2D", c->p_r, c->p_c);
    const Ws1d L = ws1d_layout(m_l, n_l, k);
    if (ws_bytes < L.total) return fail(DNMF_EWS, "mu_kl_step_1d: workspace %zu < %zu", ws_bytes, L.total);
    char* base = (char*)ws;
    const size_t kws = L.g_off;
    float* X = (float*)(base + L.x_off);
    hipStream_t st = S(stream);
    int rc;
    if (w_update) {                                               // KL_MU_update_W :813-830
        const size_t off = pad64((size_t)m_l * k);
        float* UHT = X; float* x2 = X + off;
        if ((rc = dnmf_rowsum(H, k, n_l, ldh, x2, stream))) return rc;
        if ((rc = dnmf_kl_uht(A, m_l, n_l, lda, W, ldw, H, ldh, k, eps, UHT, k, ws, kws, stream))) return rc;
        if (c->p_c != 1 && (rc = allreduce_f32(c, G_WORLD, X, off + (size_t)k, st))) return rc;
        if ((rc = dnmf_kl_update_w(W, m_l, k, ldw, UHT, k, x2, eps, stream))) return rc;
    }
    const size_t off = pad64((size_t)k * n_l);                    // KL_MU_update_H :832-849
    float* WTU = X; float* x1 = X + off;
    if ((rc = dnmf_colsum(W, m_l, k, ldw, x1, ws, kws, stream))) return rc;
    if ((rc = dnmf_kl_wtu(A, m_l, n_l, lda, W, ldw, H, ldh, k, eps, WTU, n_l, ws, kws, stream))) return rc;
    if ((c->p_r != 1 || c->always) && (rc = allreduce_f32(c, G_WORLD, X, off + (size_t)k, st))) return rc;
    if ((rc = dnmf_kl_update_h(H, k, n_l, ldh, WTU, n_l, x1, eps, clamp, stream))) return rc;
    if (clamp) return dnmf_clamp_min(W, m_l, k, ldw, eps, stream);
    return DNMF_OK;
}

size_t dnmf_ws_bytes_2d(long m_l, long n_l, int k, int p_r, int p_c) {
    if (kt_of(k) < 0 || p_r < 1 || p_c < 1 || m_l < p_c || n_l < p_r) return 0;
    return ws2d_layout(m_l, n_l, k, p_r, p_c).total;
}

// One MU / Frobenius step of a rank of a p_r x p_c grid (nmf_algorithms_2D.update, dist_nmf.py:207-263 with global_gram :107-117,
// AH_glob :186-205, ATW_glob :154-172): the rank holds A_ij (m_l x n_l), its slice W_ij (m_w x k) of the grid row's W_i and its
// slice H_ij (k x n_h) of the grid column's H_j.  Row group = the p_r ranks of its grid column (they share H_j's columns), column
// group = the p_c ranks of its grid row (they share W_i's rows).
}  // extern "C"
namespace {
template <typename TA>
int mu_fro_step_2d_impl(const TA* A, long m_l, long n_l, long lda, float* W, long m_w, long ldw, float* H, long n_h, long ldh,
                        int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* c, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && ws && c && m_l >= 1 && n_l >= 1 && lda >= n_l, "mu_fro_step_2d: bad arguments");
    Grid2d g;
    int rc;
    if ((rc = grid2d_init(g, "mu_fro_step_2d", c, m_l, n_l, m_w, n_h, ldw, ldh, k, ws, ws_bytes, stream, AOps<TA>::f32))) return rc;
    const int kp = 32 * kt;
    const size_t kws = g.L.g_off;
    float *G = g.at(g.L.g_off), *V = g.at(g.L.v_off), *Y = g.at(g.L.y_off), *Yb = g.at(g.L.yb_off);
    hipStream_t st = g.st;
    if (w_update) {                                                // Fro_MU_update_W :227-245
        if ((rc = dnmf_gram_hht(H, k, n_h, ldh, G, ws, kws, stream))) return rc;
        if ((rc = allreduce_f32(c, G_WORLD, G, (size_t)kp * kp, st))) return rc;                          // global_gram :114
        const float* Hop; long hb;
        if ((rc = g.gather_h(H, &Hop, &hb))) return rc;                                                   // AH_glob :195-197
        if constexpr (AOps<TA>::f32) { if (g.blocked) rc = dnmf_aht_hblocks(A, m_l, n_l, lda, Hop, hb, k, V, k, stream); else rc = AOps<TA>::aht(A, m_l, n_l, lda, Hop, k, hb, V, k, stream); }                 // :198, H as received
        else rc = AOps<TA>::aht(A, m_l, n_l, lda, Hop, k, hb, V, k, stream);
        if (rc) return rc;
        const float* AH;
        if ((rc = g.scatter_to_w(V, &AH))) return rc;                                                     // :202
        if ((rc = dnmf_mu_update_w(W, m_w, k, ldw, AH, k, G, eps, stream))) return rc;                    // :244-245
    }
    if ((rc = dnmf_gram_wtw(W, m_w, k, ldw, G, ws, kws, stream))) return rc;                              // Fro_MU_update_H :207-225
    if ((rc = allreduce_f32(c, G_WORLD, G, (size_t)kp * kp, st))) return rc;
    const float* Wi;
    if ((rc = g.gather_w(W, &Wi))) return rc;                                                             // ATW_glob :163-165
    if (g.sliced) {                                                // :166 slice by slice: member q's k x n_h block is contiguous
        for (int q = 0; q < c->p_r; ++q)
            if ((rc = AOps<TA>::wta(A + q * n_h, m_l, n_h, lda, Wi, k, k, Yb + (size_t)q * k * n_h, n_h, ws, kws, stream))) return rc;
    } else if ((rc = AOps<TA>::wta(A, m_l, n_l, lda, Wi, k, k, Y, n_l, ws, kws, stream))) return rc;
    const float* AtW;
    if ((rc = g.scatter_to_h(Y, &AtW))) return rc;                                                        // :169-171
    const long ldatw = (c->p_r == 1 && !g.sliced) ? n_l : n_h;
    if ((rc = dnmf_mu_update_h(H, k, n_h, ldh, AtW, ldatw, G, eps, clamp, stream))) return rc;            // :224-225
    if (clamp) return dnmf_clamp_min(W, m_w, k, ldw, eps, stream);
    return DNMF_OK;
}
}  // namespace
extern "C" {
int dnmf_mu_fro_step_2d(const float* A, long m_l, long n_l, long lda, float* W, long m_w, long ldw, float* H, long n_h, long ldh,
                        int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* c, void* stream) {
    return mu_fro_step_2d_impl<float>(A, m_l, n_l, lda, W, m_w, ldw, H, n_h, ldh, k, eps, w_update, clamp, ws, ws_bytes, c, stream);
}
int dnmf_mu_fro_step_2d_bf16a(const void* A, long m_l, long n_l, long lda, float* W, long m_w, long ldw, float* H, long n_h, long ldh,
                              int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* c, void* stream) {
    return mu_fro_step_2d_impl<bf16_t>((const bf16_t*)A, m_l, n_l, lda, W, m_w, ldw, H, n_h, ldh, k, eps, w_update, clamp, ws, ws_bytes, c, stream);
}

// The same for MU / KL (dist_nmf.py:351-407 with sum_axis :346-349, gather_W_H :268-291, UHT_glob :330-343, WTU_glob :294-318)
int dnmf_mu_kl_step_2d(const float* A, long m_l, long n_l, long lda, float* W, long m_w, long ldw, float* H, long n_h, long ldh,
                       int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* c, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && ws && c && m_l >= 1 && n_l >= 1 && lda >= n_l, "mu_kl_step_2d: bad arguments");
    Grid2d g;
    int rc;
    if ((rc = grid2d_init(g, "mu_kl_step_2d", c, m_l, n_l, m_w, n_h, ldw, ldh, k, ws, ws_bytes, stream))) return rc;
    const size_t kws = g.L.g_off;
    float *V = g.at(g.L.v_off), *Y = g.at(g.L.y_off), *Yb = g.at(g.L.yb_off), *x = g.at(g.L.x_off);
    hipStream_t st = g.st;
    // H_j: gathered once per step (the W phase changes W, not H)
    const float* Hop; long hb;
    if ((rc = g.gather_h(H, &Hop, &hb))) return rc;                                                       // gather_W_H :283-287
    const float* Wi;
    if (w_update) {                                                // KL_MU_update_W :351-369
        if ((rc = dnmf_rowsum(H, k, n_h, ldh, x, stream))) return rc;
        if ((rc = allreduce_f32(c, G_WORLD, x, (size_t)k, st))) return rc;                                // sum_axis :346-349
        if ((rc = g.gather_w(W, &Wi))) return rc;                                                         // :276-280
        if (g.blocked) rc = dnmf_kl_uht_hblocks(A, m_l, n_l, lda, Wi, k, Hop, hb, k, eps, V, k, ws, kws, stream);   // :337-338
        else rc = dnmf_kl_uht(A, m_l, n_l, lda, Wi, k, Hop, hb, k, eps, V, k, ws, kws, stream);
        if (rc) return rc;
        const float* UHT;
        if ((rc = g.scatter_to_w(V, &UHT))) return rc;                                                    // :340
        if ((rc = dnmf_kl_update_w(W, m_w, k, ldw, UHT, k, x, eps, stream))) return rc;                   // :369
    }
    if ((rc = dnmf_colsum(W, m_w, k, ldw, x, ws, kws, stream))) return rc;                                // KL_MU_update_H :371-389
    if ((rc = allreduce_f32(c, G_WORLD, x, (size_t)k, st))) return rc;
    if ((rc = g.gather_w(W, &Wi))) return rc;                                                             // :387
    if (g.sliced && c->p_r > 1) {
        // WTU_glob :311-312.  Round 5: ONE full-width product, then the k x n_l result is cut into the reduce-scatter's member blocks
        // (scatter_to_h) -- p_r sliced launches that wrote the blocks directly cost 4.19 ms against 3.97 + 0.04 ms on the config-4
        // block (each slice pays its own partial slabs and tail; tools/dbg/kl_slices.py).  The product wants H_j as one matrix:
        // assembled from the gathered blocks (k x n_l floats, one copy per member).
        if (g.blocked) {
            float* Hj = g.at(g.L.hj_off);
            for (int q = 0; q < c->p_r; ++q)
                COPY2D(Hj + (size_t)q * n_h, n_l, Hop + (size_t)q * k * n_h, n_h, n_h, k, "mu_kl_step_2d: assemble H_j");
            Hop = Hj; hb = n_l;
        }
        if ((rc = dnmf_kl_wtu(A, m_l, n_l, lda, Wi, k, Hop, hb, k, eps, Y, n_l, ws, kws, stream))) return rc;
        g.sliced = false;                                          // (scatter_to_h cuts Y into the member blocks)
    } else if (g.sliced) {                                         // one member: its "slice" is the whole width, written where the update reads it
        if ((rc = dnmf_kl_wtu(A, m_l, n_h, lda, Wi, k, Hop, g.blocked ? n_h : hb, k, eps, Yb, n_h, ws, kws, stream))) return rc;
    } else if ((rc = dnmf_kl_wtu(A, m_l, n_l, lda, Wi, k, Hop, hb, k, eps, Y, n_l, ws, kws, stream))) return rc;
    const float* WTU;
    if ((rc = g.scatter_to_h(Y, &WTU))) return rc;                                                        // :314-316
    const long ldwtu = (c->p_r == 1 && !g.sliced) ? n_l : n_h;
    if ((rc = dnmf_kl_update_h(H, k, n_h, ldh, WTU, ldwtu, x, eps, clamp, stream))) return rc;            // :389
    if (clamp) return dnmf_clamp_min(W, m_w, k, ldw, eps, stream);
    return DNMF_OK;
}

// One HALS / Frobenius step of a rank of a 1D grid (FRO_HALS_update, dist_nmf.py:873-934): products and Gram matrices as in the
// MU step (one packed allreduce per phase along the split axis), the W sweep column by column -- with row-sharded W (p_r > 1)
// the 8-byte sum of squares of every column is allreduced between the column kernels (utils.py:388-391), with local norms
// (p_r == 1) it is the persistent sweep, or k column launches when `column_sweep` != 0 -- then the H sweep.  `clamp`:
// H = max(H, eps), W = max(W, eps) afterwards (pyDNMF.py:170-172).
}  // extern "C"
namespace {
template <typename TA>
int hals_fro_step_1d_impl(const TA* A, long m_l, long n_l, long lda, float* W, long ldw, float* H, long ldh, int k,
                          float eps, int w_update, int clamp, int column_sweep, void* ws, size_t ws_bytes, dnmf_comm_t* c,
                          void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && ws && c && m_l >= 1 && n_l >= 1, "hals_fro_step_1d: bad arguments");
    REQUIRE(c->p_r == 1 || c->p_c == 1, "hals_fro_step_1d: a %d x %d grid is 2D", c->p_r, c->p_c);
    const int kp = 32 * kt;
    const Ws1d L = ws1d_layout(m_l, n_l, k);
    if (ws_bytes < L.total) return fail(DNMF_EWS, "hals_fro_step_1d: workspace %zu < %zu", ws_bytes, L.total);
    char* base = (char*)ws;
    const size_t kws = L.g_off;
    float* X = (float*)(base + L.x_off);
    const size_t big = std::max((size_t)k * n_l, (size_t)m_l * k);
    double* ss2 = reinterpret_cast<double*>(X + pad64(big) + (size_t)MAX_CHUNKS * 64 + (size_t)kp * kp + 128);
    hipStream_t st = S(stream);
    int rc;
    if (w_update) {                                               // FRO_HALS_update_W :873-891
        const size_t off = pad64((size_t)m_l * k);
        float* AH = X; float* Gx = X + off;
        if ((rc = dnmf_gram_hht(H, k, n_l, ldh, Gx, ws, kws, stream))) return rc;                 // :882
        if ((rc = AOps<TA>::aht(A, m_l, n_l, lda, H, k, ldh, AH, k, stream))) return rc;               // :883
        if (c->p_c != 1 && (rc = allreduce_f32(c, G_WORLD, X, off + (size_t)kp * kp, st))) return rc;
        if ((c->p_r != 1 && c->nranks > 1) || c->always) rc = hals_sweep_exchanged(c, W, m_l, k, ldw, AH, k, Gx, eps, ss2, ws, kws, stream, column_sweep);   // :884-891
        else rc = hals_sweep_local(W, m_l, k, ldw, AH, k, Gx, eps, column_sweep, ss2, ws, kws, stream);
        if (rc) return rc;
    }
    const size_t off = pad64((size_t)k * n_l);                    // FRO_HALS_update_H :893-909
    float* AtW = X; float* Gx = X + off;
    if ((rc = AOps<TA>::wta_gram(A, m_l, n_l, lda, W, k, ldw, AtW, n_l, Gx, ws, kws, stream))) return rc;   // :902-903
    if ((c->p_r != 1 || c->always) && (rc = allreduce_f32(c, G_WORLD, X, off + (size_t)kp * kp, st))) return rc;
    if ((rc = dnmf_hals_update_h(H, k, n_l, ldh, AtW, n_l, Gx, eps, stream))) return rc;           // :905-909
    if (clamp) {
        if ((rc = dnmf_clamp_min(H, k, n_l, ldh, eps, stream))) return rc;
        return dnmf_clamp_min(W, m_l, k, ldw, eps, stream);
    }
    return DNMF_OK;
}
}  // namespace
extern "C" {
int dnmf_hals_fro_step_1d(const float* A, long m_l, long n_l, long lda, float* W, long ldw, float* H, long ldh, int k,
                          float eps, int w_update, int clamp, int column_sweep, void* ws, size_t ws_bytes, dnmf_comm_t* c,
                          void* stream) {
    return hals_fro_step_1d_impl<float>(A, m_l, n_l, lda, W, ldw, H, ldh, k, eps, w_update, clamp, column_sweep, ws, ws_bytes, c, stream);
}
int dnmf_hals_fro_step_1d_bf16a(const void* A, long m_l, long n_l, long lda, float* W, long ldw, float* H, long ldh, int k,
                                float eps, int w_update, int clamp, int column_sweep, void* ws, size_t ws_bytes, dnmf_comm_t* c,
                                void* stream) {
    return hals_fro_step_1d_impl<bf16_t>((const bf16_t*)A, m_l, n_l, lda, W, ldw, H, ldh, k, eps, w_update, clamp, column_sweep, ws, ws_bytes, c, stream);
}

// The same on the 2D grid (FRO_HALS_update, dist_nmf.py:411-470): exchanges as dnmf_mu_fro_step_2d; the column norms of the W
// sweep are summed over ALL ranks (every rank holds other rows of W).
}  // extern "C"
namespace {
template <typename TA>
int hals_fro_step_2d_impl(const TA* A, long m_l, long n_l, long lda, float* W, long m_w, long ldw, float* H, long n_h, long ldh,
                          int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* c, void* stream) {
    const int kt = kt_of(k);
    REQUIRE(kt > 0 && A && W && H && ws && c && m_l >= 1 && n_l >= 1 && lda >= n_l, "hals_fro_step_2d: bad arguments");
    Grid2d g;
    int rc;
    if ((rc = grid2d_init(g, "hals_fro_step_2d", c, m_l, n_l, m_w, n_h, ldw, ldh, k, ws, ws_bytes, stream, AOps<TA>::f32))) return rc;
    const int kp = 32 * kt;
    const size_t kws = g.L.g_off;
    float *G = g.at(g.L.g_off), *V = g.at(g.L.v_off), *Y = g.at(g.L.y_off), *Yb = g.at(g.L.yb_off);
    double* ss2 = reinterpret_cast<double*>(g.at(g.L.x_off) + 128);
    hipStream_t st = g.st;
    if (w_update) {                                                // FRO_HALS_update_W :411-434
        if ((rc = dnmf_gram_hht(H, k, n_h, ldh, G, ws, kws, stream))) return rc;
        if ((rc = allreduce_f32(c, G_WORLD, G, (size_t)kp * kp, st))) return rc;                          // :426
        const float* Hop; long hb;
        if ((rc = g.gather_h(H, &Hop, &hb))) return rc;
        if constexpr (AOps<TA>::f32) { if (g.blocked) rc = dnmf_aht_hblocks(A, m_l, n_l, lda, Hop, hb, k, V, k, stream); else rc = AOps<TA>::aht(A, m_l, n_l, lda, Hop, k, hb, V, k, stream); }                 // AH_glob :427
        else rc = AOps<TA>::aht(A, m_l, n_l, lda, Hop, k, hb, V, k, stream);
        if (rc) return rc;
        const float* AH;
        if ((rc = g.scatter_to_w(V, &AH))) return rc;
        if (c->nranks > 1) rc = hals_sweep_exchanged(c, W, m_w, k, ldw, AH, k, G, eps, ss2, ws, kws, stream);      // :428-434
        else rc = hals_sweep_local(W, m_w, k, ldw, AH, k, G, eps, 0, ss2, ws, kws, stream);
        if (rc) return rc;
    }
    if ((rc = dnmf_gram_wtw(W, m_w, k, ldw, G, ws, kws, stream))) return rc;                              // FRO_HALS_update_H :436-452
    if ((rc = allreduce_f32(c, G_WORLD, G, (size_t)kp * kp, st))) return rc;
    const float* Wi;
    if ((rc = g.gather_w(W, &Wi))) return rc;
    if (g.sliced) {
        for (int q = 0; q < c->p_r; ++q)
            if ((rc = AOps<TA>::wta(A + q * n_h, m_l, n_h, lda, Wi, k, k, Yb + (size_t)q * k * n_h, n_h, ws, kws, stream))) return rc;
    } else if ((rc = AOps<TA>::wta(A, m_l, n_l, lda, Wi, k, k, Y, n_l, ws, kws, stream))) return rc;          // ATW_glob :448
    const float* AtW;
    if ((rc = g.scatter_to_h(Y, &AtW))) return rc;
    const long ldatw = (c->p_r == 1 && !g.sliced) ? n_l : n_h;
    if ((rc = dnmf_hals_update_h(H, k, n_h, ldh, AtW, ldatw, G, eps, stream))) return rc;                 // :449-452
    if (clamp) {
        if ((rc = dnmf_clamp_min(H, k, n_h, ldh, eps, stream))) return rc;
        return dnmf_clamp_min(W, m_w, k, ldw, eps, stream);
    }
    return DNMF_OK;
}
}  // namespace
extern "C" {
int dnmf_hals_fro_step_2d(const float* A, long m_l, long n_l, long lda, float* W, long m_w, long ldw, float* H, long n_h, long ldh,
                          int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* c, void* stream) {
    return hals_fro_step_2d_impl<float>(A, m_l, n_l, lda, W, m_w, ldw, H, n_h, ldh, k, eps, w_update, clamp, ws, ws_bytes, c, stream);
}
int dnmf_hals_fro_step_2d_bf16a(const void* A, long m_l, long n_l, long lda, float* W, long m_w, long ldw, float* H, long n_h, long ldh,
                                int k, float eps, int w_update, int clamp, void* ws, size_t ws_bytes, dnmf_comm_t* c, void* stream) {
    return hals_fro_step_2d_impl<bf16_t>((const bf16_t*)A, m_l, n_l, lda, W, m_w, ldw, H, n_h, ldh, k, eps, w_update, clamp, ws, ws_bytes, c, stream);
}

}  // extern "C"
