// Rank-to-rank exchange of the batch rule, inside the library: one RCCL sum-all-reduce of the [K, C+1] binary64
// statistics per mini-batch step, enqueued on the same HIP stream as the step kernels -- no host round trip
// between a step and its exchange (the Python loop + torch.distributed.all_reduce it replaces is host-bound at
// ~50 us per step; a step kernel is ~14 us and the 18 KB all-reduce a few microseconds of xGMI latency).
//
// RCCL is bound at run time (dlopen + dlsym), not linked: a PyTorch process already holds its own librccl.so, and a
// second copy of the collective library in one process is asking for trouble.  pxsom_comm_bind() takes the path
// of the library to use (the caller passes torch's; NULL = the system "librccl.so.1").
//
// The reference has no analogue (pyFlowSOM trains on one core, cluster_helpers.py:106-109).
#include <dlfcn.h>
#include <rccl/rccl.h>   // types only; every entry point goes through the table below

#include <mutex>

#include <cstring>

#include "pxsom_common.h"
#include "pxsom_xch.h"

namespace {

struct RcclTable {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclTable g_rccl;
std::mutex g_rccl_mutex;

template <typename F>
bool resolve(void *h, const char *name, F *out)
{
    *out = reinterpret_cast<F>(dlsym(h, name));
    return *out != nullptr;
}

int rccl_fail(const char *what, ncclResult_t r)
{
    return pxsom::fail(PXSOM_ERR_HIP, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "rccl error");
}

}  // namespace

// ---- one-shot peer-to-peer all-reduce (round 4) -------------------------------------------------------------------
// The rule's exchange is 18 KB per step and sits on the critical path between two 12 us kernels: a ring / tree collective
// pays several hops of latency for a message that fits one store burst.  Here every rank owns an exchange block in ITS
// device memory -- [2 parities][nranks] flags + [2 parities][nranks][max_count] binary64 slots -- exported as a HIP IPC
// handle and mapped by every peer (same device: two processes on one GPU, which is how a one-GPU box validates it; other
// devices of the node: over xGMI).  One launch per exchange, one workgroup:
//   1. copy the local buffer into slot [parity][my rank] of EVERY rank's block (system-scope stores),
//   2. system fence, then the epoch number into flag [parity][my rank] of every block,
//   3. wait until the own block's flags of all ranks show the epoch (bounded: ~4 s, then the buffer turns to NaN and the
//      block's error word is set -- a missing peer must not hang the GPU),
//   4. add the nranks slots of the own block in rank order into the buffer: every rank adds the same numbers in the same
//      order, so the result is bit-identical on all ranks (an RCCL ring promises that only per algorithm and size).
// Epoch parity alternates, so a rank running ahead writes the other half of the block; it cannot lap a slow rank by two
// epochs because its own step 3 of the epoch in between needs that rank's flag.
using pxsom::kP2PMaxRanks;
using pxsom::P2PBlock;
struct P2PArgs {
    char *peer[kP2PMaxRanks];   // every rank's block as mapped here (peer[rank] = the own one)
    int nranks, rank;
    unsigned long long epoch;
    size_t max_count;
};

struct pxsom_comm {
    ncclComm_t comm;
    int nranks, rank;
    // p2p mode (comm == nullptr)
    bool p2p = false;
    char *block = nullptr;                       // own exchange block (device memory)
    char *peer[kP2PMaxRanks] = {};               // mapped blocks
    size_t max_count = 0;
    unsigned long long epoch = 0;
    char *dev_table = nullptr;                   // [nranks] peer pointers + the fused steps' ticket word, in device memory (connect)
    bool fused = false;                          // pxsom_comm_p2p_set_fused: the fused 10 x 10 step exchanges inside its launch
};

namespace {

using pxsom::p2p_slot;

__global__ __launch_bounds__(1024) void p2p_allreduce_kernel(P2PArgs a, double *__restrict__ buf, size_t count)
{
    const int tid = threadIdx.x, parity = (int)(a.epoch & 1ull);
    // 1. my contribution into everybody's block
    for (int p = 0; p < a.nranks; p++) {
        double *dst = p2p_slot(a.peer[p], parity, a.rank, a.nranks, a.max_count);
        for (size_t e = tid; e < count; e += 1024) __hip_atomic_store(dst + e, buf[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 2. ... is complete before anybody sees the flag
    __threadfence_system();
    __syncthreads();
    if (tid < a.nranks) {
        P2PBlock *pb = reinterpret_cast<P2PBlock *>(a.peer[tid]);
        __hip_atomic_store(&pb->flags[parity][a.rank], a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // 3. everybody's contribution has arrived in MY block
    P2PBlock *mine = reinterpret_cast<P2PBlock *>(a.peer[a.rank]);
    // (an exchange already on record as timed out: the peer that missed it will miss this one's epoch too -- a run of exchanges
    // enqueued back to back must not wait four seconds in each of them)
    __shared__ int s_timeout;
    if (tid == 0) s_timeout = __hip_atomic_load(&mine->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0ull;
    __syncthreads();
    if (tid < a.nranks && !s_timeout) {
        const long long t0 = (long long)wall_clock64();   // 100 MHz
        while (__hip_atomic_load(&mine->flags[parity][tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != a.epoch) {
            __builtin_amdgcn_s_sleep(8);
            if ((long long)wall_clock64() - t0 > 400000000ll) {   // 4 s
                s_timeout = 1;
                break;
            }
        }
    }
    __syncthreads();
    if (s_timeout) {
        if (tid == 0) {   // the FIRST exchange that timed out stays on record (a later one must not overwrite it)
            unsigned long long none = 0ull;
            (void)__hip_atomic_compare_exchange_strong(&mine->error, &none, a.epoch, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_SYSTEM);
        }
        for (size_t e = tid; e < count; e += 1024) buf[e] = __builtin_nan("");
        return;
    }
    // 4. the sum, in rank order
    for (size_t e = tid; e < count; e += 1024) {
        double acc = 0.0;
        for (int p = 0; p < a.nranks; p++)
            acc += __hip_atomic_load(p2p_slot(a.peer[a.rank], parity, p, a.nranks, a.max_count) + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        buf[e] = acc;
    }
}

int p2p_allreduce(pxsom_comm *c, double *buf, size_t count, hipStream_t st)
{
    if (count > c->max_count)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "p2p exchange: %zu values, the blocks hold %zu", count, c->max_count);
    for (int p = 0; p < c->nranks; p++)
        if (!c->peer[p]) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "p2p exchange: pxsom_comm_p2p_connect has not run");
    P2PArgs a;
    for (int p = 0; p < kP2PMaxRanks; p++) a.peer[p] = p < c->nranks ? c->peer[p] : nullptr;
    a.nranks = c->nranks;
    a.rank = c->rank;
    a.epoch = ++c->epoch;
    a.max_count = c->max_count;
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(1), dim3(1024), 0, st, a, buf, count);
    PXSOM_LAUNCH_CHECK("p2p_allreduce_kernel");
    return PXSOM_OK;
}

}  // namespace

namespace pxsom {

// The rule's exchange inside the step launches (pxsom_batch_step.hip, EXCH instantiations): hands the training loop the blocks
// and reserves the epochs of `exchanges` fused exchanges (the next separate all-reduce continues behind them).
bool comm_fused_begin(pxsom_comm *c, int exchanges, size_t count, FusedXch *out)
{
    if (!c || !c->p2p || !c->fused || !c->dev_table || count > c->max_count || exchanges < 0) return false;
    for (int p = 0; p < c->nranks; p++)
        if (!c->peer[p]) return false;
    out->peers = reinterpret_cast<char *const *>(c->dev_table);
    out->ticket = reinterpret_cast<unsigned *>(c->dev_table + kP2PMaxRanks * sizeof(char *));
    out->nranks = c->nranks;
    out->rank = c->rank;
    out->max_count = c->max_count;
    out->epoch_base = c->epoch;
    c->epoch += (unsigned long long)exchanges;
    return true;
}

// used by the training loop (pxsom_train.hip)
int comm_allreduce_sum_f64(pxsom_comm *c, double *buf, size_t count, hipStream_t st)
{
    if (c && c->p2p) return p2p_allreduce(c, buf, count, st);
    if (!c || !g_rccl.AllReduce) return fail(PXSOM_ERR_INVALID_ARG, "pxsom: exchange without a communicator");
    ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclDouble, ncclSum, c->comm, st);
    return r == ncclSuccess ? PXSOM_OK : rccl_fail("ncclAllReduce", r);
}

}  // namespace pxsom

PXSOM_EXPORT int pxsom_comm_bind(const char *librccl_path)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return PXSOM_OK;
    const char *path = librccl_path && librccl_path[0] ? librccl_path : "librccl.so.1";
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_comm_bind: %s", dlerror());
    RcclTable t;
    t.handle = h;
    if (!resolve(h, "ncclGetUniqueId", &t.GetUniqueId) || !resolve(h, "ncclCommInitRank", &t.CommInitRank) ||
        !resolve(h, "ncclCommDestroy", &t.CommDestroy) || !resolve(h, "ncclAllReduce", &t.AllReduce) ||
        !resolve(h, "ncclGetErrorString", &t.GetErrorString)) {
        dlclose(h);
        return pxsom::fail(PXSOM_ERR_UNSUPPORTED, "pxsom_comm_bind: %s lacks the RCCL entry points", path);
    }
    g_rccl = t;
    return PXSOM_OK;
}

PXSOM_EXPORT int pxsom_comm_unique_id(void *id_out, size_t id_bytes)
{
    if (!g_rccl.GetUniqueId) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_comm_unique_id: call pxsom_comm_bind first");
    if (!id_out || id_bytes != PXSOM_COMM_ID_BYTES)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_comm_unique_id: the id is %d bytes", PXSOM_COMM_ID_BYTES);
    static_assert(sizeof(ncclUniqueId) == PXSOM_COMM_ID_BYTES, "PXSOM_COMM_ID_BYTES out of step with RCCL");
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
    memcpy(id_out, &id, sizeof(id));
    return PXSOM_OK;
}

PXSOM_EXPORT int pxsom_comm_create(const void *id, size_t id_bytes, int nranks, int rank, pxsom_comm **out)
{
    if (!g_rccl.CommInitRank) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_comm_create: call pxsom_comm_bind first");
    if (!id || id_bytes != PXSOM_COMM_ID_BYTES || !out || nranks < 1 || rank < 0 || rank >= nranks)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_comm_create: rank %d of %d", rank, nranks);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, uid, rank);   // collective: every rank of the job calls it
    if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", r);
    pxsom_comm *c = new pxsom_comm();
    c->comm = comm;
    c->nranks = nranks;
    c->rank = rank;
    *out = c;
    return PXSOM_OK;
}

// ---- p2p communicator: create (allocates the own block) -> handle (64 bytes to hand to every peer) -> connect (maps the
// peers' blocks; collective in the sense that every rank must have created its block first) ---------------------------
PXSOM_EXPORT int pxsom_comm_p2p_create(int nranks, int rank, size_t max_count, pxsom_comm **out)
{
    if (!out || nranks < 1 || nranks > kP2PMaxRanks || rank < 0 || rank >= nranks || max_count < 1)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_comm_p2p_create: rank %d of %d (at most %d), %zu values", rank, nranks,
                           kP2PMaxRanks, max_count);
    pxsom_comm *c = new pxsom_comm();
    c->comm = nullptr;
    c->nranks = nranks;
    c->rank = rank;
    c->p2p = true;
    c->max_count = max_count;
    const size_t bytes = sizeof(P2PBlock) + (size_t)2 * nranks * max_count * sizeof(double);
    // fine-grained device memory: peers write into it and this rank reads it while kernels run
    // (no coarse-grained fall-back: kernels of other devices poll and write this block while this device's kernels run, which
    // only fine-grained memory keeps coherent -- without it the communicator is not made and the caller takes another route)
    hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void **>(&c->block), bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        delete c;
        return pxsom::hip_fail(e, "pxsom_comm_p2p_create: fine-grained exchange block");
    }
    e = hipMemset(c->block, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();   // the block is clear before its handle can reach a peer
    if (e != hipSuccess) {
        (void)hipFree(c->block);
        delete c;
        return pxsom::hip_fail(e, "pxsom_comm_p2p_create: clearing the exchange block");
    }
    c->peer[rank] = c->block;
    *out = c;
    return PXSOM_OK;
}

PXSOM_EXPORT int pxsom_comm_p2p_handle(pxsom_comm *c, void *handle_out, size_t handle_bytes)
{
    static_assert(sizeof(hipIpcMemHandle_t) == PXSOM_P2P_HANDLE_BYTES, "PXSOM_P2P_HANDLE_BYTES out of step with HIP");
    if (!c || !c->p2p || !handle_out || handle_bytes != PXSOM_P2P_HANDLE_BYTES)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_comm_p2p_handle: a p2p communicator and %d bytes", PXSOM_P2P_HANDLE_BYTES);
    hipIpcMemHandle_t h;
    PXSOM_HIP_TRY(hipIpcGetMemHandle(&h, c->block));
    memcpy(handle_out, &h, sizeof(h));
    return PXSOM_OK;
}

PXSOM_EXPORT int pxsom_comm_p2p_connect(pxsom_comm *c, const void *handles, size_t handles_bytes)
{
    if (!c || !c->p2p || !handles || handles_bytes != (size_t)c->nranks * PXSOM_P2P_HANDLE_BYTES)
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_comm_p2p_connect: nranks x %d bytes of handles", PXSOM_P2P_HANDLE_BYTES);
    for (int p = 0; p < c->nranks; p++) {
        if (p == c->rank || c->peer[p]) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char *>(handles) + (size_t)p * PXSOM_P2P_HANDLE_BYTES, sizeof(h));
        void *ptr = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return pxsom::hip_fail(e, "pxsom_comm_p2p_connect: hipIpcOpenMemHandle");
        c->peer[p] = static_cast<char *>(ptr);
    }
    // the table the fused step kernels read (pxsom_xch.h FusedXch): peer pointers, then the ticket word
    if (!c->dev_table) {
        PXSOM_HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->dev_table), (kP2PMaxRanks + 1) * sizeof(char *)));
        char *host[kP2PMaxRanks + 1] = {};
        for (int p = 0; p < c->nranks; p++) host[p] = c->peer[p];
        PXSOM_HIP_TRY(hipMemcpy(c->dev_table, host, sizeof(host), hipMemcpyHostToDevice));
    }
    return PXSOM_OK;
}

// 0: every exchange so far completed; otherwise the epoch at which a peer did not show up in time (the buffers of that
// exchange were turned to NaN)
PXSOM_EXPORT int pxsom_comm_p2p_error(pxsom_comm *c, unsigned long long *epoch_out)
{
    if (!c || !c->p2p || !epoch_out) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_comm_p2p_error: bad arguments");
    P2PBlock head;
    PXSOM_HIP_TRY(hipMemcpy(&head, c->block, sizeof(head), hipMemcpyDeviceToHost));
    *epoch_out = head.error;
    return PXSOM_OK;
}

PXSOM_EXPORT int pxsom_comm_p2p_set_fused(pxsom_comm *c, int on)
{
    if (!c || !c->p2p) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_comm_p2p_set_fused: not a peer-to-peer communicator");
    c->fused = on != 0;
    return PXSOM_OK;
}

PXSOM_EXPORT int pxsom_comm_destroy(pxsom_comm *c)
{
    if (!c) return PXSOM_OK;
    if (c->p2p) {
        for (int p = 0; p < c->nranks; p++)
            if (p != c->rank && c->peer[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
        if (c->block) (void)hipFree(c->block);
        if (c->dev_table) (void)hipFree(c->dev_table);
        delete c;
        return PXSOM_OK;
    }
    ncclResult_t r = g_rccl.CommDestroy ? g_rccl.CommDestroy(c->comm) : ncclSuccess;
    delete c;
    return r == ncclSuccess ? PXSOM_OK : rccl_fail("ncclCommDestroy", r);
}

PXSOM_EXPORT int pxsom_comm_allreduce_sum_f64(pxsom_comm *c, double *buf_dev, size_t count, void *stream)
{
    if (!buf_dev && count) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_comm_allreduce_sum_f64: null buffer");
    return pxsom::comm_allreduce_sum_f64(c, buf_dev, count, reinterpret_cast<hipStream_t>(stream));
}
