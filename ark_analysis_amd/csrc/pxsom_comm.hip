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

struct pxsom_comm {
    ncclComm_t comm;
    int nranks, rank;
};

namespace pxsom {

// used by the training loop (pxsom_train.hip)
int comm_allreduce_sum_f64(pxsom_comm *c, double *buf, size_t count, hipStream_t st)
{
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
    pxsom_comm *c = new pxsom_comm{comm, nranks, rank};
    *out = c;
    return PXSOM_OK;
}

PXSOM_EXPORT int pxsom_comm_destroy(pxsom_comm *c)
{
    if (!c) return PXSOM_OK;
    ncclResult_t r = g_rccl.CommDestroy ? g_rccl.CommDestroy(c->comm) : ncclSuccess;
    delete c;
    return r == ncclSuccess ? PXSOM_OK : rccl_fail("ncclCommDestroy", r);
}

PXSOM_EXPORT int pxsom_comm_allreduce_sum_f64(pxsom_comm *c, double *buf_dev, size_t count, void *stream)
{
    if (!buf_dev && count) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_comm_allreduce_sum_f64: null buffer");
    return pxsom::comm_allreduce_sum_f64(c, buf_dev, count, reinterpret_cast<hipStream_t>(stream));
}
