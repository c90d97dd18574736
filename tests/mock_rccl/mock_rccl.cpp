// A stand-in for librccl.so with the five entry points pxsom_comm.hip binds, for ranks that SHARE one GPU (RCCL itself
// refuses two ranks on a device): all-reduce through a POSIX shared-memory segment -- synchronous, slow, and exact
// (binary64 sums in rank order).  TEST INFRASTRUCTURE: lets tests/test_gpu_exchange.py run the in-library step loop
// of pxsom_batch_train_steps_sharded on two real ranks on the 1-GPU boxes; what it cannot stand in for is RCCL.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

namespace {
constexpr size_t kMaxCount = 1 << 17;   // doubles per rank (1 MiB): far above K * (C + 1)
constexpr int kMaxRanks = 8;
struct Segment {
    std::atomic<int> joined;
    std::atomic<int> arrived;
    std::atomic<int> generation;
    double buf[kMaxRanks][kMaxCount];
};
struct Comm {
    Segment *seg;
    int nranks, rank;
    char name[64];
};
void barrier(Comm *c)
{
    const int gen = c->seg->generation.load();
    if (c->seg->arrived.fetch_add(1) + 1 == c->nranks) {
        c->seg->arrived.store(0);
        c->seg->generation.fetch_add(1);
    } else {
        while (c->seg->generation.load() == gen) usleep(20);
    }
}
}  // namespace

extern "C" {
__attribute__((visibility("default"))) ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "pxsom_mock_%d_%ld", (int)getpid(), (long)time(nullptr));
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm;
    c->nranks = nranks;
    c->rank = rank;
    snprintf(c->name, sizeof(c->name), "/%.60s", id.internal);
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Segment)) != 0) return ncclSystemError;
    void *p = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    c->seg = static_cast<Segment *>(p);   // (a fresh segment is zero-filled: counters start at 0)
    c->seg->joined.fetch_add(1);
    while (c->seg->joined.load() < nranks) usleep(50);
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    if (c->rank == 0) shm_unlink(c->name);
    munmap(c->seg, sizeof(Segment));
    delete c;
    return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count,
                                                                  ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                                                                  hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || datatype != ncclDouble || op != ncclSum || count > kMaxCount) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->seg->buf[c->rank], sendbuff, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return ncclUnhandledCudaError;
    barrier(c);   // every rank's contribution is in the segment
    std::vector<double> total(count, 0.0);
    for (int r = 0; r < c->nranks; r++)
        for (size_t i = 0; i < count; i++) total[i] += c->seg->buf[r][i];
    barrier(c);   // everybody has read: the slots may be overwritten by the next call
    // fault injection (tests/test_gpu_exchange.py): a collective library whose sums are off by one in the first word
    if (count && getenv("MOCK_RCCL_WRONG_SUM")) total[0] += 1.0;
    if (hipMemcpy(recvbuff, total.data(), count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        return ncclUnhandledCudaError;
    return ncclSuccess;
}

__attribute__((visibility("default"))) const char *ncclGetErrorString(ncclResult_t r)
{
    return r == ncclSuccess ? "no error" : "mock collective error";
}
}
