// pxsom_xch.h -- layout of a rank's peer-to-peer exchange block (pxsom_comm's p2p mode, pxsom_comm.hip) and what the fused
// mini-batch step needs to run the rule's exchange INSIDE its own launch (round 5, pxsom_batch_step.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace pxsom {

constexpr int kP2PMaxRanks = 16;
struct P2PBlock {           // head of a rank's exchange block: [2 parities][nranks] flags, then the binary64 slots
    unsigned long long flags[2][kP2PMaxRanks];
    unsigned long long error;
    unsigned long long pad[7];
};

__host__ __device__ inline double *p2p_slot(char *block, int parity, int src, int nranks, size_t max_count)
{
    return reinterpret_cast<double *>(block + sizeof(P2PBlock)) + ((size_t)parity * nranks + src) * max_count;
}

// a p2p communicator as a step kernel sees it (FusedXch::peers / ticket are device memory owned by the communicator)
struct FusedXch {
    char *const *peers = nullptr;        // [nranks] every rank's block as mapped here (peers[rank]: the own one)
    unsigned *ticket = nullptr;          // one device word, zero between launches
    int nranks = 0, rank = 0;
    size_t max_count = 0;
    unsigned long long epoch_base = 0;   // the communicator's epoch before the run of fused exchanges
};

}  // namespace pxsom

struct pxsom_comm;
namespace pxsom {
// true for a connected peer-to-peer communicator; reserves `exchanges` epochs and describes the blocks (pxsom_comm.hip)
bool comm_fused_begin(pxsom_comm *c, int exchanges, size_t count, FusedXch *out);
}  // namespace pxsom
