// What does the end-of-kernel flush of per-workgroup tables into ONE statistics buffer cost, and what do replicas buy?
// Every workgroup of a launch adds kStats binary64 words (K * (C + 1) = 2 300 for the Pixie shape) with device-scope
// atomics -- the one-pass labels + mean-table kernel does this from 512 workgroups at once, a large training step from 256.
//   mode 0: one table                         mode 1: 8 replicas, chosen by HW_REG_XCC_ID
//   mode 2: 8 replicas, chosen by blockIdx % 8   mode 3: R replicas by blockIdx % R (R = 16, 32, 64)
//   mode 4: mode 1 + the LAST workgroup (device-scope ticket) folds the replicas into the table and clears them
// Output: average launch duration (hipEvents over 50 launches) per mode and grid, and a check of the sums.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kStats = 2300;
constexpr int kMaxRep = 64;
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }

__global__ __launch_bounds__(256) void flush_probe(double *table, double *replicas, unsigned *ticket, int mode, int nrep)
{
    __shared__ double ls[kStats];
    __shared__ unsigned s_last;
    for (int e = threadIdx.x; e < kStats; e += 256) ls[e] = 1.0;
    __syncthreads();
    double *dst = table;
    if (mode == 1 || mode == 4) dst = replicas + (size_t)(xcc_id() & 7u) * kStats;
    else if (mode == 2) dst = replicas + (size_t)(blockIdx.x & 7u) * kStats;
    else if (mode == 3) dst = replicas + (size_t)(blockIdx.x % (unsigned)nrep) * kStats;
    // staggered start, as the kernels do
    const int shift = (int)((blockIdx.x * 97u) % (unsigned)kStats);
    for (int e0 = threadIdx.x; e0 < kStats; e0 += 256) {
        int e = e0 + shift;
        if (e >= kStats) e -= kStats;
        __hip_atomic_fetch_add(dst + e, ls[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (mode == 4) {
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
        __syncthreads();
        if (s_last) {
            __threadfence();
            for (int e = threadIdx.x; e < kStats; e += 256) {
                double v[8];
#pragma unroll
                for (int r = 0; r < 8; r++) v[r] = __hip_atomic_load(replicas + (size_t)r * kStats + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                double s = table[e];
#pragma unroll
                for (int r = 0; r < 8; r++) s += v[r];
                table[e] = s;
#pragma unroll
                for (int r = 0; r < 8; r++) replicas[(size_t)r * kStats + e] = 0.0;
            }
            if (threadIdx.x == 0) *ticket = 0u;
        }
    }
}

int main()
{
    double *table, *replicas;
    unsigned *ticket;
    hipMalloc(&table, kStats * sizeof(double));
    hipMalloc(&replicas, (size_t)kMaxRep * kStats * sizeof(double));
    hipMalloc(&ticket, 256);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int grids[] = {69, 256, 512};
    struct Case { int mode, nrep; const char *name; };
    const Case cases[] = {{0, 1, "one table"}, {1, 8, "8 replicas by XCC_ID"}, {2, 8, "8 replicas by blockIdx % 8"},
                          {3, 16, "16 replicas"}, {3, 32, "32 replicas"}, {3, 64, "64 replicas"},
                          {4, 8, "8 replicas by XCC_ID + last workgroup folds"}};
    for (int grid : grids)
        for (const Case &cs : cases) {
            hipMemset(table, 0, kStats * sizeof(double));
            hipMemset(replicas, 0, (size_t)kMaxRep * kStats * sizeof(double));
            hipMemset(ticket, 0, 256);
            const int reps = 50;
            for (int w = 0; w < 3; w++) hipLaunchKernelGGL(flush_probe, dim3(grid), dim3(256), 0, 0, table, replicas, ticket, cs.mode, cs.nrep);
            hipDeviceSynchronize();
            hipEventRecord(a, 0);
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(flush_probe, dim3(grid), dim3(256), 0, 0, table, replicas, ticket, cs.mode, cs.nrep);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms = 0.f;
            hipEventElapsedTime(&ms, a, b);
            std::vector<double> t(kStats), r((size_t)kMaxRep * kStats);
            hipMemcpy(t.data(), table, kStats * sizeof(double), hipMemcpyDeviceToHost);
            hipMemcpy(r.data(), replicas, r.size() * sizeof(double), hipMemcpyDeviceToHost);
            double total = 0.0;
            for (double v : t) total += v;
            for (double v : r) total += v;
            const double want = (double)(reps + 3) * grid * kStats;
            printf("grid %3d  %-46s %7.2f us per launch   sums %s\n", grid, cs.name, 1000.0 * ms / reps, total == want ? "ok" : "WRONG");
        }
    // reference: an empty launch of the same shape
    return 0;
}
