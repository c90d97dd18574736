// LDS atomic add throughput per CU on gfx950: binary64 add against 64-bit and 32-bit integer adds, random addresses
// in a 128 KB table (the per-cluster table of pxsom_cluster_sums at K = 400, C = 40).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, int iters, unsigned seed)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *tf = reinterpret_cast<double *>(smem);
    unsigned long long *tu = reinterpret_cast<unsigned long long *>(smem);
    unsigned *tw = reinterpret_cast<unsigned *>(smem);
    for (int e = threadIdx.x; e < 16384; e += 1024) tu[e] = 0ull;
    __syncthreads();
    unsigned s = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            s = s * 1664525u + 1013904223u;
            const unsigned a = (s >> 10) & 16383u;
            if (MODE == 0) __hip_atomic_fetch_add(tf + a, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) __hip_atomic_fetch_add(tu + a, 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) __hip_atomic_fetch_add(tw + 2 * a, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tu[seed & 1023];
}
template <int MODE>
float run(const char *name)
{
    unsigned long long *out;
    (void)hipMalloc(&out, 8 * 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    k<MODE><<<256, 1024, 131072>>>(out, 10, 1);
    (void)hipEventRecord(e0);
    k<MODE><<<256, 1024, 131072>>>(out, iters, 1);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ops_per_cu = 1024.0 * iters * 8;
    printf("%-28s %.3f ms: %.2f lane-atomics per ns per CU (%.2f per clk at 2.4 GHz)\n", name, ms, ops_per_cu / (ms * 1e6), ops_per_cu / (ms * 1e6) / 2.4);
    return ms;
}
int main()
{
    run<0>("ds_add_f64");
    run<1>("ds_add_u64");
    run<2>("ds_add_u32");
    return 0;
}
