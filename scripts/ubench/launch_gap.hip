// What does a back-to-back dependent launch cost as a function of the launch shape (workgroups, threads, dynamic LDS)?
// 64 launches of a kernel that does (almost) nothing, timed with events; plus a body of ~10 us to see the gap beside work.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *p, int spin)
{
    extern __shared__ int s[];
    if (threadIdx.x == 0) s[0] = spin;
    __syncthreads();
    long long t0 = clock64();
    while (clock64() - t0 < (long long)s[0]) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, 1);
}
int main()
{
    int *d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    const int cfgs[][3] = {{128, 256, 0}, {128, 512, 0}, {128, 512, 100 * 1024}, {128, 512, 140 * 1024}, {256, 256, 0},
                           {256, 256, 64 * 1024}, {64, 512, 100 * 1024}, {32, 1024, 100 * 1024}, {128, 1024, 100 * 1024}};
    for (auto &c : cfgs)
        for (int spin : {0, 24000}) {   // 0 and ~10 us at 2.4 GHz
            for (int i = 0; i < 8; i++) hipLaunchKernelGGL(k, dim3(c[0]), dim3(c[1]), c[2], 0, d, spin);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            for (int i = 0; i < 64; i++) hipLaunchKernelGGL(k, dim3(c[0]), dim3(c[1]), c[2], 0, d, spin);
            hipEventRecord(e1, 0); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("wgs %4d threads %4d lds %6d spin %5d: %.2f us per launch\n", c[0], c[1], c[2], spin, ms * 1e3 / 64);
        }
    return 0;
}
