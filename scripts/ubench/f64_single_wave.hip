// Micro-benchmark: what does ONE wave on an otherwise idle MI355X see?  Shader clock (s_memtime
// ticks vs the 100 MHz wall clock), binary64 add latency (dependent chain) and issue rate
// (8 independent chains), LDS broadcast read latency, DPP/permlane wave-min cost.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(64) void k(double *out, long long *t, int iters, double seed)
{
    __shared__ double sm[64];
    sm[threadIdx.x] = seed * threadIdx.x;
    __syncthreads();
    double a = seed, b[8];
    for (int i = 0; i < 8; i++) b[i] = seed * i;
    long long w0 = wall_clock64(), c0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(seed));
    }
    long long w1 = wall_clock64(), c1 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(b[u & 7]) : "v"(seed));
    }
    long long w2 = wall_clock64(), c2 = clock64();
    int idx = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) idx = (int)sm[idx & 63] & 63;   // dependent LDS reads
    }
    long long w3 = wall_clock64(), c3 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(seed));
    }
    long long w4 = wall_clock64(), c4 = clock64();
    if (threadIdx.x == 0) {
        t[0] = w1 - w0; t[1] = c1 - c0; t[2] = w2 - w1; t[3] = c2 - c1; t[4] = w3 - w2; t[5] = c3 - c2;
        t[6] = w4 - w3; t[7] = c4 - c3;
    }
    double s = a + idx;
    for (int i = 0; i < 8; i++) s += b[i];
    out[threadIdx.x] = s;
}

int main()
{
    double *out; long long *t;
    hipMalloc(&out, 64 * 8); hipMalloc(&t, 64);
    const int iters = 20000;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, t, iters, 1e-9);
        hipDeviceSynchronize();
        long long h[8];
        hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        const double ops = iters * 16.0;
        const char *nm[4] = {"dep v_add_f64", "indep v_add_f64 x8", "dep LDS read", "dep v_fma_f64"};
        for (int i = 0; i < 4; i++)
            printf("rep %d %-20s wall %.2f ns/op  clock64 %.2f ticks/op  (ticks per wall ns %.3f)\n", rep, nm[i],
                   h[2 * i] * 10.0 / ops, h[2 * i + 1] / ops, (double)h[2 * i + 1] / (h[2 * i] * 10.0));
    }
    return 0;
}
