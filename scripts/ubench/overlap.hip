// Do an MFMA-only wave and a VALU-only wave on the SAME SIMD overlap on gfx950?
// 512 blocks of 256 threads (2 per CU, so every SIMD hosts one wave of each of two blocks).
// role(block) decides MFMA-only or VALU-only; compare mixed roles against each role alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(float *out, int iters, float seed, int mode, int *cu_roles)
{
    // mode 0: all MFMA, 1: all VALU, 2: role = blockIdx parity, 3: role = (blockIdx / 256) parity
    int role = mode == 0 ? 0 : mode == 1 ? 1 : mode == 2 ? (blockIdx.x & 1) : ((blockIdx.x >> 8) & 1);
    half8 a, b;
    for (int j = 0; j < 8; j++) {
        a[j] = (_Float16)(seed * (threadIdx.x + j));
        b[j] = (_Float16)(seed * (threadIdx.x * 3 + j));
    }
    f32x4 acc = {seed, 0.f, 0.f, 0.f};
    float v0 = seed, v1 = 2 * seed, v2 = 3 * seed, v3 = 4 * seed;
    if (role == 0) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 32; i++) {
                v0 = __builtin_amdgcn_fmed3f(v0, v1, v2);
                v1 = __builtin_amdgcn_fmed3f(v1, v2, v3);
                v2 = __builtin_amdgcn_fmed3f(v2, v3, v0);
                v3 = __builtin_amdgcn_fmed3f(v3, v0, v1);
            }
        }
    }
    if (threadIdx.x == 0 && cu_roles) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        cu_roles[blockIdx.x] = (int)((xcc & 0xf) << 16 | (hw & 0xffff));
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + v0 + v1 + v2 + v3;
}

float run(int mode, int iters_m, int nblocks = 512)
{
    float *out;
    int *roles;
    (void)hipMalloc(&out, sizeof(float) * 512 * 256);
    (void)hipMalloc(&roles, sizeof(int) * 512);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<<<nblocks, 256>>>(out, 10, 1e-3f, mode, nullptr);
    (void)hipEventRecord(e0);
    k<<<nblocks, 256>>>(out, iters_m, 1e-3f, mode, roles);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (mode >= 2) {
        int h[512];
        (void)hipMemcpy(h, roles, sizeof(h), hipMemcpyDeviceToHost);
        // count CUs (xcc, hw_id cu/se bits) that got one block of each role
        int mixed = 0, same = 0;
        for (int i = 0; i < 512; i++)
            for (int j = i + 1; j < 512; j++)
                if ((h[i] & 0xffffff00) == (h[j] & 0xffffff00) && (((h[i] >> 8) & 0xf) == ((h[j] >> 8) & 0xf))) {
                    int ri = mode == 2 ? (i & 1) : ((i >> 8) & 1), rj = mode == 2 ? (j & 1) : ((j >> 8) & 1);
                    (ri != rj ? mixed : same)++;
                }
        printf("  (pairs on same hw id: mixed-role %d, same-role %d)\n", mixed, same);
    }
    (void)hipFree(out);
    (void)hipFree(roles);
    return ms;
}

int main()
{
    int it = 20000;
    float tm = run(0, it), tv = run(1, it);
    printf("all MFMA  : %.3f ms\nall VALU  : %.3f ms\n", tm, tv);
    printf("one MFMA wave per SIMD alone : %.3f ms\none VALU wave per SIMD alone : %.3f ms\n", run(0, it, 256), run(1, it, 256));
    float t2 = run(2, it);
    printf("mixed by blockIdx parity      : %.3f ms\n", t2);
    float t3 = run(3, it);
    printf("mixed by (blockIdx/256) parity: %.3f ms\n", t3);
    printf("if roles overlap on a SIMD expect ~max(%.3f, %.3f)/2*... ; serialised expect ~(%.3f+%.3f)/2\n", tm, tv, tm, tv);
    return 0;
}
