// Micro-benchmark: marginal cost of VALU instruction types issued between f16 MFMAs on gfx950.
// Each kernel runs 4 sequential 3-deep v_mfma_f32_16x16x32_f16 chains per iteration with N filler
// ops of one type after every MFMA; cost per op = (t(N=8) - t(N=0)) / 8.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OP, int NV>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    half8 a[4], b[4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) {
            a[i][j] = (_Float16)(seed * (threadIdx.x + i + j));
            b[i][j] = (_Float16)(seed * (threadIdx.x * 3 + i * 7 + j));
        }
    f32x4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = f32x4{seed, 0.f, 0.f, 0.f};
    float v[4] = {seed, seed * 2, seed * 3, seed * 4};
    unsigned mask = 0xffffffc0u + (unsigned)(iters & 1);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            f32x4 t0 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
                t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk], b[i], t0, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NV; j++) {
                    float &x = v[j & 3];
                    float y = v[(j + 1) & 3], z = v[(j + 2) & 3];
                    if (OP == 0) x = __builtin_amdgcn_fmed3f(x, y, z);                    // v_med3_f32 (3 VGPR)
                    if (OP == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                    if (OP == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(y));   // VOP2
                    if (OP == 3) x = __uint_as_float((__float_as_uint(x) & mask) | (unsigned)j);  // v_and_or (1 VGPR)
                    if (OP == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                    if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
                    if (OP == 6) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(x) : "v"(y));
                    if (OP == 7) asm volatile("v_med3_f32 %0, %0, %1, 1.0" : "+v"(x) : "v"(y));  // 2 VGPR + const
                    if (OP == 8) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));   // 2 distinct VGPR
                    if (OP == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&v[(j & 1) * 2]) : "v"(*(double *)&v[((j + 1) & 1) * 2]));
                    if (OP == 10) asm volatile("v_dot2c_f32_f16 %0, %1, %1" : "+v"(x) : "v"(y));
                    if (OP == 11) asm volatile("v_fma_mixlo_f16 %0, %0, %1, 0" : "+v"(x) : "v"(y));
                    if (OP == 12) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(v[(j + 1) & 3]));
                    if (OP == 13) asm volatile("v_cvt_f32_f16 %0, %1" : "+v"(x) : "v"(y));
                    if (OP == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y));
                    if (OP == 15) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(y));
                }
            }
            acc[i] = t0;
        }
    }
    float s = v[0] + v[1] + v[2] + v[3];
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP, int NV>
float run1()
{
    int cus = 256, iters = 10000, w = 2;
    float *out;
    (void)hipMalloc(&out, sizeof(float) * cus * 8 * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<OP, NV><<<cus * w, 256>>>(out, 100, 1e-3f);
    (void)hipEventRecord(e0);
    k<OP, NV><<<cus * w, 256>>>(out, iters, 1e-3f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms * 1e6f / (iters * 12.0f * w);  // ns per MFMA slot per SIMD
}

template <int OP>
void run(const char *name)
{
    float t0 = run1<OP, 0>(), t8 = run1<OP, 8>();
    printf("%-34s  base %.2f ns/MFMA, +8 ops %.2f ns  => %.2f ns per op\n", name, t0, t8, (t8 - t0) / 8);
}

int main()
{
    run<0>("v_med3_f32 (3 VGPR)");
    run<1>("v_max3_f32 (3 VGPR)");
    run<8>("v_max3_f32 (2 distinct VGPR)");
    run<7>("v_med3_f32 (2 VGPR + const)");
    run<2>("v_max_f32 VOP2");
    run<4>("v_add_f32 VOP2");
    run<5>("v_fma_f32 (3 VGPR)");
    run<3>("v_and_or_b32 (1 VGPR, sgpr, imm)");
    run<6>("v_cvt_pkrtz_f16_f32");
    run<9>("v_pk_mul_f32");
    run<10>("v_dot2c_f32_f16");
    run<11>("v_fma_mixlo_f16");
    run<12>("s_nop1 + v_permlane32_swap");
    run<13>("v_cvt_f32_f16");
    run<14>("v_cndmask_b32");
    run<15>("v_mov_b32");
    return 0;
}
