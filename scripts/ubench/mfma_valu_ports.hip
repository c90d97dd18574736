// Follow-up to mfma_valu_overlap.hip: what takes the overlap away in the real filter?  One stream per wave with 12 MFMAs (16x16x32 f16, four
// accumulator chains of three, as a node block of the packed-K filter) and 48 vector instructions that do not depend on them, 4 waves per SIMD:
//   ops 0: v_med3_f32 d, d, y, y (two registers read)      1: v_med3_f32 d, d, p, q with p, q from a pool of 16 registers (three read)
//   mfma 0: every MFMA reads the same A / B registers         1: twelve different B operands and three different A operands (the filter's)
//   mfma 2: as 1 with the B operands in AccVGPRs (v_accvgpr_write once, outside the loop)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MF, int OPS>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float seed, int what)
{
    half8 a[3], b[12];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 8; j++) a[i][j] = (_Float16)(seed * (threadIdx.x + i + j));
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 8; j++) b[i][j] = (_Float16)(seed * (threadIdx.x * 3 + i * 5 + j));
    f32x4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = f32x4{seed, 0.f, 0.f, 0.f};
    float v[16], pool[16];
    for (int i = 0; i < 16; i++) {
        v[i] = seed * (threadIdx.x + i);
        pool[i] = seed * (i + 2);
    }
    const float y = seed * 3.f;
    for (int it = 0; it < iters; it++) {
        if (what & 1) {
#pragma unroll
            for (int m = 0; m < 3; m++)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if constexpr (MF == 0) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], acc[u], 0, 0, 0);
                    else if constexpr (MF == 1) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[u * 3 + m], acc[u], 0, 0, 0);
                    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a[m]), "a"(b[u * 3 + m]));
                }
        }
        if (what & 2) {
#pragma unroll
            for (int i = 0; i < 48; i++) {
                if constexpr (OPS == 0) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(v[i & 15]) : "v"(y));
                else asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i & 15]) : "v"(pool[(i * 5 + 1) & 15]), "v"(pool[(i * 3 + 7) & 15]));
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < 4; i++) r += acc[i][0];
    for (int i = 0; i < 16; i++) r += v[i];
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}

template <typename F>
float timed(F launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int MF, int OPS>
void go(const char *name, float *out)
{
    const int it = 3000;
    const float tm = timed([&] { k<MF, OPS><<<256, 1024>>>(out, it, 1e-3f, 1); });
    const float tv = timed([&] { k<MF, OPS><<<256, 1024>>>(out, it, 1e-3f, 2); });
    const float tb = timed([&] { k<MF, OPS><<<256, 1024>>>(out, it, 1e-3f, 3); });
    printf("%-70s matrix %.3f ms, vector %.3f ms, both %.3f ms -> hidden %.0f %% of the shorter\n", name, tm, tv, tb,
           100.0 * (tm + tv - tb) / (tm < tv ? tm : tv));
}

int main()
{
    float *out;
    (void)hipMalloc(&out, sizeof(float) * 1024 * 256);
    go<0, 0>("same A / B for every MFMA, vector ops read two registers:", out);
    go<0, 1>("same A / B for every MFMA, vector ops read three registers:", out);
    go<1, 0>("three A / twelve B operands, vector ops read two registers:", out);
    go<1, 1>("three A / twelve B operands, vector ops read three registers:", out);
    go<2, 0>("three A / twelve B operands in AccVGPRs, vector ops read two registers:", out);
    go<2, 1>("three A / twelve B operands in AccVGPRs, vector ops read three registers:", out);
    return 0;
}
