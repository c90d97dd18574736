// Do matrix and vector instructions of DIFFERENT waves on one SIMD overlap on gfx950, and does the MFMA shape matter?
// (Round 1 measured ~13 % for v_mfma_f32_16x16x32_f16 against v_med3_f32 chains, scripts/ubench/overlap.hip: the filters' time is
// the SUM of their MFMA and VALU time.  If a longer-running shape left the issue port free, a 32 x 32 tiling would hide one in the other.)
// One workgroup of 512 threads per CU: waves 0-3 and 4-7 land on SIMDs 0-3 pairwise; waves 0-3 run the matrix loop, waves 4-7 the
// vector loop.  Three launches per shape: matrix waves alone, vector waves alone, both.  Also one wave doing both, interleaved.
//   build: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__device__ __forceinline__ float matrix_loop(int iters, float seed)
{
    half8 a, b;
    half4 a4, b4;
    for (int j = 0; j < 8; j++) {
        a[j] = (_Float16)(seed * (threadIdx.x + j));
        b[j] = (_Float16)(seed * (threadIdx.x * 3 + j));
    }
    for (int j = 0; j < 4; j++) {
        a4[j] = a[j];
        b4[j] = b[j];
    }
    float r = 0.f;
    if constexpr (SHAPE == 0 || SHAPE == 2) {   // 16x16x32 f16 (4 independent chains) / 16x16x16 f16
        f32x4 acc[4];
        for (int i = 0; i < 4; i++) acc[i] = f32x4{seed, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (SHAPE == 0) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i & 3], 0, 0, 0);
                else acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i & 3], 0, 0, 0);
            }
        for (int i = 0; i < 4; i++) r += acc[i][0];
    } else {                                   // 32x32x16 f16 (2 independent chains) / 32x32x8 f16
        f32x16 acc[2];
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 16; j++) acc[i][j] = seed * j;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if constexpr (SHAPE == 1) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 1], 0, 0, 0);
                else acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc[i & 1], 0, 0, 0);
            }
        for (int i = 0; i < 2; i++) r += acc[i][0];
    }
    return r;
}

template <int VOP>
__device__ __forceinline__ float vector_loop(int iters, float seed)
{
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed * (threadIdx.x + i);
    const float y = seed * 3.f;
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int rpt = 0; rpt < 4; rpt++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (VOP == 0) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(y));
                else if constexpr (VOP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(y));
                else asm volatile("v_and_or_b32 %0, %0, %1, 3" : "+v"(v[i]) : "v"(y));
            }
    float r = 0.f;
    for (int i = 0; i < 8; i++) r += v[i];
    return r;
}

// mode 1: matrix waves only, 2: vector waves only, 3: both
template <int SHAPE, int VOP>
__global__ __launch_bounds__(512) void split_roles(float *out, int iters_m, int iters_v, float seed, int mode)
{
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) r = matrix_loop<SHAPE>(iters_m, seed);
    } else {
        if (mode & 2) r = vector_loop<VOP>(iters_v, seed);
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

// one wave per SIMD doing both: per trip NM matrix instructions and NV vector instructions the compiler may interleave
template <int SHAPE, int NV>
__global__ __launch_bounds__(1024) void one_wave(float *out, int iters, float seed, int what)
{
    half8 a, b;
    for (int j = 0; j < 8; j++) {
        a[j] = (_Float16)(seed * (threadIdx.x + j));
        b[j] = (_Float16)(seed * (threadIdx.x * 3 + j));
    }
    f32x4 acc[4];
    f32x16 big[2];
    for (int i = 0; i < 4; i++) acc[i] = f32x4{seed, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 16; j++) big[i][j] = seed * j;
    float v[16];
    for (int i = 0; i < 16; i++) v[i] = seed * (threadIdx.x + i);
    const float y = seed * 3.f;
    for (int it = 0; it < iters; it++) {
        if (what & 1) {
            if constexpr (SHAPE == 0) {
#pragma unroll
                for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 2; i++) big[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, big[i], 0, 0, 0);
            }
        }
        if (what & 2) {
#pragma unroll
            for (int i = 0; i < NV; i++) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(v[i & 15]) : "v"(y));
        }
    }
    float r = 0.f;
    for (int i = 0; i < 4; i++) r += acc[i][0];
    for (int i = 0; i < 2; i++) r += big[i][0];
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

template <int SHAPE, int VOP>
void split(const char *shape, const char *vop, float *out, int flops_per_inst)
{
    const int im = SHAPE == 0 || SHAPE == 2 ? 4000 : 4000, iv = 8000;
    const float tm = timed([&] { split_roles<SHAPE, VOP><<<256, 512>>>(out, im, iv, 1e-3f, 1); });
    const float tv = timed([&] { split_roles<SHAPE, VOP><<<256, 512>>>(out, im, iv, 1e-3f, 2); });
    const float tb = timed([&] { split_roles<SHAPE, VOP><<<256, 512>>>(out, im, iv, 1e-3f, 3); });
    const int per_trip = SHAPE == 0 || SHAPE == 2 ? 8 : 4;
    printf("%-14s + %-12s: matrix alone %.3f ms (%.1f ns per instruction, %.0f TFLOP/s at one wave per SIMD), vector alone %.3f ms (%.2f ns per "
           "instruction), both %.3f ms -> hidden %.0f %% of the shorter\n",
           shape, vop, tm, tm * 1e6 / (im * per_trip), 1024.0 * flops_per_inst / (tm * 1e6 / (im * per_trip)) * 1e-3, tv, tv * 1e6 / (iv * 32.0), tb,
           100.0 * (tm + tv - tb) / (tm < tv ? tm : tv));
}

int main()
{
    float *out;
    (void)hipMalloc(&out, sizeof(float) * 1024 * 512);
    split<0, 0>("16x16x32 f16", "v_med3_f32", out, 16 * 16 * 32 * 2);
    split<0, 1>("16x16x32 f16", "v_add_f32", out, 16 * 16 * 32 * 2);
    split<0, 2>("16x16x32 f16", "v_and_or_b32", out, 16 * 16 * 32 * 2);
    split<1, 0>("32x32x16 f16", "v_med3_f32", out, 32 * 32 * 16 * 2);
    split<1, 1>("32x32x16 f16", "v_add_f32", out, 32 * 32 * 16 * 2);
    split<2, 0>("16x16x16 f16", "v_med3_f32", out, 16 * 16 * 16 * 2);
    split<3, 0>("32x32x8 f16", "v_med3_f32", out, 32 * 32 * 8 * 2);
    const int it = 4000;
    for (int threads : {256, 512, 1024})
        for (int shape = 0; shape < 2; shape++) {
            auto go = [&](int what, int nv) {
                return timed([&] {
                    if (shape == 0) {
                        if (nv == 8) one_wave<0, 8><<<256, threads>>>(out, it, 1e-3f, what);
                        else if (nv == 24) one_wave<0, 24><<<256, threads>>>(out, it, 1e-3f, what);
                        else one_wave<0, 48><<<256, threads>>>(out, it, 1e-3f, what);
                    } else {
                        if (nv == 8) one_wave<1, 8><<<256, threads>>>(out, it, 1e-3f, what);
                        else if (nv == 24) one_wave<1, 24><<<256, threads>>>(out, it, 1e-3f, what);
                        else one_wave<1, 48><<<256, threads>>>(out, it, 1e-3f, what);
                    }
                });
            };
            for (int nv : {8, 24, 48}) {
                const float tm = go(1, nv), tv = go(2, nv), tb = go(3, nv);
                printf("%d wave(s) per SIMD, every wave both: %s, trip = %d matrix + %d v_med3_f32: matrix %.3f ms, vector %.3f ms, both %.3f ms (%.2f ns per vector instruction and SIMD alone) -> "
                       "hidden %.0f %% of the shorter\n", threads / 256, shape == 0 ? "16x16x32 f16" : "32x32x16 f16", shape == 0 ? 4 : 2, nv, tm, tv, tb, tv * 1e6 / ((double)it * nv * (threads / 256)),
                       100.0 * (tm + tv - tb) / (tm < tv ? tm : tv));
            }
        }
    return 0;
}
