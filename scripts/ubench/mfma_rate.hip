// Micro-benchmark: v_mfma_f32_16x16x32_f16 issue rate on gfx950 under different dependency patterns.
// build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PATTERN>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    half8 a[4], b[4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) {
            a[i][j] = (_Float16)(seed * (threadIdx.x + i + j));
            b[i][j] = (_Float16)(seed * (threadIdx.x * 3 + i * 7 + j));
        }
    f32x4 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = f32x4{seed, 0.f, 0.f, 0.f};
    float m = 0.f;
    for (int it = 0; it < iters; it++) {
        if (PATTERN == 0) {  // 8 independent accumulators, each updated once per iteration
#pragma unroll
            for (int i = 0; i < 8; i++)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i + 1) & 3], acc[i], 0, 0, 0);
        } else if (PATTERN == 1) {  // one serial chain
#pragma unroll
            for (int i = 0; i < 8; i++)
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i + 1) & 3], acc[0], 0, 0, 0);
        } else if (PATTERN == 2) {  // two interleaved chains (the filter kernel's pattern)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i + 1) & 3], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i + 2) & 3], acc[1], 0, 0, 0);
            }
        } else if (PATTERN == 3) {  // fresh C operand (not the destination) at the head of each 3-chain
#pragma unroll
            for (int i = 0; i < 2; i++) {
                f32x4 t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[i], acc[4], 0, 0, 0);
                f32x4 t1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[i + 2], acc[5], 0, 0, 0);
                t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[i], t0, 0, 0, 0);
                t1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[i + 2], t1, 0, 0, 0);
                m += t0[0] + t1[0];
            }
        } else if (PATTERN == 5 || PATTERN == 6) {  // 4 sequential 3-chains; head C = 0 (5) or a VGPR (6)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f32x4 c0 = (PATTERN == 5) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[4 + i];
                f32x4 t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[i], c0, 0, 0, 0);
                t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[i], t0, 0, 0, 0);
                t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2], b[i], t0, 0, 0, 0);
                m = __builtin_amdgcn_fmed3f(m, t0[0], t0[1]);
            }
        } else if (PATTERN == 7) {  // 2 x (two interleaved 3-chains), head C = 0
#pragma unroll
            for (int i = 0; i < 2; i++) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                f32x4 t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[i], z, 0, 0, 0);
                f32x4 t1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[i + 2], z, 0, 0, 0);
                t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[i], t0, 0, 0, 0);
                t1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b[i + 2], t1, 0, 0, 0);
                t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2], b[i], t0, 0, 0, 0);
                t1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2], b[i + 2], t1, 0, 0, 0);
                m = __builtin_amdgcn_fmed3f(m, t0[0], t1[1]);
            }
        } else if (PATTERN >= 10 && PATTERN < 20) {  // 4 seq 3-chains + N independent VALU after each MFMA
            constexpr int NV = PATTERN - 10;
            float v0 = m, v1 = m + 1.f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f32x4 t0 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 3; kk++) {
                    t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk], b[i], t0, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NV; j++) {
                        if (j & 1) v0 = __builtin_amdgcn_fmed3f(v0, seed + j, v1);
                        else v1 = __builtin_amdgcn_fmed3f(v1, seed + j, v0 + 0.f);
                    }
                }
                acc[i] = t0;
            }
            m = v0 + v1;
        } else if (PATTERN >= 20 && PATTERN < 30) {  // 32x32x16 f16: 2 seq 6-chains + N VALU after each
            constexpr int NV = PATTERN - 20;
            float v0 = m, v1 = m + 1.f;
#pragma unroll
            for (int i = 0; i < 2; i++) {
                f32x16 t0;
#pragma unroll
                for (int e = 0; e < 16; e++) t0[e] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 6; kk++) {
                    t0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk & 3], b[i], t0, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NV; j++) {
                        if (j & 1) v0 = __builtin_amdgcn_fmed3f(v0, seed + j, v1);
                        else v1 = __builtin_amdgcn_fmed3f(v1, seed + j, v0 + 0.f);
                    }
                }
                acc[i][0] += t0[0] + t0[5] + t0[10] + t0[15];
            }
            m = v0 + v1;
        } else if (PATTERN >= 30 && PATTERN < 40) {  // 16x16 3-chains + N VOP2 adds (not VOP3)
            constexpr int NV = PATTERN - 30;
            float v0 = m, v1 = m + 1.f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f32x4 t0 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 3; kk++) {
                    t0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk], b[i], t0, 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < NV; j++) {
                        if (j & 1) v0 = __uint_as_float(__float_as_uint(v0) + __float_as_uint(v1));
                        else v1 = __uint_as_float(__float_as_uint(v1) ^ __float_as_uint(v0));
                    }
                }
                acc[i] = t0;
            }
            m = v0 + v1;
        } else if (PATTERN == 4) {  // PATTERN 0 + 5 independent VALU ops per MFMA
#pragma unroll
            for (int i = 0; i < 8; i++) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i + 1) & 3], acc[i], 0, 0, 0);
                float v = m;
#pragma unroll
                for (int j = 0; j < 5; j++) v = __builtin_amdgcn_fmed3f(v, seed + j, m + j);
                m = v;
            }
        }
    }
    float s = m;
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int P>
void run(const char *name, int blocks_per_cu, int mfma_per_iter, int valu_per_iter)
{
    int cus = 256, iters = 20000;
    float *out;
    hipMalloc(&out, sizeof(float) * cus * 8 * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<P><<<cus * blocks_per_cu, 256>>>(out, 100, 1e-3f);
    hipEventRecord(e0);
    k<P><<<cus * blocks_per_cu, 256>>>(out, iters, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: blocks_per_cu waves (4 waves per block, one per SIMD)
    double mfma_per_simd = (double)iters * mfma_per_iter * blocks_per_cu;
    printf("%-34s waves/SIMD=%d  %.3f ms  %.1f ns per MFMA per SIMD (= %.1f cycles @2.4GHz), valu/mfma=%d\n", name,
           blocks_per_cu, ms, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, valu_per_iter);
    hipFree(out);
}

int main()
{
    for (int w = 2; w <= 2; w++) {
        run<0>("8 independent accumulators", w, 8, 0);
        run<1>("1 serial chain", w, 8, 0);
        run<2>("2 interleaved chains", w, 8, 0);
        run<3>("fresh-C 2x2 chains + VALU read", w, 8, 0);
        run<4>("8 independent + 5 VALU each", w, 8, 5);
        run<5>("4 seq 3-chains, head C=0", w, 12, 0);
        run<6>("4 seq 3-chains, head C=VGPR", w, 12, 0);
        run<7>("2x2 interleaved 3-chains, C=0", w, 12, 0);
        run<10>("3-chains + 0 VALU/MFMA", w, 12, 0);
        run<12>("3-chains + 2 VALU/MFMA", w, 12, 2);
        run<14>("3-chains + 4 VALU/MFMA", w, 12, 4);
        run<16>("3-chains + 6 VALU/MFMA", w, 12, 6);
        run<18>("3-chains + 8 VALU/MFMA", w, 12, 8);
        run<20>("32x32x16 6-chains + 0 VALU", w, 12, 0);
        run<24>("32x32x16 6-chains + 4 VALU", w, 12, 4);
        run<28>("32x32x16 6-chains + 8 VALU", w, 12, 8);
        run<34>("3-chains + 4 VOP2 int ops", w, 12, 4);
        run<38>("3-chains + 8 VOP2 int ops", w, 12, 8);
    }
    return 0;
}
