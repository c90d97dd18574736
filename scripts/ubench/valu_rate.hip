// Issue rate of single VALU instructions on gfx950 (clocks per wave64 instruction per SIMD), measured with NW waves per SIMD
// each running 8 independent dependency chains of one opcode.  Round 5: what do v_cvt_f64_f32 / v_add_f64 -- the fixed-point
// conversion of the one-pass kernel's table adds -- cost next to the 32-bit integer / binary32 alternatives?
#include <hip/hip_runtime.h>
#include <cstdio>
#define OPS(X)                                                                                                   \
    X(0, "v_add_f32", "v_add_f32 %0, %0, %1", 1)                                                                 \
    X(1, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, 3", 1)                                                        \
    X(2, "v_max3_f32", "v_max3_f32 %0, %0, %1, %1", 1)                                                           \
    X(3, "v_cvt_f64_f32", "v_cvt_f64_f32 %0, %2", 2)                                                             \
    X(4, "v_add_f64", "v_add_f64 %0, %0, %3", 2)                                                                 \
    X(5, "v_fma_f64", "v_fma_f64 %0, %0, %3, %3", 2)                                                             \
    X(6, "v_cvt_i32_f32", "v_cvt_i32_f32 %0, %0", 1)                                                             \
    X(7, "v_mul_f32", "v_mul_f32 %0, %0, %1", 1)                                                                 \
    X(8, "v_ashrrev_i32", "v_ashrrev_i32 %0, 31, %0", 1)                                                         \
    X(9, "v_cvt_f64_i32", "v_cvt_f64_i32 %0, %2", 2)                                                             \
    X(10, "v_lshlrev_b64", "v_lshlrev_b64 %0, 3, %0", 2)                                                         \
    X(11, "v_mad_i64_i32", "v_mad_i64_i32 %0, vcc, %2, %2, %0", 2)                                               \
    X(12, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1", 1)                                                          \
    X(13, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1", 1)                                                        \
    X(14, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %3, %3", 2)                                                      \
    X(15, "v_cvt_pk_f16_f32", "v_cvt_pk_f16_f32 %0, %0, %1", 1)                                                  \
    X(16, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 3, %1", 1)                                                   \
    X(17, "v_add_co_u32", "v_add_co_u32 %0, vcc, %0, %1", 1)                                                     \
    X(18, "v_cvt_u32_f32", "v_cvt_u32_f32 %0, %0", 1)                                                            \
    X(19, "v_ldexp_f32", "v_ldexp_f32 %0, %0, %1", 1)                                                            \
    X(20, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc", 1)                                                   \
    X(21, "v_permlane16_swap", "v_permlane16_swap_b32 %0, %1", 1)                                                \
    X(22, "v_med3_f32", "v_med3_f32 %0, %0, %1, %1", 1)                                                          \
    X(23, "v_dot2c_f32_f16", "v_dot2c_f32_f16 %0, %1, %1", 1)                                                    \
    X(24, "v_max_f32", "v_max_f32 %0, %0, %1", 1)                                                                \
    X(25, "v_min_f32", "v_min_f32 %0, %0, %1", 1)                                                                \
    X(26, "v_and_b32", "v_and_b32 %0, %0, %1", 1)                                                                \
    X(27, "v_or_b32", "v_or_b32 %0, %0, %1", 1)                                                                  \
    X(28, "v_xor_b32", "v_xor_b32 %0, %0, %1", 1)                                                                \
    X(29, "v_add_u32", "v_add_u32 %0, %0, %1", 1)                                                                \
    X(30, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0", 1)                                                         \
    X(31, "v_fmac_f32", "v_fmac_f32 %0, %1, %1", 1)                                                              \
    X(32, "v_fma_f32", "v_fma_f32 %0, %0, %1, %1", 1)                                                            \
    X(33, "v_mov_b32", "v_mov_b32 %0, %1", 1)                                                                    \
    X(34, "v_cndmask_e64_sgpr", "v_cndmask_b32_e64 %0, %0, %1, s[10:11]", 1)                                      \
    X(35, "v_cmp_gt_f32_e64", "v_cmp_gt_f32_e64 s[10:11], %0, %1", 1)                                            \
    X(36, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 28", 1)                                                             \
    X(37, "v_bfi_b32", "v_bfi_b32 %0, %1, %0, %1", 1)                                                            \
    X(38, "v_perm_b32", "v_perm_b32 %0, %0, %1, %1", 1)                                                          \
    X(39, "v_permlane32_swap", "v_permlane32_swap_b32 %0, %1", 1)                                                \
    X(40, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %0", 1)                                                            \
    X(41, "v_sqrt_f32", "v_sqrt_f32 %0, %0", 1)                                                                  \
    X(42, "v_mov_dpp_quad", "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 1)            \
    X(43, "v_add_f32_dpp", "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", 1)          \
    X(44, "v_readlane", "v_readlane_b32 s12, %0, 3", 1)                                                          \
    X(45, "v_sub_f32", "v_sub_f32 %0, %0, %1", 1)                                                                \
    X(46, "v_mul_f64", "v_mul_f64 %0, %0, %3", 2)                                                                \
    X(47, "v_pk_add_f32", "v_pk_add_f32 %0, %0, %3", 2)                                                          \
    X(48, "v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %3", 2)                                                          \
    X(49, "v_max3_f32_e", "v_max3_f32 %0, %0, %1, 1.0", 1)                                                       \
    X(50, "v_add3_u32", "v_add3_u32 %0, %0, %1, %1", 1)                                                          \
    X(51, "v_or3_b32", "v_or3_b32 %0, %0, %1, %1", 1)                                                            \
    X(52, "v_cndmask_vcc_set", "v_cndmask_b32 %0, %0, %1, vcc", 1)
template <int OP>
__global__ __launch_bounds__(1024) void k(float *out, int iters, float seed)
{
    float a[8];
    double d[8];
    for (int i = 0; i < 8; i++) {
        a[i] = seed * (threadIdx.x + i);
        d[i] = (double)a[i];
    }
    float y = seed * 3.f;
    double z = (double)seed;
    asm volatile("s_mov_b64 s[10:11], 0x5555\n\ts_mov_b64 vcc, 0x3333" ::: "s10", "s11", "vcc");
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
#define X(id, name, text, w)                                                                       \
    if (OP == id) {                                                                                \
        if (w == 1) asm volatile(text : "+v"(a[i]) : "v"(y), "v"(y), "v"(z) : "s10", "s11", "s12", "vcc");                       \
        else asm volatile(text : "+v"(d[i]) : "v"(y), "v"(a[i]), "v"(z) : "s10", "s11", "s12", "vcc");                           \
    }
                OPS(X)
#undef X
            }
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += a[i] + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(t1 - t0) * 1e-30f;
}
template <int OP>
void run(const char *name, float *out)
{
    const int iters = 4000;
    for (int nw : {1, 2, 4}) {   // waves per SIMD
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<OP><<<256, 256 * nw>>>(out, 10, 1.0f);
        (void)hipEventRecord(e0);
        k<OP><<<256, 256 * nw>>>(out, iters, 1.0f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        // per SIMD: nw waves x iters x 32 instructions
        const double ns_per_instr = ms * 1e6 / ((double)nw * iters * 32);
        printf("%-20s %d wave(s)/SIMD: %.2f ns per wave-instruction per SIMD (%.1f clk at 2.4 GHz)\n", name, nw, ns_per_instr, ns_per_instr * 2.4);
    }
}
int main()
{
    float *out;
    (void)hipMalloc(&out, 4 * 256 * 1024);
#define X(id, name, text, w) run<id>(name, out);
    OPS(X)
#undef X
    return 0;
}
