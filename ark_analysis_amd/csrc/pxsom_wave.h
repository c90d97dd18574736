// pxsom_wave.h -- wave64 cross-lane helpers for gfx950 (DPP inside rows of 16 lanes, v_permlane16/32_swap
// across rows): no LDS crossbar round trips on latency-critical paths.
#pragma once
#include <hip/hip_runtime.h>

namespace pxsom {

// wave-wide minimum of a non-NaN double, result in every lane.  Rows of 16 lanes reduce with DPP
// (quad_perm xor1 / xor2, row_half_mirror, row_mirror), rows combine with v_permlane16/32_swap:
// no LDS crossbar round trips on the per-step critical path.
__device__ __forceinline__ double dpp_f64(double v, int ctrl_sel)
{
    int lo = (int)__double_as_longlong(v), hi = (int)(__double_as_longlong(v) >> 32);
    switch (ctrl_sel) {
        case 0: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); break;   // quad_perm [1,0,3,2]
        case 1: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, false); break;   // quad_perm [2,3,0,1]
        case 2: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, false); break; // row_half_mirror
        default: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, false); break; // row_mirror
    }
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double wave_min_f64(double v)
{
    v = fmin(v, dpp_f64(v, 0));
    v = fmin(v, dpp_f64(v, 1));
    v = fmin(v, dpp_f64(v, 2));
    v = fmin(v, dpp_f64(v, 3));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)(__double_as_longlong(v) >> 32);
    u2 rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = fmin(__longlong_as_double(((long long)rh[0] << 32) | rl[0]), __longlong_as_double(((long long)rh[1] << 32) | rl[1]));
    lo = (unsigned)__double_as_longlong(v);
    hi = (unsigned)(__double_as_longlong(v) >> 32);
    rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return fmin(__longlong_as_double(((long long)rh[0] << 32) | rl[0]), __longlong_as_double(((long long)rh[1] << 32) | rl[1]));
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    // v_min_u32 with the DPP operand fused (the compiler emits mov + mov_dpp + min per stage)
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "+v"(v));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = min(r[0], r[1]);
    r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return min(r[0], r[1]);
}

// value of the lane to the left inside a 16-lane row (row_shr:1); lane 0 of a row keeps its own
__device__ __forceinline__ double shr1_f64(double v)
{
    int lo = (int)__double_as_longlong(v), hi = (int)(__double_as_longlong(v) >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x111, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x111, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane((int)__double_as_longlong(v), l);
    const int hi = __builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

}  // namespace pxsom
