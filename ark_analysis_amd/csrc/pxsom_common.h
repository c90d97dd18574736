// pxsom_common.h -- shared host-side plumbing for libpxsom.so (gfx950 only; no other target).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "pxsom.h"

#define PXSOM_EXPORT extern "C" __attribute__((visibility("default")))

namespace pxsom {

// thread-local last-error text (pxsom_last_error)
char *err_buf();
int fail(int code, const char *fmt, ...);

inline int hip_fail(hipError_t e, const char *what)
{
    return fail(PXSOM_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

#define PXSOM_HIP_TRY(expr)                                          \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) return ::pxsom::hip_fail(_e, #expr);   \
    } while (0)

// launch check: kernels are async, this only catches configuration errors
#define PXSOM_LAUNCH_CHECK(name)                                     \
    do {                                                             \
        hipError_t _e = hipGetLastError();                           \
        if (_e != hipSuccess) return ::pxsom::hip_fail(_e, name);    \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline bool dtype_ok(int dtype) { return dtype == PXSOM_F32 || dtype == PXSOM_F64 || dtype == PXSOM_F16; }

// `return CALL<T>(static_cast<const T *>(ptr), ...)` for the pixel-matrix dtype (validated by the caller)
#define PXSOM_DISPATCH_DTYPE(dtype, ptr, XP, CALL)                      \
    do {                                                                \
        if ((dtype) == PXSOM_F32) {                                     \
            typedef float T;                                            \
            const T *XP = reinterpret_cast<const T *>(ptr);             \
            return CALL;                                                \
        }                                                               \
        if ((dtype) == PXSOM_F16) {                                     \
            typedef _Float16 T;                                         \
            const T *XP = reinterpret_cast<const T *>(ptr);             \
            return CALL;                                                \
        }                                                               \
        typedef double T;                                               \
        const T *XP = reinterpret_cast<const T *>(ptr);                 \
        return CALL;                                                    \
    } while (0)

// number of CUs of the current device (256 on MI355X); cached per device
int device_cu_count();

// One slot per device for what the launch code caches (occupancy, raised LDS limits, CU counts): a process that drives
// several GPUs -- not this build's usual one process per GPU -- gets every device's own values.
template <typename T>
struct PerDevice {
    static constexpr int kSlots = 64;
    T slot[kSlots] = {};
    T &here()
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kSlots) dev = 0;
        return slot[dev];
    }
};

// Rows of a scheduled mini-batch step where they lie in the caller's matrix (round 6: the generic training route's kernels read
// them there; until then every pass began by gathering each step's rows into a copy -- 217 us of config 4's 2.0 ms pass, 800 MB
// of traffic).  Row f of the step starts at element (f / gw) * gstride + (f % gw) * ldx: gw consecutive rows out of every
// `phases`; gw <= 1: the plain strided matrix, f * ldx.  A view is set for the calling thread around the launches that take it
// (RowViewScope); a launch that reads rows and does NOT take views must refuse to run under one (row_view_active()).
struct RowView {
    int gw = 1;
    int64_t gstride = 0;
    unsigned magic = 0;   // f / gw == __umulhi(f, magic) >> shift for f < 2^31 (as StepArgs::group_magic)
    int shift = 0;
    __device__ __forceinline__ int64_t offset(int64_t f, int64_t ldx) const
    {
        if (gw <= 1) return f * ldx;
        const unsigned grp = __umulhi((unsigned)f, magic) >> shift, sub = (unsigned)f - grp * (unsigned)gw;
        return (int64_t)grp * gstride + (int64_t)sub * ldx;
    }
};
RowView make_row_view(int gw, int64_t gstride);
const RowView &current_row_view();
inline bool row_view_active() { return current_row_view().gw > 1; }
struct RowViewScope {
    RowView saved;
    explicit RowViewScope(const RowView &v);
    ~RowViewScope();
};

// Optional in-library kernel timer (pxsom_prof_*): HIP event pairs recorded on the launch stream
// immediately around the dominant kernel of a call, so a caller can report that kernel's
// duration without a profiler attached.
struct Prof;
Prof *current_prof();
void prof_mark(Prof *p, hipStream_t st, bool start, int64_t rows);
// Round 6: the timer's event pair rides ON the dispatch it times (hipExtLaunchKernelGGL's start / stop events: the kernel's own
// begin and end, what rocprofv3's kernel trace reports) instead of in two marker packets around it -- those made the stream wait
// at both ends of the launch (~8 us around a 0.21 ms kernel, inside the bench's timed region).  prof_mark(start) arms the pair,
// the launch between the two marks takes it with PXSOM_TIMED_LAUNCH, prof_mark(end) counts it if it was taken.
bool prof_take(hipEvent_t *start, hipEvent_t *stop);
#define PXSOM_TIMED_LAUNCH(kern, grid, block, lds, st, ...)                                           \
    do {                                                                                              \
        hipEvent_t pxsom_e0_, pxsom_e1_;                                                              \
        if (pxsom::prof_take(&pxsom_e0_, &pxsom_e1_))                                                 \
            hipExtLaunchKernelGGL(kern, grid, block, lds, st, pxsom_e0_, pxsom_e1_, 0, __VA_ARGS__);  \
        else                                                                                          \
            hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                              \
    } while (0)

}  // namespace pxsom
