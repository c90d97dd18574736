// pxsom_api.hip -- ABI bookkeeping + host-only helpers of libpxsom.so.
#include <cstring>

#include "pxsom_common.h"

namespace pxsom {

char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

int device_cu_count()
{
    static int cached = 0;
    if (cached > 0) return cached;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return 256;
    cached = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return cached;
}

}  // namespace pxsom

PXSOM_EXPORT int pxsom_abi_version(void) { return PXSOM_ABI_VERSION; }

PXSOM_EXPORT const char *pxsom_last_error(void) { return pxsom::err_buf(); }

// glibc rand(): TYPE_3 additive feedback generator (r[i] = r[i-31] + r[i-3], output >> 1),
// seeded by the minimal-standard LCG, first 310 outputs discarded.
PXSOM_EXPORT int pxsom_host_glibc_rand_fill(uint32_t seed, int64_t count, int32_t *out)
{
    if (count < 0 || (count > 0 && !out))
        return pxsom::fail(PXSOM_ERR_INVALID_ARG, "glibc_rand_fill: bad count/out");
    uint32_t st[31];
    int64_t word = seed == 0 ? 1 : (int64_t)(int32_t)seed;
    st[0] = (uint32_t)word;
    for (int i = 1; i < 31; i++) {
        int64_t hi = word / 127773, lo = word % 127773;
        word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        st[i] = (uint32_t)word;
    }
    int f = 3, b = 0;
    for (int i = 0; i < 310; i++) {
        st[f] += st[b];
        f = f == 30 ? 0 : f + 1;
        b = b == 30 ? 0 : b + 1;
    }
    for (int64_t i = 0; i < count; i++) {
        st[f] += st[b];
        out[i] = (int32_t)(st[f] >> 1);
        f = f == 30 ? 0 : f + 1;
        b = b == 30 ? 0 : b + 1;
    }
    return PXSOM_OK;
}
