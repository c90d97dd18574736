// pxsom_api.hip -- ABI bookkeeping + host-only helpers of libpxsom.so.
#include <cstring>
#include <utility>
#include <vector>

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
    static PerDevice<int> counts;
    int &cached = counts.here();
    if (cached > 0) return cached;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
        return 256;
    cached = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return cached;
}

static thread_local RowView g_row_view;
const RowView &current_row_view() { return g_row_view; }
RowView make_row_view(int gw, int64_t gstride)
{
    RowView v;
    if (gw > 1) {
        const unsigned w = (unsigned)gw;
        int s = 0;
        while (((uint64_t)1 << s) < w) s++;
        v.gw = gw;
        v.gstride = gstride;
        v.magic = (unsigned)((((uint64_t)1 << (31 + s)) / w) + 1u);
        v.shift = s - 1;
    }
    return v;
}
RowViewScope::RowViewScope(const RowView &v) : saved(g_row_view) { g_row_view = v; }
RowViewScope::~RowViewScope() { g_row_view = saved; }

struct Prof {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
    int64_t min_rows = 0;
    bool open = false, taken = false;
};

static thread_local Prof *g_prof = nullptr;

Prof *current_prof() { return g_prof; }

void prof_mark(Prof *p, hipStream_t st, bool start, int64_t rows)
{
    if (!p || rows < p->min_rows) return;
    if (start) {
        if (p->used == p->ev.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            p->ev.emplace_back(a, b);
        }
        p->open = true;    // armed: the launch between the two marks takes the pair (prof_take)
        p->taken = false;
    } else if (p->open) {
        if (p->taken) p->used++;
        p->open = false;
    }
    (void)st;
}

bool prof_take(hipEvent_t *start, hipEvent_t *stop)
{
    Prof *p = g_prof;
    if (!p || !p->open || p->taken) return false;
    *start = p->ev[p->used].first;
    *stop = p->ev[p->used].second;
    p->taken = true;
    return true;
}

}  // namespace pxsom

PXSOM_EXPORT int pxsom_prof_create(void **out)
{
    if (!out) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_prof_create: null");
    *out = new pxsom::Prof();
    return PXSOM_OK;
}

PXSOM_EXPORT int pxsom_prof_destroy(void *h)
{
    auto *p = reinterpret_cast<pxsom::Prof *>(h);
    if (!p) return PXSOM_OK;
    if (pxsom::g_prof == p) pxsom::g_prof = nullptr;
    for (auto &e : p->ev) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    delete p;
    return PXSOM_OK;
}

PXSOM_EXPORT int pxsom_prof_attach(void *h, int64_t min_rows)
{
    auto *p = reinterpret_cast<pxsom::Prof *>(h);
    pxsom::g_prof = p;
    if (p) {
        p->min_rows = min_rows;
        p->used = 0;
        p->open = false;
    }
    return PXSOM_OK;
}

PXSOM_EXPORT int pxsom_prof_collect(void *h, double *total_ms, int64_t *launches)
{
    auto *p = reinterpret_cast<pxsom::Prof *>(h);
    if (!p || !total_ms || !launches) return pxsom::fail(PXSOM_ERR_INVALID_ARG, "pxsom_prof_collect: null");
    double tot = 0.0;
    for (size_t i = 0; i < p->used; i++) {
        PXSOM_HIP_TRY(hipEventSynchronize(p->ev[i].second));
        float ms = 0.f;
        PXSOM_HIP_TRY(hipEventElapsedTime(&ms, p->ev[i].first, p->ev[i].second));
        tot += ms;
    }
    *total_ms = tot;
    *launches = (int64_t)p->used;
    p->used = 0;
    return PXSOM_OK;
}

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
