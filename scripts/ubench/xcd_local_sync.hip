// Can the BMU-only tail of a batch training pass run as ONE persistent launch whose workgroups all sit on one XCD and
// synchronise through that XCD's L2 alone?  (MI355X: 8 XCDs, each with its own L2; agent-scope traffic goes past it.)
//
// Protocol under test (placement-independent: HIP promises nothing about which XCD a workgroup lands on):
//   * every workgroup reads HW_REG_XCC_ID and takes a ticket on a device-scope counter; ticket 0 is the LEADER, its XCD is
//     the chosen one (published with a device-scope store);
//   * workgroups on other XCDs leave; workgroups on the chosen XCD register with one returning device-scope atomic on a
//     member word; the leader CLOSES registration (atomic OR of a flag bit) once all gridDim.x tickets are out or after a
//     bounded wait -- a workgroup that registers after the close leaves as well: no spin in the kernel is unbounded by
//     something that is not running;
//   * from then on P members, rank r: per step every member adds its table into a shared statistics buffer with
//     L2-LOCAL atomics (no sc1: executed in the XCD's L2), clears its slice of the next buffer, waits for its own
//     memory operations (vmcnt(0)), arrives on a monotonic counter (L2-local atomic) and polls it with L1-bypassing
//     loads (sc1: served by the L2); then reads the statistics with L1-bypassing loads and checks every word.
// Output: members, step time with / without the statistics traffic, the same with agent-scope (sc1) atomics, errors.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kThreads = 512;
constexpr int kStats = 2300;          // K * (C + 1) doubles
constexpr unsigned kClosed = 0x80000000u;

struct Ctl {
    unsigned tickets;      // device scope
    unsigned chosen;       // xcc + 1 of the leader (0: not yet known)
    unsigned members;      // registrations | kClosed
    unsigned pad0[13];
    unsigned arrive;       // L2-local barrier counter (its own 64 bytes)
    unsigned pad1[15];
    unsigned errors;
    unsigned p_out, xcc_out;
    unsigned pad2[13];
    long long t0, t1;
    unsigned flags[64];    // flag barrier: one word per member (two 128-byte lines)
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf; }   // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ long long wall_clock64x() { return (long long)wall_clock64(); }   // s_memrealtime, 100 MHz

__device__ __forceinline__ unsigned ld_l2(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_l2(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool AGENT>
__device__ __forceinline__ void add_f64(double *p, double v)
{
    if (AGENT) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else asm volatile("global_atomic_add_f64 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
template <bool AGENT>
__device__ __forceinline__ void add_u32(unsigned *p, unsigned v)
{
    if (AGENT) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else asm volatile("global_atomic_add %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

// mode 0: barrier only; 1: + statistics traffic and check
template <bool AGENT>
__global__ __launch_bounds__(kThreads) void tail_probe(Ctl *ctl, double *ring, int steps, int mode, int spin_limit, int all_xcc)
{
    extern __shared__ char smem[];
    __shared__ unsigned s_rank, s_p;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const unsigned xcc = xcc_id();
        const unsigned t = __hip_atomic_fetch_add(&ctl->tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned rank = 0xffffffffu, p = 0;
        if (t == 0) {   // leader: its XCD is the chosen one; it is member 0
            __hip_atomic_store(&ctl->chosen, xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            rank = __hip_atomic_fetch_add(&ctl->members, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(&ctl->tickets, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && spins++ < spin_limit)
                __builtin_amdgcn_s_sleep(2);
            // everybody has a ticket -- but a member may still be between its ticket and its registration: wait for the
            // registrations of the tickets seen (bounded), then close
            spins = 0;
            while (spins++ < 64) __builtin_amdgcn_s_sleep(1);
            p = __hip_atomic_fetch_or(&ctl->members, kClosed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned ch;
            int spins = 0;
            while ((ch = __hip_atomic_load(&ctl->chosen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && spins++ < (1 << 20))
                __builtin_amdgcn_s_sleep(1);
            if (ch == xcc + 1 || all_xcc) {
                const unsigned old = __hip_atomic_fetch_add(&ctl->members, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!(old & kClosed)) {
                    rank = old;
                    unsigned m;
                    while (!((m = __hip_atomic_load(&ctl->members, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & kClosed))
                        __builtin_amdgcn_s_sleep(1);
                    // registrations after the close bump the word as well: P is what the leader's OR returned = the
                    // count at the close; late arrivals only add above it.  The leader publishes it:
                    while ((p = __hip_atomic_load(&ctl->p_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(1);
                }
            }
        }
        if (t == 0) {
            __hip_atomic_store(&ctl->xcc_out, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->p_out, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_rank = rank;
        s_p = p;
    }
    __syncthreads();
    const unsigned rank = s_rank, P = s_p;
    if (rank == 0xffffffffu) return;
    if (rank >= P) return;   // (cannot happen: ranks below the closing count are exactly the members)
    const long long t0 = wall_clock64x();
    unsigned errs = 0;
    for (int g = 0; g < steps; g++) {
        double *cur = ring + (size_t)(g % 3) * kStats, *nxt = ring + (size_t)((g + 1) % 3) * kStats;
        if (mode & 1) {
            for (int e = tid; e < kStats; e += kThreads) add_f64<AGENT>(cur + e, (double)((g & 7) + 1) * (double)(e % 5 + 1));
            const int per = (kStats + (int)P - 1) / (int)P;
            for (int e = rank * per + tid; e < min((int)(rank + 1) * per, kStats); e += kThreads) {
                if (AGENT) __hip_atomic_store(nxt + e, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else nxt[e] = 0.0;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (mode & 2) {   // flag barrier: no atomics -- a plain store per member, the first wave polls the P words
            if (tid == 0) {
                if (AGENT) __hip_atomic_store(&ctl->flags[rank], (unsigned)(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else ctl->flags[rank] = (unsigned)(g + 1);
            }
            if (tid < 64) {
                const unsigned want = (unsigned)(g + 1);
                for (;;) {
                    const unsigned f = tid < (int)P ? ld_l2(&ctl->flags[tid]) : want;
                    if (__ballot(f < want) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
        } else if (tid == 0) {
            add_u32<AGENT>(&ctl->arrive, 1u);
            const unsigned want = (unsigned)(g + 1) * P;
            while (ld_l2(&ctl->arrive) < want) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (mode & 1) {
            for (int e = tid; e < kStats; e += kThreads) {
                const double v = ld_l2(cur + e);
                if (v != (double)P * (double)((g & 7) + 1) * (double)(e % 5 + 1)) errs++;
            }
        }
    }
    const long long t1 = wall_clock64x();
    if (errs) atomicAdd(&ctl->errors, errs);
    if (rank == 0 && tid == 0) {
        ctl->t0 = t0;
        ctl->t1 = t1;
    }
}

template <bool AGENT>
void run(const char *name, int grid, int steps, int mode, size_t lds, int all_xcc = 0)
{
    Ctl *ctl;
    double *ring;
    hipMalloc(&ctl, sizeof(Ctl));
    hipMalloc(&ring, 3 * kStats * 8);
    hipFuncSetAttribute((const void *)tail_probe<AGENT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(ctl, 0, sizeof(Ctl));
        hipMemset(ring, 0, 3 * kStats * 8);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(tail_probe<AGENT>, dim3(grid), dim3(kThreads), lds, 0, ctl, ring, steps, mode, 1 << 14, all_xcc);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        Ctl h;
        hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
        printf("%-28s grid %3d mode %d: P %2u on xcc %u | %d steps: launch %.1f us, %.3f us/step (host), %.3f us/step (100 MHz clock) | errors %u\n",
               name, grid, mode, h.p_out, h.xcc_out, steps, ms * 1e3, ms * 1e3 / steps, (double)(h.t1 - h.t0) * 0.01 / steps, h.errors);
    }
    hipFree(ctl);
    hipFree(ring);
}

int main()
{
    const size_t lds = 81 * 1024;   // one workgroup per CU
    for (int grid : {256, 128, 64}) {
        run<false>("L2-local barrier only", grid, 2000, 0, lds);
        run<false>("L2-local + statistics", grid, 2000, 1, lds);
        run<false>("flag barrier only", grid, 2000, 2, lds);
        run<false>("flag barrier + statistics", grid, 2000, 3, lds);
    }
    run<true>("agent-scope barrier only", 256, 2000, 0, lds);
    run<true>("agent-scope + statistics", 256, 2000, 1, lds);
    run<false>("L2-local + stats, 2 WG/CU", 512, 2000, 1, 40 * 1024);
    run<true>("agent-scope, ALL XCDs, barrier", 256, 2000, 0, lds, 1);
    run<true>("agent-scope, ALL XCDs, +stats", 256, 2000, 1, lds, 1);
    run<true>("agent-scope, ALL XCDs, 64 WGs", 64, 2000, 1, lds, 1);
    return 0;
}
