// Phase times inside the persistent BMU-only tail (pxsom_batch_tail.hip): s_memrealtime stamps of every member's thread 0
// in one chosen step; config 2's tail (16 steps over the last 1/6 of 1 048 576 x 22 rows: 15 x 8 738 + 43 690 rows).
#define PXSOM_TAIL_TIMING 1
#include "../../ark_analysis_amd/csrc/pxsom_batch_tail.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

int main(int argc, char **argv)
{
    const int c = 22, K = 100;
    const int64_t n = 1048576;
    std::vector<float> x((size_t)n * c);
    srand(1);
    std::vector<float> cen(32 * c);
    for (auto &v : cen) v = (float)rand() / RAND_MAX;
    for (int64_t i = 0; i < n; i++) {
        const int z = rand() % 32;
        for (int j = 0; j < c; j++) {
            float v = cen[z * c + j] + 0.05f * ((float)rand() / RAND_MAX - 0.5f) * 3.4f;
            x[(size_t)i * c + j] = (rand() % 10 == 0 || v < 0) ? 0.f : v;
        }
    }
    std::vector<double> w((size_t)K * c);
    for (int k = 0; k < K; k++)
        for (int j = 0; j < c; j++) w[(size_t)k * c + j] = x[(size_t)(k * 9973) * c + j];
    float *dx, *dmu; double *dwbuf, *dring; char *scratch;
    const size_t nw = (size_t)K * c, ns = (size_t)K * (c + 1);
    using namespace pxsom_bmu;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dwbuf, 2 * nw * 8); hipMalloc(&dring, 3 * ns * 8); hipMalloc(&dmu, 40 * 4);
    hipMalloc(&scratch, tail_scratch_bytes(c));
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dmu, 0, 160);
    const int stamp_step = argc > 1 ? atoi(argv[1]) : 8;
    hipMemcpyToSymbol(HIP_SYMBOL(g_tail_stamp_step), &stamp_step, sizeof(int));
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpy(dwbuf, w.data(), nw * 8, hipMemcpyHostToDevice);
        hipMemcpy(dwbuf + nw, w.data(), nw * 8, hipMemcpyHostToDevice);
        hipMemset(dring, 0, 3 * ns * 8);
        TailArgs ta;
        ta.nsteps = 16; ta.phases = 960; ta.first_has_update = 0; ta.final_update = 0; ta.q_final = 1.0; ta.sat_final = 0x1p62;
        ta.stats_first = dring; ta.w_in = dwbuf; ta.w_last = dwbuf + nw; ta.stats_last = dring + ns; ta.stats_zero = dring + 2 * ns;
        ta.w_final = nullptr; ta.scratch = scratch;
        ta.tol_rel = (float)(2.5 * (ldexp(1.0, -16) + (3.0 * c + 2.0) * ldexp(1.0, -24) + ldexp(1.0, -19) + ldexp(1.0, -23) + ldexp(1.0, -24)));
        ta.tol_abs = (float)(2.5 * ldexp(1.0, -24) * sqrt((double)c));
        ta.mu32 = dmu; ta.qmagic = 0.0;
        for (int s = 0; s < 16; s++) {
            ta.st[s].e0 = 800 + 8 * s; ta.st[s].width = s == 15 ? 40 : 8;
            ta.st[s].rows = (n / 960) * ta.st[s].width; ta.st[s].q = 1.0 - 0.015; ta.st[s].sat = pxsom_bmu::batch_gain_saturation(ta.st[s].q);
        }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        int rc = launch_batch_tail<float>(dx, c, c, ta, 0);
        hipEventRecord(e1, 0);
        if (rc) { printf("rc %d\n", rc); return 1; }
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long t[kMaxMembers][16];
        hipMemcpyFromSymbol(t, HIP_SYMBOL(g_tail_ticks), sizeof(t));
        TailCtl hc; hipMemcpy(&hc, scratch, sizeof(hc), hipMemcpyDeviceToHost);
        const int P = (int)hc.p_pub;
        printf("rep %d: %d members (chosen xcc %u), 16 steps in %.1f us (memset + launch + kernel)\n", rep, P, hc.chosen - 1, ms * 1e3);
        if (rep == 0) continue;
        const char *names[] = {"A reduce slots", "A update+publish", "-", "B wait(skew+flag)", "C gather", "C norms/scale",
                               "D search", "E wait waves", "E settle", "F table->slot", "A-flag wait"};
        for (int i = 0; i < 11; i++) {
            double mn = 1e9, mx = 0, sum = 0;
            for (int r = 0; r < P; r++) { const double d = (t[r][i + 1] - t[r][i]) * 0.01; mn = std::min(mn, d); mx = std::max(mx, d); sum += d; }
            printf("   %-20s min %6.2f mean %6.2f max %6.2f us\n", names[i], mn, sum / P, mx);
        }
        {
            double cv = 0, mf = 0, ac = 0;
            for (int r = 0; r < P; r++) { cv += (t[r][12] - t[r][6]) * 0.01; mf += (t[r][13] - t[r][12]) * 0.01; ac += (t[r][14] - t[r][13]) * 0.01; }
            printf("   D, wave 0, first round: convert %.2f | mfma + top-2 %.2f | merge + table (issue, all rounds) %.2f us\n", cv / P, mf / P, ac / P);
        }
        double tot = 0; for (int r = 0; r < P; r++) tot += (t[r][11] - t[r][0]) * 0.01;
        printf("   step %d total (mean over members) %.2f us\n", stamp_step, tot / P);
    }
    return 0;
}
