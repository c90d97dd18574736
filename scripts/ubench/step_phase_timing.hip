// Diagnosis: phase times (s_memtime, workgroup 0) inside the fused mini-batch step kernel
// (pxsom_batch_step.hip) for config 2's mini-batch (16384 rows of a strided 1 M x 22 view).
#define PXSOM_PHASE_TIMING 1
#define PXSOM_PHASE_BLOCK0_ONLY 1
#include "../../ark_analysis_amd/csrc/pxsom_batch_step.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

int main(int argc, char **argv)
{
    // argv: tiles per wave (0 = by step size), steps per pass M (64: round 2's 16 K-row steps; 144: the 7 280-row steps of round 3's tail)
    const int M = argc > 2 ? atoi(argv[2]) : 64, c = 22, K = 100, tpw = argc > 1 ? atoi(argv[1]) : 2;
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
    float *dx; double *dwbuf, *dring;
    const size_t nw = (size_t)K * c, ns = (size_t)K * (c + 1);
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dwbuf, 2 * nw * 8); hipMalloc(&dring, 3 * ns * 8);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    using namespace pxsom_bmu;
    for (int rep = 0; rep < 2; rep++) {
        hipMemcpy(dwbuf, w.data(), nw * 8, hipMemcpyHostToDevice);
        hipMemset(dring, 0, 3 * ns * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        for (int g = 0; g < M; g++) {
            StepArgs sa;
            sa.w_in = g > 0 ? dwbuf + ((g + 1) % 2) * nw : dwbuf; sa.w_out = dwbuf + (g % 2) * nw;
            sa.stats_prev = dring + ((g + 2) % 3) * ns; sa.stats_zero = dring + ((g + 1) % 3) * ns;
            sa.zero_count = (int)ns; sa.has_update = g > 0;
            double thr = 6.0 - 6.0 * (g > 0 ? g - 1 : 0) / M; if (thr < 1) thr = 0.5;
            sa.thr = thr; sa.q = 1.0 - (0.05 - 0.04 * (g > 0 ? g - 1 : 0) / M); sa.sat = pxsom_bmu::batch_gain_saturation(sa.q);
            sa.tol_rel = (float)(2.5 * (ldexp(1.0, -16) + (3.0 * c + 2.0) * ldexp(1.0, -24) + ldexp(1.0, -19) + ldexp(1.0, -23)));
            sa.tol_abs = (float)(2.5 * ldexp(1.0, -24) * sqrt((double)c));
            const int64_t rows = (n - g + M - 1) / M;
            int rc = launch_batch_step<float>(dx + (size_t)g * c, rows, c, (int64_t)c * M, dring + (g % 3) * ns, sa, tpw, 0);
            if (rc) { printf("rc %d %s\n", rc, pxsom_last_error()); return 1; }
            if (rep == 1 && (g == 1 || g == M / 3 || g == (2 * M) / 3 || g == M - 4 || g == M - 1)) {
                hipDeviceSynchronize();
                long long t[32];
                hipMemcpyFromSymbol(t, HIP_SYMBOL(g_phase_ticks), sizeof(t));
                auto us = [&](int a, int b) { return (t[b] - t[a]) / 2400.0; };   // s_memtime ticks at the 2.4 GHz shader clock (approx.)
                printf("step %2d thr %.2f: rows+loads %.2f | pass1 %.2f pass2 %.2f | neww+norms %.2f | scale+frags %.2f | dups %.2f | bias %.2f | "
                       "filter %.2f | exact(%lld) %.2f | flush-issue %.2f flush-done %.2f | total %.2f us\n",
                       g, thr, us(8, 9), us(9, 10), us(10, 11), us(11, 12), us(12, 13), us(13, 14), us(14, 15), us(15, 16), t[21],
                       us(16, 17), us(17, 18), us(18, 19), us(8, 19));
                printf("         filter: barrier %.2f | convert %.2f | mfma + top-2 %.2f | merge + table %.2f\n", us(15, 22), us(22, 23),
                       us(23, 24), us(24, 16));
                {
                    long long bt[512];
                    hipMemcpyFromSymbol(bt, HIP_SYMBOL(g_block_ticks), sizeof(bt));
                    const int nwg = std::min(256, (int)((rows + (tpw <= 1 ? 127 : 255)) / (tpw <= 1 ? 128 : 256)));
                    long long s0 = bt[0], s1 = bt[0], e0 = bt[1], e1 = bt[1];
                    double dsum = 0;
                    for (int b = 0; b < nwg; b++) {
                        s0 = std::min(s0, bt[2 * b]); s1 = std::max(s1, bt[2 * b]);
                        e0 = std::min(e0, bt[2 * b + 1]); e1 = std::max(e1, bt[2 * b + 1]);
                        dsum += (double)(bt[2 * b + 1] - bt[2 * b]);
                    }
                    printf("         %d workgroups: starts spread %.2f us, ends spread %.2f us, mean in-workgroup time %.2f us, first start -> last end %.2f us\n",
                           nwg, (s1 - s0) / 100.0, (e1 - e0) / 100.0, dsum / nwg / 100.0, (e1 - s0) / 100.0);
                }
            }
        }
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rep %d: %d steps %.3f ms (%.2f us/step)\n", rep, M, ms, ms * 1e3 / M);
    }
    return 0;
}
