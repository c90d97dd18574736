// Diagnosis: where does a step of the exact-online SOM kernel spend its cycles?  Includes the product
// kernel source with PXSOM_STEP_TIMING (s_memtime deltas per segment, wave 0) and runs config-2-like
// input (C = 22, 10x10) for 200k steps.
#define PXSOM_STEP_TIMING 1
#include "../../ark_analysis_amd/csrc/pxsom_train.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main()
{
    const int64_t n = 200000;
    const int c = 22, xdim = 10, ydim = 10, K = 100;
    std::vector<float> x((size_t)n * c);
    srand(1);
    for (auto &v : x) v = (float)rand() / RAND_MAX;
    std::vector<double> w((size_t)K * c);
    for (int k = 0; k < K; k++)
        for (int j = 0; j < c; j++) w[(size_t)k * c + j] = x[(size_t)(k * 997) * c + j];
    std::vector<int64_t> order(n);
    for (auto &o : order) o = rand() % n;
    float *dx; double *dw; int64_t *dor;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dw, w.size() * 8); hipMalloc(&dor, n * 8);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dor, order.data(), n * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++) {
        hipMemcpy(dw, w.data(), w.size() * 8, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        int rc = pxsom_train_online(dx, n, c, c, PXSOM_F32, dw, xdim, ydim, 1, 0.05, 0.01, 6.0, 0.0, dor, 0);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long t[8];
        hipMemcpyFromSymbol(t, HIP_SYMBOL(g_step_ticks), sizeof(t));
        printf("rc %d  %.3f ms  %.1f ns/step\n", rc, ms, ms * 1e6 / n);
        const char *nm[8] = {"loop top / epoch logic", "sub+mul+chain", "key min + candidate", "readlane+write+barrier",
                             "exchange read + select", "update", "commit + barrier (per chunk)", "gather issue (per chunk)"};
        for (int i = 0; i < 8; i++) printf("  seg %d %-30s %8.1f ticks/step\n", i, nm[i], (double)t[i] / n);
    }
    return 0;
}
