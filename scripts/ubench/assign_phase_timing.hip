// Diagnosis: phase times (s_memtime) inside the single-workgroup prep kernel and workgroup 0 of the exact
// kernel for one 16K-row mini-batch of the config-2 shape (C = 22, K = 100).
#define PXSOM_PHASE_TIMING 1
#include "../../ark_analysis_amd/csrc/pxsom_assign.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char **argv)
{
    const int64_t n = 16384;
    const int c = argc > 1 ? atoi(argv[1]) : 22, K = argc > 2 ? atoi(argv[2]) : 100;
    std::vector<float> x((size_t)n * c);
    srand(1);
    for (auto &v : x) v = (float)rand() / RAND_MAX;
    for (int j = 0; j < c; j++) x[(size_t)5 * c + j] = x[(size_t)6 * c + j] = 0.5f;   // guaranteed ties
    std::vector<double> w((size_t)K * c);
    for (int k = 0; k < K; k++)
        for (int j = 0; j < c; j++) w[(size_t)k * c + j] = x[(size_t)(k * 97) * c + j];
    for (int j = 0; j < c; j++) w[(size_t)9 * c + j] = w[(size_t)4 * c + j] + (j == 0 ? 1e-9 : 0.0);
    float *dx; double *dw, *dstats; int *dl; char *ws;
    const size_t wsb = pxsom_assign_workspace_bytes(n, c, K);
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dw, w.size() * 8); hipMalloc(&dl, n * 4); hipMalloc(&ws, wsb);
    hipMalloc(&dstats, (size_t)K * (c + 1) * 8);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw, w.data(), w.size() * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; rep++) {
        bool fused = false;
        int rc = argc > 1 ? pxsom_assign(dx, n, c, c, PXSOM_F32, dw, K, dl, nullptr, ws, wsb, nullptr)
                          : pxsom_bmu::assign_accumulate(dx, n, c, c, PXSOM_F32, dw, K, dl, dstats, ws, wsb, 0, &fused);
        hipDeviceSynchronize();
        long long t[32];
        hipMemcpyFromSymbol(t, HIP_SYMBOL(g_phase_ticks), sizeof(t));
        unsigned cnt;
        hipMemcpy(&cnt, ws, 4, hipMemcpyDeviceToHost);
        printf("rc %d fused %d listed rows %u\n", rc, (int)fused, cnt);
        const char *pn[7] = {"zero stats + stage W", "norms + max reduce", "norm-max reduce", "header",
                             "fragments", "duplicate hash", "bias"};
        for (int i = 0; i < 7; i++) printf("  prep  %-22s %8.2f us\n", pn[i], (t[i + 1] - t[i]) / 2400.0 * 1.0);
        printf("  prep  total %.2f us\n", (t[7] - t[0]) / 2400.0);
        printf("  exact hdr read %.2f us, W staging %.2f us, distances %.2f us, reduce+store %.2f us, total %.2f us\n",
               (t[9] - t[8]) / 2400.0, (t[10] - t[9]) / 2400.0, (t[11] - t[10]) / 2400.0, (t[12] - t[11]) / 2400.0,
               (t[12] - t[8]) / 2400.0);
    }
    return 0;
}
