// Diagnosis: phase times (s_memtime) inside batch_update_prep_kernel for config 5's codebook (20 x 20, C = 40)
// and config 4's (10 x 10, C = 100).
#define PXSOM_PHASE_TIMING 1
#define PXSOM_PHASE_BLOCK0_ONLY 1
#include "../../ark_analysis_amd/csrc/pxsom_batch_step.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
int main()
{
    using namespace pxsom_bmu;
    for (int cfg = 0; cfg < 2; cfg++) {
        const int xd = cfg == 0 ? 20 : 10, c = cfg == 0 ? 40 : 100, K = xd * xd;
        std::vector<double> w((size_t)K * c), st((size_t)K * (c + 1));
        srand(3);
        for (auto &v : w) v = (double)rand() / RAND_MAX;
        for (size_t i = 0; i < (size_t)K * c; i++) st[i] = 60.0 * rand() / RAND_MAX;
        for (int i = 0; i < K; i++) st[(size_t)K * c + i] = 30 + rand() % 80;
        double *dw, *dw2, *ds, *dz; char *ws;
        const Layout L = make_layout(30000, c, K);
        hipMalloc(&dw, w.size() * 8); hipMalloc(&dw2, w.size() * 8); hipMalloc(&ds, st.size() * 8); hipMalloc(&dz, st.size() * 8);
        hipMalloc(&ws, L.total);
        hipMemcpy(dw, w.data(), w.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(ds, st.data(), st.size() * 8, hipMemcpyHostToDevice);
        for (double thr : {11.0, 5.5, 0.5}) {
            StepArgs sa;
            sa.w_in = dw; sa.w_out = dw2; sa.stats_prev = ds; sa.stats_zero = dz; sa.zero_count = (int)st.size(); sa.has_update = 1;
            sa.thr = thr; sa.q = 1.0 - 0.04; sa.sat = pxsom_bmu::batch_gain_saturation(sa.q); sa.tol_rel = 1e-5f; sa.tol_abs = 1e-6f;
            int rc = 0;
            for (int rep = 0; rep < 3; rep++) {
                bool ok = launch_update_prepare(sa, xd, xd, c, ws, L, 0, &rc);
                hipDeviceSynchronize();
                if (!ok || rc) { printf("not covered / rc %d %s\n", rc, pxsom_last_error()); return 1; }
            }
            long long t[32];
            hipMemcpyFromSymbol(t, HIP_SYMBOL(g_phase_ticks), sizeof(t));
            auto us = [&](int a, int b) { return (t[b] - t[a]) / 2400.0; };
            printf("%dx%d c=%d thr %.1f: loads+zero %.2f | pass1 %.2f | pass2 %.2f | gain %.2f | new w %.2f | to LDS+norms %.2f | "
                   "reduce+scale %.2f | dups %.2f | frags+bias %.2f | wt %.2f | total %.2f us\n", xd, xd, c, thr, us(0, 1), us(1, 2),
                   us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(6, 7), us(7, 8), us(8, 9), us(9, 10), us(0, 10));
        }
    }
    return 0;
}
