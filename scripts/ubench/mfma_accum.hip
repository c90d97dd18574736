// How does v_mfma_f32_16x16x32_f16 round?  One instruction per case on operands read from a file, results to a file; the cases
// and the fit of the rounding model are scripts/debug/mfma_accum_cases.py's (exact integer arithmetic on the host).
// The filter's tolerance charges every one of the 3C + 2 additions of a score with a binary32 rounding (pxsom_prep.h); what the
// matrix unit really does -- how many roundings per instruction, in which order -- decides how tight that term can be.
//   build: hipcc --offload-arch=gfx950 -O2 mfma_accum.hip -o mfma_accum;   run: mfma_accum in.bin out.bin
// in.bin : int32 n, then n x { half A[16][32] (row i, slot k), half B[32][16] (slot k, column j), float C[16][16] }
// out.bin: n x float D[16][16],  D = A B + C by ONE v_mfma_f32_16x16x32_f16; then n x float D2[16][16] by two chained
//          v_mfma_f32_16x16x16_f16 (slots 0..15, then 16..31) for comparison
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Case {
    _Float16 a[16][32];
    _Float16 b[32][16];
    float c[16][16];
};

__global__ __launch_bounds__(64) void run_cases(const Case *cases, float *d, float *d2, int n)
{
    const int lane = threadIdx.x, m = lane & 15, q = lane >> 4;
    const Case &cs = cases[blockIdx.x];
    half8 a, b;
    for (int i = 0; i < 8; i++) {   // lane (q, m): A[m][8 q + i], B[8 q + i][m]
        a[i] = cs.a[m][8 * q + i];
        b[i] = cs.b[8 * q + i][m];
    }
    f32x4 c;
    for (int r = 0; r < 4; r++) c[r] = cs.c[4 * q + r][m];   // lane (q, m): D[4 q + r][m]
    const f32x4 out = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) d[(size_t)blockIdx.x * 256 + (4 * q + r) * 16 + m] = out[r];
    // 16x16x16: lane (q, m): A[m][4 q + i], B[4 q + i][m], i = 0..3
    f32x4 acc = c;
    for (int h = 0; h < 2; h++) {
        half4 a4, b4;
        for (int i = 0; i < 4; i++) {
            a4[i] = cs.a[m][16 * h + 4 * q + i];
            b4[i] = cs.b[16 * h + 4 * q + i][m];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) d2[(size_t)blockIdx.x * 256 + (4 * q + r) * 16 + m] = acc[r];
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    int n = 0;
    if (fread(&n, 4, 1, f) != 1 || n < 1) return 4;
    std::vector<Case> host(n);
    if (fread(host.data(), sizeof(Case), n, f) != (size_t)n) return 5;
    fclose(f);
    Case *dev;
    float *d, *d2;
    if (hipMalloc(&dev, sizeof(Case) * n) != hipSuccess || hipMalloc(&d, 1024 * (size_t)n) != hipSuccess ||
        hipMalloc(&d2, 1024 * (size_t)n) != hipSuccess)
        return 6;
    (void)hipMemcpy(dev, host.data(), sizeof(Case) * n, hipMemcpyHostToDevice);
    run_cases<<<n, 64>>>(dev, d, d2, n);
    if (hipDeviceSynchronize() != hipSuccess) return 7;
    std::vector<float> out(512 * (size_t)n);
    (void)hipMemcpy(out.data(), d, 1024 * (size_t)n, hipMemcpyDeviceToHost);
    (void)hipMemcpy(out.data() + 256 * (size_t)n, d2, 1024 * (size_t)n, hipMemcpyDeviceToHost);
    f = fopen(argv[2], "wb");
    if (!f) return 8;
    fwrite(out.data(), 4, out.size(), f);
    fclose(f);
    printf("%d cases\n", n);
    return 0;
}
