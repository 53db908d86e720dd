// tools/experiments/tower_skip_ubench.hip — the zero-work elasticity experiment: the product trunk
// kernel k_tower8_c128<fp16> with the MFMAs of every wave's third cell tile REMOVED for the first CZ_T8_SKIPTEST taps of every
// layer (wrong results; timing only).  19 % of the MFMAs the kernel issues multiply padding rows or the zero row; removing them
// for real needs a cell permutation and cannot be balanced over the eight waves.  This measures the upper bound instead: a
// perfectly balanced removal of 3/27 (one tile, 3 taps: ~ all the zero taps), 9/27 = 1/3 (one tile, every tap) of the MFMAs — how
// much of the saved issue time survives the power governor?  Built four times by build.sh: CZ_T8_SKIPTEST = 0 (none: the base), 3, 9.
// args: B blocks iters.  Data: half-zero activations, Glorot-sized weights (finite through all layers).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "cz_experiments_slab_asm.inc"
#if !defined(CZ_T8_SKIPTEST) && !defined(CZ_T8_PRODUCT)   // -DCZ_T8_PRODUCT: the kernel exactly as the library builds it
#define CZ_T8_SKIPTEST 0
#endif
#include "../../cchess_zero_amd/csrc/cz_conv_kernel.h"
#ifdef CZ_T8_SKIPTEST
#define SKIPN CZ_T8_SKIPTEST
#else
#define SKIPN 0
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
static unsigned rs = 12345;
static float urand() { rs = rs * 1664525u + 1013904223u; return (float)(rs >> 8) * (1.0f / 16777216.0f); }
static uint16_t to16(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
int main(int argc, char **argv) {
    using namespace czconv;
    const int B = argc > 1 ? atoi(argv[1]) : 8192, nblocks = argc > 2 ? atoi(argv[2]) : 7, iters = argc > 3 ? atoi(argv[3]) : 20;
    const int nl = 2 * nblocks;
    const size_t n = (size_t)B * 90 * 128, nw = (size_t)nl * 9 * 128 * 128;
    uint16_t *in, *w, *out; float *bias; unsigned long long *clk;
    CK(hipMalloc(&in, n * 2)); CK(hipMalloc(&out, n * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&bias, nl * 128 * 4));
    CK(hipMalloc(&clk, (size_t)(B / 4 + 1) * 32));
    std::vector<uint16_t> h(n > nw ? n : nw);
    for (size_t i = 0; i < n; ++i) h[i] = urand() < 0.5f ? 0 : to16(urand());
    CK(hipMemcpy(in, h.data(), n * 2, hipMemcpyHostToDevice));
    for (size_t i = 0; i < nw; ++i) h[i] = to16((urand() - 0.5f) * 0.102f);
    CK(hipMemcpy(w, h.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, nl * 128 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tower8_c128<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS_BYTES));
    const int grid = (B + 3) / 4;
    auto launch = [&]() {
        hipLaunchKernelGGL((k_tower8_c128<true, 4>), dim3(grid), dim3(T8_THREADS), T8_LDS_BYTES, 0, in, w, bias, out, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr, B, nl, nullptr, clk);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> c((size_t)grid * 4);
        CK(hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0, ref = 0;
        for (int g = 0; g < grid; ++g) { cyc += (double)(c[g * 4 + 1] - c[g * 4]); ref += (double)(c[g * 4 + 3] - c[g * 4 + 2]) * 10e-9; }
        printf("skip %d of 9 taps x 1 of 3 tiles (%.1f %% of the MFMAs removed): %8.1f us per launch, %.0f cycles per workgroup, effective clock %.3f GHz\n",
               SKIPN, 100.0 * SKIPN / 27.0, ms * 1e3 / iters, cyc / grid, cyc / ref / 1e9);
    }
    return 0;
}
