// tools/tower_ubench.hip — stand-alone timing of the fused tower kernel (k_tower_c128).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../cchess_zero_amd/csrc/cz_conv_kernel.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char **argv) {
    using namespace czconv;
    const int B = argc > 1 ? atoi(argv[1]) : 8192;
    const int nblocks = argc > 2 ? atoi(argv[2]) : 7;
    const int iters = argc > 3 ? atoi(argv[3]) : 10;
    const size_t n = (size_t)B * 90 * 128, nw = (size_t)2 * nblocks * 9 * 128 * 128;
    uint16_t *in, *out, *w; float *bias;
    CK(hipMalloc(&in, n * 2)); CK(hipMalloc(&out, n * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&bias, 2 * nblocks * 128 * 4));
    std::vector<uint16_t> h(n > nw ? n : nw);
    unsigned s = 12345;
    // activations: ~half zeros (post-ReLU look), weights: small signed values so the tower does not blow up
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s & 0x10000) ? 0 : (uint16_t)(0x3C00 + ((s >> 17) & 0x3FF)); }
    CK(hipMemcpy(in, h.data(), n * 2, hipMemcpyHostToDevice));
    for (size_t i = 0; i < nw; ++i) { s = s * 1664525u + 1013904223u; h[i] = (uint16_t)(0x3A00 + ((s >> 16) & 0x1FF) + ((s >> 31) << 15)); }
    CK(hipMemcpy(w, h.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 2 * nblocks * 128 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tower_c128), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS_BYTES));
    const int grid = (B + TW_P - 1) / TW_P;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_tower_c128, dim3(grid), dim3(TW_THREADS), TW_LDS_BYTES, 0, in, w, bias, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, 2 * nblocks);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_tower_c128, dim3(grid), dim3(TW_THREADS), TW_LDS_BYTES, 0, in, w, bias, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, 2 * nblocks);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * nblocks * 2.0 * B * 90 * 1152 * 128 / (us * 1e-6) / 1e12;
#if CZ_TTRACE
    {
        std::vector<unsigned long long> t(8192);
        CK(hipMemcpy(t.data(), out, 8192 * 8, hipMemcpyDeviceToHost));
        const int ns = 2 * nblocks * 18;
        double h1 = 0, w = 0, b = 0, h2 = 0; int cnt = 0;
        for (int g = 4; g + 1 < ns; ++g) {
            h1 += (double)(t[8 + g * 4 + 1] - t[8 + g * 4 + 0]); w += (double)(t[8 + g * 4 + 2] - t[8 + g * 4 + 1]);
            b += (double)(t[8 + g * 4 + 3] - t[8 + g * 4 + 2]); h2 += (double)(t[8 + (g + 1) * 4 + 0] - t[8 + g * 4 + 3]); ++cnt;
        }
        double ep = 0; for (int l = 0; l < 2 * nblocks; ++l) ep += (double)(t[4096 + 2 * l + 1] - t[4096 + 2 * l]);
        printf("trace WG1500 wave0: per slab: first half %.0f, vmcnt wait %.0f, barrier %.0f, second half(+next) %.0f ticks; epilogue %.0f ticks/layer; whole WG %.0f ticks\n",
               h1 / cnt, w / cnt, b / cnt, h2 / cnt, ep / (2 * nblocks), (double)(t[4096 + 2 * (2 * nblocks - 1) + 1] - t[8]));
        for (int g = 30; g < 40; ++g) printf("  slab %d: %llu %llu %llu | next %llu\n", g, t[8+g*4+1]-t[8+g*4+0], t[8+g*4+2]-t[8+g*4+1], t[8+g*4+3]-t[8+g*4+2], t[8+(g+1)*4]-t[8+g*4+3]);
    }
#endif
    printf("tower P=%d threads=%d lds=%d B=%d blocks=%d : %9.1f us/launch (%7.1f us/layer) %7.1f TF/s\n", TW_P, TW_THREADS, TW_LDS_BYTES, B, nblocks, us, us / (2 * nblocks), tf);
    return 0;
}
