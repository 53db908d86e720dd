// tools/experiments/tower_ubench.hip — stand-alone timing of the fused tower kernel (k_tower_c128).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "cz_trunk_experiments.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
int main(int argc, char **argv) {
    using namespace czconv;
    const int B = argc > 1 ? atoi(argv[1]) : 8192;
    const int nblocks = argc > 2 ? atoi(argv[2]) : 7;
    const int iters = argc > 3 ? atoi(argv[3]) : 10;
    const int variant = argc > 4 ? atoi(argv[4]) : 8;   // 4 = 2 positions / 4 waves, 8 = 4 positions / 8 waves, 1 = 4 positions, one per wave
    const int zero = argc > 5 ? atoi(argv[5]) : 0;      // 1 = all-zero activations and weights (power experiment)
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
    if (zero) { CK(hipMemset(in, 0, n * 2)); CK(hipMemset(w, 0, nw * 2)); }
    CK(hipMemset(bias, 0, 2 * nblocks * 128 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tower_c128), hipFuncAttributeMaxDynamicSharedMemorySize, TW_LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tower8_c128<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_towerp_c128), hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS_BYTES));
    const int grid = variant == 4 ? (B + TW_P - 1) / TW_P : (B + T8_P - 1) / T8_P;
#define LAUNCH() do { if (variant == 4) hipLaunchKernelGGL(k_tower_c128, dim3(grid), dim3(TW_THREADS), TW_LDS_BYTES, 0, in, w, bias, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, 2 * nblocks); \
                      else if (variant == 1) hipLaunchKernelGGL(k_towerp_c128, dim3(grid), dim3(TP_THREADS), T8_LDS_BYTES, 0, in, w, bias, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, 2 * nblocks); \
                      else hipLaunchKernelGGL((k_tower8_c128<false, 4>), dim3(grid), dim3(T8_THREADS), T8_LDS_BYTES, 0, in, w, bias, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, 2 * nblocks, (const int *)nullptr, (unsigned long long *)nullptr); } while (0)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) LAUNCH();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) LAUNCH();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * nblocks * 2.0 * B * 90 * 1152 * 128 / (us * 1e-6) / 1e12;
    printf("tower variant=%dw B=%d blocks=%d : %9.1f us/launch (%7.1f us/layer) %7.1f TF/s\n", variant, B, nblocks, us, us / (2 * nblocks), tf);
    return 0;
}
