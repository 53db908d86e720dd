// tools/conv_ubench.hip — stand-alone timing of the fused conv3x3 kernel and its ablations.
// Build variants with -DCZ_CONV_P=.. -DCZ_ABL=.. and run on the GPU box; prints us/launch and TF/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../cchess_zero_amd/csrc/cz_conv_kernel.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
    using namespace czconv;
    const int B = argc > 1 ? atoi(argv[1]) : 8192;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    const size_t n = (size_t)B * 90 * 128;
    uint16_t *in, *out, *res, *w; float *bias;
    CK(hipMalloc(&in, n * 2)); CK(hipMalloc(&out, n * 2)); CK(hipMalloc(&res, n * 2));
    CK(hipMalloc(&w, 9 * 128 * 128 * 2)); CK(hipMalloc(&bias, 128 * 4));
    std::vector<uint16_t> h(n);
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (uint16_t)(0x3C00 + ((s >> 16) & 0x3FF) + ((s >> 31) << 15)); }  // random-ish bf16 in +-[0.0078,0.031]
    CK(hipMemcpy(in, h.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(res, h.data(), n * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, h.data(), 9 * 128 * 128 * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 512));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv3x3_c128), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS_BYTES));
    const int grid = (B + CV_P - 1) / CV_P;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_conv3x3_c128, dim3(grid), dim3(CV_THREADS), CV_LDS_BYTES, 0, in, w, bias, res, out, B, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_conv3x3_c128, dim3(grid), dim3(CV_THREADS), CV_LDS_BYTES, 0, (i & 1) ? out : in, w, bias, res, (i & 1) ? in : out, B, 1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * B * 90 * 1152 * 128 / (us * 1e-6) / 1e12;
    printf("P=%d threads=%d lds=%d abl=%d B=%d : %8.1f us/launch  %7.1f TF/s\n", CV_P, CV_THREADS, CV_LDS_BYTES, CZ_ABL, B, us, tf);
    return 0;
}
