// tools/experiments/variants_check.hip — k_towersk_c128 (half-workgroups four slabs apart) against k_tower8_c128: bit-equality of the trunk
// output and of the head-conv output on the same data (both entry paths: 128-channel input, and input planes through the
// first layer), then alternating timings of the two kernels on that data (Glorot-sized weights, half-zero activations: the
// activations stay finite, unlike tools/experiments/tower_ubench.hip's).  args: B blocks fp16(0|1) iters variant(1 = k_towersk_c128, 2 = k_towerd_c128: weight fragments from global memory, no ring)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "cz_trunk_experiments.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
static unsigned rs = 12345;
static float urand() { rs = rs * 1664525u + 1013904223u; return (float)(rs >> 8) * (1.0f / 16777216.0f); }
static uint16_t to16(float f, int f16) {
    if (f16) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
    uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16);
}
int main(int argc, char **argv) {
    using namespace czconv;
    const int B = argc > 1 ? atoi(argv[1]) : 8192, nblocks = argc > 2 ? atoi(argv[2]) : 7, f16 = argc > 3 ? atoi(argv[3]) : 1, iters = argc > 4 ? atoi(argv[4]) : 20, var = argc > 5 ? atoi(argv[5]) : 1;
    const int nl = 2 * nblocks;
    const size_t n = (size_t)B * 90 * 128, nw = (size_t)nl * 9 * 128 * 128, np = (size_t)B * 90 * 16, nw0 = 9 * 2 * 128 * 8;
    uint16_t *in, *pl, *w, *w0, *out[2]; float *bias, *b0, *hw, *hb, *ho[2];
    CK(hipMalloc(&in, n * 2)); CK(hipMalloc(&pl, np * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&w0, nw0 * 2));
    CK(hipMalloc(&bias, nl * 128 * 4)); CK(hipMalloc(&b0, 128 * 4)); CK(hipMalloc(&hw, 3 * 128 * 4)); CK(hipMalloc(&hb, 3 * 4));
    for (int k = 0; k < 2; ++k) { CK(hipMalloc(&out[k], n * 2)); CK(hipMalloc(&ho[k], (size_t)B * 90 * 3 * 4)); }
    std::vector<uint16_t> h(n > nw ? n : nw);
    for (size_t i = 0; i < n; ++i) h[i] = urand() < 0.5f ? 0 : to16(urand(), f16);
    CK(hipMemcpy(in, h.data(), n * 2, hipMemcpyHostToDevice));
    for (size_t i = 0; i < np; ++i) h[i] = (i & 15) < 14 && urand() < 0.07f ? to16(1.0f, f16) : 0;
    CK(hipMemcpy(pl, h.data(), np * 2, hipMemcpyHostToDevice));
    for (size_t i = 0; i < nw; ++i) h[i] = to16((urand() - 0.5f) * 0.102f, f16);     // Glorot: +-sqrt(6 / 2304)
    CK(hipMemcpy(w, h.data(), nw * 2, hipMemcpyHostToDevice));
    for (size_t i = 0; i < nw0; ++i) h[i] = to16((urand() - 0.5f) * 0.14f, f16);
    CK(hipMemcpy(w0, h.data(), nw0 * 2, hipMemcpyHostToDevice));
    std::vector<float> hf(nl * 128);
    for (auto &x : hf) x = (urand() - 0.5f) * 0.1f;
    CK(hipMemcpy(bias, hf.data(), nl * 128 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b0, hf.data(), 128 * 4, hipMemcpyHostToDevice));
    for (int i = 0; i < 384; ++i) hf[i] = (urand() - 0.5f) * 0.4f;
    CK(hipMemcpy(hw, hf.data(), 384 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(hb, hf.data(), 12, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tower8_c128<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tower8_c128<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_towersk_c128<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_towersk_c128<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_towerd_c128<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TD_LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_towerd_c128<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TD_LDS_BYTES));
    const char *vname = var == 2 ? "k_towerd_c128 " : "k_towersk_c128";
    const int grid = (B + 3) / 4;
    auto launch = [&](int which, bool planes_path) {
        const uint16_t *i_ = planes_path ? nullptr : in, *p_ = planes_path ? pl : nullptr;
        if (which == 0) {
            if (f16) hipLaunchKernelGGL((k_tower8_c128<true, 4>), dim3(grid), dim3(T8_THREADS), T8_LDS_BYTES, 0, i_, w, bias, out[0], hw, hb, ho[0], p_, w0, b0, B, nl, nullptr, nullptr);
            else hipLaunchKernelGGL((k_tower8_c128<false, 4>), dim3(grid), dim3(T8_THREADS), T8_LDS_BYTES, 0, i_, w, bias, out[0], hw, hb, ho[0], p_, w0, b0, B, nl, nullptr, nullptr);
        } else if (var == 2) {
            if (f16) hipLaunchKernelGGL((k_towerd_c128<true>), dim3(grid), dim3(TD_THREADS), TD_LDS_BYTES, 0, i_, w, bias, out[1], hw, hb, ho[1], p_, w0, b0, B, nl, nullptr);
            else hipLaunchKernelGGL((k_towerd_c128<false>), dim3(grid), dim3(TD_THREADS), TD_LDS_BYTES, 0, i_, w, bias, out[1], hw, hb, ho[1], p_, w0, b0, B, nl, nullptr);
        } else {
            if (f16) hipLaunchKernelGGL((k_towersk_c128<true>), dim3(grid), dim3(SK_THREADS), SK_LDS_BYTES, 0, i_, w, bias, out[1], hw, hb, ho[1], p_, w0, b0, B, nl, nullptr);
            else hipLaunchKernelGGL((k_towersk_c128<false>), dim3(grid), dim3(SK_THREADS), SK_LDS_BYTES, 0, i_, w, bias, out[1], hw, hb, ho[1], p_, w0, b0, B, nl, nullptr);
        }
    };
    int bad = 0;
    std::vector<uint16_t> o0(n), o1(n); std::vector<float> g0((size_t)B * 270), g1((size_t)B * 270);
    for (int path = 0; path < 2; ++path) {
        for (int k = 0; k < 2; ++k) { CK(hipMemset(out[k], 0xEE, n * 2)); CK(hipMemset(ho[k], 0xEE, (size_t)B * 270 * 4)); }
        launch(0, path); launch(1, path);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o0.data(), out[0], n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), out[1], n * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(g0.data(), ho[0], (size_t)B * 270 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(g1.data(), ho[1], (size_t)B * 270 * 4, hipMemcpyDeviceToHost));
        size_t d = 0, dh = 0, nz = 0, first = (size_t)-1;
        for (size_t i = 0; i < n; ++i) { if (o0[i] != o1[i]) { if (!d) first = i; ++d; } nz += o0[i] != 0; }
        for (size_t i = 0; i < g0.size(); ++i) dh += memcmp(&g0[i], &g1[i], 4) != 0;
        printf("%s path (%s, B=%d, %d layers): trunk outputs differing %zu of %zu (non-zero in the reference kernel: %zu), head outputs differing %zu of %zu",
               path ? "planes" : "128-channel input", f16 ? "fp16" : "bf16", B, nl, d, n, nz, dh, g0.size());
        if (d) printf("; first at position %zu cell %zu channel %zu (%04x vs %04x)", first / 11520, first / 128 % 90, first % 128, o0[first], o1[first]);
        printf("\n");
        bad += d != 0 || dh != 0 || nz == 0;
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep)
        for (int k = 0; k < 2; ++k) {
            for (int i = 0; i < 5; ++i) launch(k, true);
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) launch(k, true);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s: %8.1f us per launch\n", k ? vname : "k_tower8_c128 ", ms * 1e3 / iters);
        }
    printf(bad ? "MISMATCH\n" : "bit-identical\n");
    return bad != 0;
}
