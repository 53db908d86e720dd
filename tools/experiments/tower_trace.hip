// tools/experiments/tower_trace.hip — where the cycles of a layer of k_tower8_c128 go.  Built twice by tools/tower_trace.sh
// (-DCZ_T8_TRACE=1: four stamps per layer at points where the wave waits for its scalar/LDS counters anyway; =2: plus one per
// tap, which perturbs the software pipeline) and run at the benchmark's batch; prints per-layer means over all workgroups and
// waves, in shader-clock ticks and as a share of the layer.  args: B blocks fp16(0|1) warm_launches kernel(0 = k_tower8_c128, 1 = k_towersk_c128, 2 = k_towerd_c128)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "cz_trunk_experiments.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned long long u64;
int main(int argc, char **argv) {
    using namespace czconv;
    const int B = argc > 1 ? atoi(argv[1]) : 8192, nblocks = argc > 2 ? atoi(argv[2]) : 7, f16 = argc > 3 ? atoi(argv[3]) : 1, warm = argc > 4 ? atoi(argv[4]) : 300, sk = argc > 5 ? atoi(argv[5]) : 0;
    const int nl = 2 * nblocks, grid = (B + T8_P - 1) / T8_P, W = 8;
    const size_t n = (size_t)B * 90 * 128, nw = (size_t)nl * 9 * 128 * 128;
    uint16_t *in, *out, *w; float *bias; u64 *tr;
    CK(hipMalloc(&in, n * 2)); CK(hipMalloc(&out, n * 2)); CK(hipMalloc(&w, nw * 2)); CK(hipMalloc(&bias, nl * 128 * 4));
    const size_t ntr = (size_t)grid * W * nl * 16;
    CK(hipMalloc(&tr, ntr * 8)); CK(hipMemset(tr, 0, ntr * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(cz_t8_trace_buf), &tr, sizeof tr));
    std::vector<uint16_t> h(std::max(n, nw));
    unsigned s = 12345;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s & 0x10000) ? 0 : (uint16_t)(0x3C00 + ((s >> 17) & 0x3FF)); }
    CK(hipMemcpy(in, h.data(), n * 2, hipMemcpyHostToDevice));
    // weights around +-0.03 (fp16: exponent 0x25..0x29; bf16 patterns of similar size are used for the bf16 run) so that the activations stay finite
    for (size_t i = 0; i < nw; ++i) { s = s * 1664525u + 1013904223u; h[i] = f16 ? (uint16_t)(0x2400 + ((s >> 16) & 0x7FF) + ((s >> 31) << 15)) : (uint16_t)(0x3C80 + ((s >> 16) & 0x7F) + ((s >> 31) << 15)); }
    CK(hipMemcpy(w, h.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, nl * 128 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tower8_c128<false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_tower8_c128<true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, T8_LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_towersk_c128<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_towersk_c128<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SK_LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_towerd_c128<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TD_LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_towerd_c128<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TD_LDS_BYTES));
    auto launch = [&] {
        if (sk == 2 && f16) hipLaunchKernelGGL((k_towerd_c128<true>), dim3(grid), dim3(TD_THREADS), TD_LDS_BYTES, 0, in, w, bias, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, nl, nullptr);
        else if (sk == 2) hipLaunchKernelGGL((k_towerd_c128<false>), dim3(grid), dim3(TD_THREADS), TD_LDS_BYTES, 0, in, w, bias, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, nl, nullptr);
        else if (sk && f16) hipLaunchKernelGGL((k_towersk_c128<true>), dim3(grid), dim3(SK_THREADS), SK_LDS_BYTES, 0, in, w, bias, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, nl, nullptr);
        else if (sk) hipLaunchKernelGGL((k_towersk_c128<false>), dim3(grid), dim3(SK_THREADS), SK_LDS_BYTES, 0, in, w, bias, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, nl, nullptr);
        else if (f16) hipLaunchKernelGGL((k_tower8_c128<true, 4>), dim3(grid), dim3(T8_THREADS), T8_LDS_BYTES, 0, in, w, bias, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, nl, nullptr, nullptr);
        else hipLaunchKernelGGL((k_tower8_c128<false, 4>), dim3(grid), dim3(T8_THREADS), T8_LDS_BYTES, 0, in, w, bias, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, nl, nullptr, nullptr);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < warm; ++i) launch();   // clocks and power settle on this kernel (0.6 s by default)
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<u64> t(ntr);
    CK(hipMemcpy(t.data(), tr, ntr * 8, hipMemcpyDeviceToHost));
    auto at = [&](int g, int wv, int l, int k) { return t[(((size_t)g * W + wv) * nl + l) * 16 + k]; };
    // the shader clock counters of the 8 XCDs do not share an origin: only differences inside one workgroup mean anything
    double wgdur = 0;
    for (int g = 0; g < grid; ++g) wgdur += (double)(at(g, 0, nl - 1, 3) - at(g, 0, 0, 0));
    wgdur /= grid;
    const double rounds = (double)grid / 256.0;
    printf("%s, trace level %d, %s, B=%d, %d layers: kernel %.1f us; a workgroup spends %.0f ticks in its %d layers; %.1f rounds of workgroups "
           "=> >= %.2f ticks/ns if a round were nothing but its layers\n", sk == 2 ? "k_towerd_c128" : sk ? "k_towersk_c128" : "k_tower8_c128", (int)CZ_T8_TRACE, f16 ? "fp16" : "bf16", B, nl, ms * 1e3, wgdur, nl,
           rounds, wgdur * rounds / (ms * 1e6));
    // per layer (layers 1.. : layer 0 has no predecessor stamp), means over workgroups and waves
    double top = 0, loop = 0, wait = 0, epi = 0, gap = 0, layer = 0, skew = 0; size_t cnt = 0, cntg = 0;
    double waitw[8] = {0}, loopw[8] = {0}, tapd[9] = {0};
    for (int g = 0; g < grid; ++g)
        for (int l = 1; l < nl; ++l) {
            u64 mn = ~0ull, mx = 0;
            for (int wv = 0; wv < W; ++wv) {
                const u64 t0 = at(g, wv, l, 0), t1 = at(g, wv, l, 1), t2 = at(g, wv, l, 2), t13 = at(g, wv, l, 13), t3 = at(g, wv, l, 3), p3 = at(g, wv, l - 1, 3);
                top += (double)(t1 - t0); loop += (double)(t2 - t1); wait += (double)(t13 - t2); epi += (double)(t3 - t13); gap += (double)(t0 - p3);
                layer += (double)(t3 - p3); waitw[wv] += (double)(t13 - t2); loopw[wv] += (double)(t2 - t1);
                mn = std::min(mn, t2); mx = std::max(mx, t2); ++cnt;
                if (CZ_T8_TRACE >= 2) { u64 pv = t1; for (int k = 0; k < 9; ++k) { tapd[k] += (double)(at(g, wv, l, 4 + k) - pv); pv = at(g, wv, l, 4 + k); } }
            }
            skew += (double)(mx - mn); ++cntg;
        }
    const double L = layer / cnt;
    printf("per layer and wave, mean ticks (share of the layer): layer %.0f = top %.0f (%.1f%%) + main loop %.0f (%.1f%%) + wait at the barrier %.0f (%.1f%%) + epilogue incl. its barrier %.0f (%.1f%%) + gap %.0f\n",
           L, top / cnt, 100 * top / cnt / L, loop / cnt, 100 * loop / cnt / L, wait / cnt, 100 * wait / cnt / L, epi / cnt, 100 * epi / cnt / L, gap / cnt);
    printf("ideal MFMA time of a layer on one SIMD: 18 slabs x 48 MFMAs x 32 cycles = 27648 cycles; main loop / ideal = %.3f (if a tick is a shader cycle)\n", loop / cnt / 27648.0);
    printf("spread of the main-loop end over the 8 waves of a workgroup (max - min), mean: %.0f ticks\n", skew / cntg);
    printf("by wave: main loop"); for (int wv = 0; wv < W; ++wv) printf(" %.0f", loopw[wv] / (cnt / W)); printf("\n");
    printf("by wave: barrier wait"); for (int wv = 0; wv < W; ++wv) printf(" %.0f", waitw[wv] / (cnt / W)); printf("\n");
    if (CZ_T8_TRACE >= 2) { printf("by tap (2 slabs each; ideal 3072):"); for (int k = 0; k < 9; ++k) printf(" %.0f", tapd[k] / cnt); printf("\n"); }
    return 0;
}
