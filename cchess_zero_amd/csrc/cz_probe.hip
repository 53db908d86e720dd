// cz_probe.hip — measurement entry points for bench.py (not on the search path).
//
// cz_probe_mfma_peak: what the MFMA pipes of THIS chip sustain, now, with nothing else going on: back-to-back
// v_mfma_f32_32x32x16 on 12 independent accumulators per wave, two waves per SIMD, operands in registers, one workgroup
// per CU and round.  On dense random operands the chip's power governor holds this well below the nominal 2.5 PFLOP/s
// (1.7-1.8 measured in rounds 2-3, tools/mfma_peak.hip); bench.py runs it for ~50 ms beside the timed region so that the
// roofline line carries the box's own practical ceiling instead of a constant.
#include "cz_internal.h"
#include <stdlib.h>

typedef __bf16 pk_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pk_f16x8 __attribute__((ext_vector_type(8)));
typedef float pk_f32x16 __attribute__((ext_vector_type(16)));

template <bool F16>
__global__ __launch_bounds__(512, 2) void k_mfma_peak(const uint4 *__restrict__ src, float *__restrict__ dst, int iters) {
    pk_bf16x8 a[3], b[4];
    for (int i = 0; i < 3; ++i) a[i] = __builtin_bit_cast(pk_bf16x8, src[(threadIdx.x + 64 * i) & 1023]);
    for (int j = 0; j < 4; ++j) b[j] = __builtin_bit_cast(pk_bf16x8, src[(threadIdx.x + 64 * (j + 3)) & 1023]);
    pk_f32x16 acc[3][4];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = pk_f32x16{};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int j = k / 3, i = k % 3;
                if constexpr (F16)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pk_f16x8, b[j]), __builtin_bit_cast(pk_f16x8, a[i]), acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// dtype CZ_BF16 | CZ_F16; data: 0 = random mantissas and signs, magnitudes in [1, 2) (dense operands), 1 = all zero,
// 2 = random with half the elements zero (post-ReLU look); iters: 48 MFMAs per wave and iteration.
// *tflops <- dense-equivalent TFLOP/s of the launch, *ms <- its duration.  Synchronises the context's stream.
extern "C" int cz_probe_mfma_peak(cz_ctx *c, int dtype, int data, int iters, double *tflops, double *ms) {
    CZ_REQUIRE(c && tflops && ms && iters > 0 && (dtype == CZ_BF16 || dtype == CZ_F16) && data >= 0 && data <= 2,
               "cz_probe_mfma_peak: bad argument");
    CZ_HIP(hipSetDevice(c->device));
    uint4 *src = nullptr;
    float *dst = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const int grid = 256 * 4, threads = 512;
    float best = 0.f;
    // every HIP call through `step`: the first failure is remembered and the buffers / events are released on every path
    hipError_t bad = hipSuccess;
    auto step = [&](hipError_t e) { if (bad == hipSuccess && e != hipSuccess) bad = e; return bad == hipSuccess; };
    if (step(hipMalloc(&src, 1024 * 16)) && step(hipMalloc(&dst, (size_t)grid * threads * 4))) {
        unsigned short h[8192];
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
        for (int i = 0; i < 8192; ++i) {
            const unsigned r0 = rnd(), r1 = rnd(), r2 = rnd();
            h[i] = (data == 1 || (data == 2 && (r0 & 1))) ? 0 : (unsigned short)(0x3c00 + (r1 & 0x3ff) + ((r2 & 1) << 15));
        }
        step(hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice));
        step(hipEventCreate(&e0));
        step(hipEventCreate(&e1));
        for (int rep = 0; rep < 2 && bad == hipSuccess; ++rep) {   // the first launch loads the code object and lets the clocks settle
            step(hipEventRecord(e0, c->stream));
            if (dtype == CZ_F16) hipLaunchKernelGGL((k_mfma_peak<true>), dim3(grid), dim3(threads), 0, c->stream, src, dst, iters);
            else hipLaunchKernelGGL((k_mfma_peak<false>), dim3(grid), dim3(threads), 0, c->stream, src, dst, iters);
            step(hipEventRecord(e1, c->stream));
            step(hipEventSynchronize(e1));
            step(hipEventElapsedTime(&best, e0, e1));
        }
        step(hipGetLastError());
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    if (bad != hipSuccess) { cz_set_error(hipGetErrorString(bad)); return CZ_EHIP; }
    *ms = best;
    *tflops = (double)grid * (threads / 64) * iters * 48.0 * 32 * 32 * 16 * 2 / (best * 1e-3) / 1e12;
    return CZ_OK;
}
