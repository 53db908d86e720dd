// tools/mfma_peak.hip — what the MFMA pipe sustains on this chip with nothing else going on:
// back-to-back v_mfma_f32_32x32x16_bf16 on 12 independent accumulators, operands in registers.
// args: waves_per_simd(1|2) iters data(0 = random bf16, 1 = zeros, 2 = random with half the elements zero)
//       order(0 = b outer / a inner, 1 = every MFMA the same two operands, 2 = snake: one operand changes per MFMA,
//             3 = both operands change with every MFMA) — does the operand sequence matter for the power-limited rate?
//       dtype(0 = bf16, 1 = fp16: the same bit patterns read as IEEE half — 10 random mantissa bits instead of 7: round 3,
//             is the fp16 tower's ~3 % deficit against bf16 the multipliers' power?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int ORDER, bool F16>
__global__ __launch_bounds__(512, 1) void k_peak(const uint4 *__restrict__ src, float *__restrict__ dst, int iters) {
    bf16x8 a[3], b[4];
    for (int i = 0; i < 3; ++i) a[i] = __builtin_bit_cast(bf16x8, src[(threadIdx.x + 64 * i) & 1023]);
    for (int j = 0; j < 4; ++j) b[j] = __builtin_bit_cast(bf16x8, src[(threadIdx.x + 64 * (j + 3)) & 1023]);
    f32x16 acc[3][4];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x16{};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                int i, j, ai, bj;
                if (ORDER == 0) { j = k / 3; i = k % 3; ai = i; bj = j; }
                else if (ORDER == 1) { j = k / 3; i = k % 3; ai = 0; bj = 0; }
                else if (ORDER == 2) { j = k / 3; i = (j & 1) ? 2 - k % 3 : k % 3; ai = i; bj = j; }
                else { i = k % 3; j = k % 4; ai = i; bj = j; }
                if constexpr (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, b[bj]), __builtin_bit_cast(f16x8, a[ai]), acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[bj], a[ai], acc[i][j], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    dst[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1, iters = argc > 2 ? atoi(argv[2]) : 20000, zero = argc > 3 ? atoi(argv[3]) : 0, order = argc > 4 ? atoi(argv[4]) : 0, f16 = argc > 5 ? atoi(argv[5]) : 0;
    uint4 *src; float *dst;
    CK(hipMalloc(&src, 1024 * 16)); CK(hipMalloc(&dst, 4096 * 512 * 4));
    unsigned short h[8192];
    srand(1);
    for (int i = 0; i < 8192; ++i) h[i] = (zero == 1 || (zero == 2 && (rand() & 1))) ? 0 : (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    CK(hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice));
    const int grid = 256 * 4, threads = 256 * wps;   // 4 rounds of one workgroup per CU
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        if (f16) hipLaunchKernelGGL((k_peak<0, true>), dim3(grid), dim3(threads), 0, 0, src, dst, iters);
        else if (order == 0) hipLaunchKernelGGL((k_peak<0, false>), dim3(grid), dim3(threads), 0, 0, src, dst, iters);
        else if (order == 1) hipLaunchKernelGGL((k_peak<1, false>), dim3(grid), dim3(threads), 0, 0, src, dst, iters);
        else if (order == 2) hipLaunchKernelGGL((k_peak<2, false>), dim3(grid), dim3(threads), 0, 0, src, dst, iters);
        else hipLaunchKernelGGL((k_peak<3, false>), dim3(grid), dim3(threads), 0, 0, src, dst, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)grid * (threads / 64) * iters * 48.0 * 32 * 32 * 16 * 2;
        printf("waves/SIMD=%d zero=%d order=%d %s: %.3f ms  %.1f TF/s\n", wps, zero, order, f16 ? "fp16" : "bf16", ms, flops / ms / 1e9);
    }
    return 0;
}
