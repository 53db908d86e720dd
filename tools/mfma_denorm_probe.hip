// tools/mfma_denorm_probe.hip — does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs?  The strict trunk
// (k_trunk_split_c128) stores x = hi + lo with lo = rn16(x - hi): for |x| < 0.125 the fp16 lo half is subnormal.
// A = the probe value in every element, B = 1.0: D[m][n] = 16 * a if the input survives, 0 if it is flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const unsigned short *av, float *out, int bf) {
    f16x8 a, b;
    unsigned short one = bf ? 0x3F80 : 0x3C00;
    for (int i = 0; i < 8; ++i) { a[i] = __builtin_bit_cast(_Float16, av[0]); b[i] = __builtin_bit_cast(_Float16, one); }
    f32x16 c = {};
    if (bf) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    unsigned short *d; float *o; hipMalloc(&d, 2); hipMalloc(&o, 4);
    const unsigned short probes[] = {0x0001, 0x0200, 0x03FF, 0x0400, 0x8001};   // fp16: min subnormal, 2^-15, max subnormal, min normal, -min subnormal
    const char *names[] = {"2^-24 (min subnormal)", "2^-15 (subnormal)", "max subnormal", "2^-14 (min normal)", "-2^-24"};
    for (int bf = 0; bf < 2; ++bf)
        for (int i = 0; i < 5; ++i) {
            hipMemcpy(d, &probes[i], 2, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, bf);
            float r; hipMemcpy(&r, o, 4, hipMemcpyDeviceToHost);
            double expect;
            if (!bf) { _Float16 h; memcpy(&h, &probes[i], 2); expect = 16.0 * (double)(float)h; }
            else { unsigned u = (unsigned)probes[i] << 16; float f; memcpy(&f, &u, 4); expect = 16.0 * (double)f; }
            printf("%s input 0x%04x %-22s: D = %.9g, exact %.9g -> %s\n", bf ? "bf16" : "fp16", probes[i], bf ? "(bf16 pattern)" : names[i], r, expect,
                   (double)r == expect ? "KEPT" : (r == 0.f ? "FLUSHED" : "OTHER"));
        }
    return 0;
}
