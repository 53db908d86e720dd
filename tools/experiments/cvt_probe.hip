// cvt_probe.hip — round 6: v_cvt_scalef32_pk32_fp6_f16 measured before the 4-positions mx kernel is built on it.
//   part 1  semantics: slot i of the 192-bit result = E2M3(RNE, saturating)(in[i] / 2^floor(log2 scale)), in[i] = the i-th fp16 of the
//           16 source registers (low half first)?  Checked on the host against that hypothesis.
//   part 2  throughput: back-to-back independent conversions, 2 waves per SIMD, cycles per conversion
// build: hipcc --offload-arch=gfx950 -O3 -o bin/cvt_probe cvt_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstdint>
#include <cstring>
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ __forceinline__ u32x6 cvt32(u32x16 a, float s) {
    u32x6 r;
    asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(r) : "v"(a), "v"(s));
    return r;
}
__global__ void k_cvt(const unsigned *in, const float *scale, unsigned *out) {
    const int l = threadIdx.x;
    u32x16 a;
    for (int i = 0; i < 16; ++i) a[i] = in[l * 16 + i];
    u32x6 r = cvt32(a, scale[l]);
    for (int i = 0; i < 6; ++i) out[l * 6 + i] = r[i];
}
__global__ __launch_bounds__(512, 2) void k_rate(const unsigned *in, unsigned *sink, int iters, unsigned long long *clk) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    u32x16 a, b;
    for (int i = 0; i < 16; ++i) { a[i] = in[(tid & 1023) * 16 + i]; b[i] = a[i] ^ 0x00010001u; }
    u32x6 r0 = {}, r1 = {}, r2 = {}, r3 = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %4, %6\n\tv_cvt_scalef32_pk32_fp6_f16 %1, %5, %6\n\t"
                     "v_cvt_scalef32_pk32_fp6_f16 %2, %4, %6\n\tv_cvt_scalef32_pk32_fp6_f16 %3, %5, %6"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a), "v"(b), "v"(1.0f));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
    unsigned acc = r0[0] ^ r1[1] ^ r2[2] ^ r3[3];
    if (acc == 0x12345u) sink[tid] = acc;
}
static float h2f(uint16_t h) {
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    float v = e == 0 ? ldexpf((float)m, -24) : (e == 31 ? (m ? NAN : INFINITY) : ldexpf((float)(m | 1024), e - 25));
    return s ? -v : v;
}
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static int e2m3(float x) {   // RNE onto the E2M3 grid, saturating at 7.5 -> code
    const int s = std::signbit(x) ? 32 : 0;
    float a = fabsf(x);
    if (std::isnan(a)) return s | 31;
    if (a > 7.5f) a = 7.5f;
    const int e = a >= 4 ? 2 : a >= 2 ? 1 : 0;            // binade (below 1: subnormal, step 1/8 as in binade 0)
    const float step = ldexpf(1.0f, e - 3);
    float q = nearbyintf(a / step) * step;
    if (q > 7.5f) q = 7.5f;
    int eb, mant;
    if (q < 1.0f) { eb = 0; mant = (int)(q * 8); }
    else { const int e2 = q >= 4 ? 2 : q >= 2 ? 1 : 0; eb = e2 + 1; mant = (int)((q / ldexpf(1.0f, e2) - 1.0f) * 8); }
    return s | (eb << 3) | mant;
}
int main(int argc, char **argv) {
    const int L = 64, iters = argc > 1 ? atoi(argv[1]) : 20000;
    std::vector<unsigned> in(L * 16);
    std::vector<float> sc(L, 1.0f);
    srand(7);
    for (int l = 0; l < L; ++l)
        for (int i = 0; i < 32; ++i) {
            float v = l == 0 ? 0.25f * i : l == 1 ? -0.25f * i : l == 2 ? 0.0625f * i + 0.03f : ((rand() & 0xffff) - 32768) / 4096.0f;
            if (l == 3) v = (i & 1 ? -1.f : 1.f) * (7.0f + 0.25f * i);
            uint16_t h = f2h(v);
            in[l * 16 + i / 2] = (i & 1) ? (in[l * 16 + i / 2] | ((unsigned)h << 16)) : (unsigned)h;
        }
    sc[4] = 2.0f; sc[5] = 0.5f; sc[6] = 4.0f; sc[7] = 3.0f; sc[8] = 0.25f; sc[9] = 1.5f; sc[10] = 1024.0f; sc[11] = 1.0f / 64;
    unsigned *din, *dout; float *dsc;
    CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dsc, L * 4)); CK(hipMalloc(&dout, L * 6 * 4));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, sc.data(), L * 4, hipMemcpyHostToDevice));
    k_cvt<<<1, L>>>(din, dsc, dout);
    std::vector<unsigned> out(L * 6);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < L; ++l) {
        const float sdiv = ldexpf(1.0f, (int)floorf(log2f(sc[l])));
        for (int i = 0; i < 32; ++i) {
            const uint16_t h = (in[l * 16 + i / 2] >> (16 * (i & 1))) & 0xffff;
            const int want = e2m3(h2f(h) / sdiv);
            const int bit = 6 * i;
            unsigned long long w = (unsigned long long)out[l * 6 + bit / 32] | ((bit / 32 + 1 < 6) ? ((unsigned long long)out[l * 6 + bit / 32 + 1] << 32) : 0ull);
            const int got = (int)((w >> (bit % 32)) & 63);
            if (got != want) { if (bad < 12) printf("  lane %d slot %d: in %g scale %g -> code %d, hypothesis %d\n", l, i, h2f(h), sc[l], got, want); ++bad; }
        }
    }
    printf("part 1: %d of %d slots differ from the hypothesis (slot i = E2M3_RNE_sat(in[i] / 2^floor(log2 scale)), in[i] = i-th fp16 of the 16 source registers)\n", bad, L * 32);
    const int grid = 256 * 4;
    unsigned *dsink; unsigned long long *dclk;
    CK(hipMalloc(&dsink, grid * 512 * 4)); CK(hipMalloc(&dclk, grid * 8));
    std::vector<unsigned> seed(1024 * 16);
    for (auto &x : seed) x = (f2h(((rand() & 0xffff) - 32768) / 8192.0f)) | ((unsigned)f2h(((rand() & 0xffff) - 32768) / 8192.0f) << 16);
    unsigned *dseed; CK(hipMalloc(&dseed, seed.size() * 4)); CK(hipMemcpy(dseed, seed.data(), seed.size() * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) {
        k_rate<<<grid, 512>>>(dseed, dsink, iters, dclk);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> clk(grid);
        CK(hipMemcpy(clk.data(), dclk, grid * 8, hipMemcpyDeviceToHost));
        double c = 0; for (auto x : clk) c += x; c /= grid;
        printf("part 2: %d iterations x 4 conversions per wave, 2 waves per SIMD: %.0f cycles per wave = %.1f cycles per conversion per wave (%.1f per SIMD-conversion)\n",
               iters, c, c / (4.0 * iters), c / (8.0 * iters));
    }
    return 0;
}
