// mx_probe.hip — round 5: what the block-scaled fp6 / fp8 path of gfx950 does, measured before a kernel is built on it.
//   part 1  v_cvt_scalef32_2xpk16_fp6_f32: slot order, direction of the scale, rounding, saturation (raw dwords dumped)
//   part 2  v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 operands made by that instruction: which slots meet, what the E8M0
//           scale bytes and op_sel do (inputs and outputs dumped; tools/experiments/mx_probe_check.py holds the hypothesis)
//   part 3  sustained rate of MFMA mixes on the whole chip (2 waves per SIMD, register operands, random data):
//           9 fp16 (the strict kernel's k-step pair) / 6 fp16 + 3 fp6 / 6 fp16 + 3 fp8 / fp6 only / fp8 only / fp16 only
// build: tools/experiments/build.sh -> bin/mx_probe ; run on the GPU box: bin/mx_probe <out.bin> [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

// hipcc (ROCm 7.2) lets the builtin's destination overlap its sources (v[0:5] <- v[2:17], v[18:33]): the instruction writes
// its result while it still reads, and slots 12.. come out as garbage.  Inline asm with an early-clobber destination.
__device__ __forceinline__ u32x6 cvt6(f32x16 a, f32x16 b, float s) {
    u32x6 r;
    asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(s));
    return r;
}

__global__ void k_cvt(const float *in, const float *scale, unsigned *out) {
    const int l = threadIdx.x;
    f32x16 a, b;
    for (int i = 0; i < 16; ++i) { a[i] = in[l * 32 + i]; b[i] = in[l * 32 + 16 + i]; }
    u32x6 r = cvt6(a, b, scale[l]);
    for (int i = 0; i < 6; ++i) out[l * 6 + i] = r[i];
}

// one MFMA: lane l supplies A block from ina[l][32] (cvt with scale 1), B block from inb[l][32]; scale VGPRs sa[l], sb[l]
template <int OA, int OB> __global__ void k_mfma6(const float *ina, const float *inb, const int *sa, const int *sb, float *out) {
    const int l = threadIdx.x;
    f32x16 a0, a1, b0, b1;
    for (int i = 0; i < 16; ++i) { a0[i] = ina[l * 32 + i]; a1[i] = ina[l * 32 + 16 + i]; b0[i] = inb[l * 32 + i]; b1[i] = inb[l * 32 + 16 + i]; }
    u32x6 ra = cvt6(a0, a1, 1.0f), rb = cvt6(b0, b1, 1.0f);
    i32x8 A = {(int)ra[0], (int)ra[1], (int)ra[2], (int)ra[3], (int)ra[4], (int)ra[5], 0, 0};
    i32x8 B = {(int)rb[0], (int)rb[1], (int)rb[2], (int)rb[3], (int)rb[4], (int)rb[5], 0, 0};
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 2, 2, OA, sa[l], OB, sb[l]);
    for (int i = 0; i < 16; ++i) out[l * 16 + i] = c[i];
}

template <int MIX> __global__ __launch_bounds__(512, 2) void k_rate(const unsigned *seed, float *sink, int iters, unsigned long long *clk) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    f16x8 wh0, wh1, ah0[3], ah1[3];
    i32x8 wx, ax[3];
    const unsigned *s = seed + (tid & 4095) * 64;
    auto h8 = [&](int o) { f16x8 v; for (int i = 0; i < 8; ++i) v[i] = (_Float16)(((int)(s[o + i] & 1023) - 512) * (1.0f / 256)); return v; };
    wh0 = h8(0); wh1 = h8(8);
    for (int i = 0; i < 3; ++i) { ah0[i] = h8(16 + 8 * i); ah1[i] = h8(40 + 8 * i); }
    for (int j = 0; j < 8; ++j) { wx[j] = (int)(s[j] * 2654435761u); for (int i = 0; i < 3; ++i) ax[i][j] = (int)(s[8 + 8 * i + j] * 40503u + 17); }
    if (MIX == 2 || MIX == 4)   // fp8: keep the bytes away from NaN (0x7f / 0xff)
        for (int j = 0; j < 8; ++j) { wx[j] &= 0x77777777; for (int i = 0; i < 3; ++i) ax[i][j] &= 0x77777777; }
    const int sc = 0x7f7f7f7f;
    f32x16 c[3] = {};
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            if (MIX == 0 || MIX == 5) {   // nine fp16 MFMAs (MIX 5: six)
#pragma unroll
                for (int i = 0; i < 3; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, ah0[i], c[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 3; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, ah0[i], c[i], 0, 0, 0);
                if (MIX == 0) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, ah1[i], c[i], 0, 0, 0);
                }
            } else if (MIX == 1 || MIX == 2) {
#pragma unroll
                for (int i = 0; i < 3; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh0, ah0[i], c[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 3; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh1, ah1[i], c[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 3; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wx, ax[i], c[i], MIX == 1 ? 2 : 0, MIX == 1 ? 2 : 0, 0, sc, 0, sc);
            } else {   // 3: fp6 only, 4: fp8 only, 6: fp4 only
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int i = 0; i < 3; ++i) c[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wx, ax[i], c[i], MIX == 3 ? 2 : MIX == 4 ? 0 : 4, MIX == 3 ? 2 : MIX == 4 ? 0 : 4, 0, sc, 0, sc);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
    float acc = 0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 16; ++j) acc += c[i][j];
    if (acc == 1.2345f) sink[tid] = acc;
}

static FILE *fo;
static void dump(const char *tag, const void *p, size_t n) {
    char name[16] = {0}; strncpy(name, tag, 15);
    unsigned long long nn = n;
    fwrite(name, 1, 16, fo); fwrite(&nn, 8, 1, fo); fwrite(p, 1, n, fo);
}

template <int MIX> static void rate(const char *what, double passes_per_iter, int iters, const unsigned *dseed, float *dsink, unsigned long long *dclk) {
    const int grid = 256 * 4;
    std::vector<unsigned long long> clk(grid * 2);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        k_rate<MIX><<<grid, 512>>>(dseed, dsink, iters, dclk);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(clk.data(), dclk, clk.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0, ref = 0;
        for (int b = 0; b < grid; ++b) { cyc += clk[2 * b]; ref += clk[2 * b + 1]; }
        cyc /= grid; ref /= grid;
        const double ghz = cyc / (ref * 10.0);   // 100 MHz reference clock
        // per wave: iters * passes_per_iter passes of 4 cycles; 2 waves share a SIMD
        const double busy = 2.0 * iters * passes_per_iter * 4.0 / cyc;
        printf("  %-34s %8.3f ms  %10.0f cycles per wave  clock %.3f GHz  matrix-pipe busy %.3f of the cycles (if the pass counts are right)  %.1f passes/us/SIMD\n",
               what, ms, cyc, ghz, busy, 2.0 * iters * passes_per_iter / (ms * 1e3) * 4);
    }
}

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "mx_probe.bin";
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;   // 0: parts 1 and 2 only
    fo = fopen(path, "wb");
    if (!fo) { perror(path); return 2; }
    // ---- part 1
    {
        const int L = 64;
        std::vector<float> in(L * 32), sc(L, 1.0f);
        auto grid = [](int code) { int s = code >> 5, e = (code >> 3) & 3, m = code & 7; float v = e ? (1 + m / 8.0f) * (float)(1 << (e - 1)) : m / 8.0f; return s ? -v : v; };
        for (int l = 0; l < L; ++l) for (int i = 0; i < 32; ++i) in[l * 32 + i] = grid(i);                  // default: codes 0..31 in slot order
        for (int i = 0; i < 32; ++i) in[1 * 32 + i] = grid(32 + i);                                          // negatives
        for (int i = 0; i < 32; ++i) in[2 * 32 + i] = 0.5f * (grid(i) + grid(i + 1 < 32 ? i + 1 : 31));      // ties
        const float big[8] = {7.5f, 7.75f, 8.0f, 9.0f, 100.0f, 1e30f, INFINITY, NAN};
        for (int i = 0; i < 32; ++i) in[3 * 32 + i] = (i & 8 ? -1.0f : 1.0f) * big[i & 7];
        sc[4] = 2.0f; sc[5] = 0.5f; sc[6] = 4.0f; sc[7] = 3.0f; sc[8] = 0.25f; sc[9] = 1.5f;
        for (int i = 0; i < 32; ++i) in[10 * 32 + i] = 0.03125f * i + 0.01f;                                 // subnormal region, off-grid
        srand(5);
        for (int l = 11; l < L; ++l) { for (int i = 0; i < 32; ++i) in[l * 32 + i] = ((rand() & 0xffff) - 32768) / 4096.0f; sc[l] = (l & 1) ? 1.0f : 2.0f; }
        float *din, *dsc; unsigned *dout;
        CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dsc, sc.size() * 4)); CK(hipMalloc(&dout, L * 6 * 4));
        CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
        k_cvt<<<1, L>>>(din, dsc, dout);
        std::vector<unsigned> out(L * 6);
        CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
        dump("cvt_in", in.data(), in.size() * 4); dump("cvt_scale", sc.data(), sc.size() * 4); dump("cvt_out", out.data(), out.size() * 4);
        printf("part 1: lane 0 (codes 0..31, scale 1) -> %08x %08x %08x %08x %08x %08x\n", out[0], out[1], out[2], out[3], out[4], out[5]);
    }
    // ---- part 2
    {
        std::vector<float> ina(64 * 32), inb(64 * 32);
        std::vector<int> sa(64), sb(64);
        srand(9);
        auto gv = [](int r) { int code = r & 63; int s = code >> 5, e = (code >> 3) & 3, m = code & 7; float v = e ? (1 + m / 8.0f) * (float)(1 << (e - 1)) : m / 8.0f; return s ? -v : v; };
        for (auto &v : ina) v = gv(rand());
        for (auto &v : inb) v = gv(rand());
        for (int l = 0; l < 64; ++l) {
            sa[l] = (127 + (rand() % 5 - 2)) | ((127 + (rand() % 5 - 2)) << 8) | ((127 + (rand() % 5 - 2)) << 16) | ((127 + (rand() % 5 - 2)) << 24);
            sb[l] = (127 + (rand() % 5 - 2)) | ((127 + (rand() % 5 - 2)) << 8) | ((127 + (rand() % 5 - 2)) << 16) | ((127 + (rand() % 5 - 2)) << 24);
        }
        float *da, *db, *dout; int *dsa, *dsb;
        CK(hipMalloc(&da, ina.size() * 4)); CK(hipMalloc(&db, inb.size() * 4)); CK(hipMalloc(&dout, 64 * 16 * 4)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256));
        CK(hipMemcpy(da, ina.data(), ina.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, inb.data(), inb.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
        dump("mf_a", ina.data(), ina.size() * 4); dump("mf_b", inb.data(), inb.size() * 4); dump("mf_sa", sa.data(), 256); dump("mf_sb", sb.data(), 256);
        std::vector<float> out(64 * 16);
#define RUN(OA, OB) k_mfma6<OA, OB><<<1, 64>>>(da, db, dsa, dsb, dout); CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost)); dump("mf_d" #OA #OB, out.data(), out.size() * 4);
        RUN(0, 0) RUN(1, 0) RUN(0, 1) RUN(2, 3) RUN(3, 2)
        printf("part 2: dumped\n");
    }
    fclose(fo);
    // ---- part 3
    if (iters > 0) {
        std::vector<unsigned> seed(4096 * 64);
        srand(3);
        for (auto &v : seed) v = (unsigned)rand() * 2654435761u ^ (unsigned)rand();
        unsigned *dseed; float *dsink; unsigned long long *dclk;
        CK(hipMalloc(&dseed, seed.size() * 4)); CK(hipMalloc(&dsink, 1024 * 512 * 4)); CK(hipMalloc(&dclk, 1024 * 2 * 8));
        CK(hipMemcpy(dseed, seed.data(), seed.size() * 4, hipMemcpyHostToDevice));
        printf("part 3: 1024 workgroups x 512 threads (2 waves per SIMD, 4 rounds of the chip), %d iterations x 2 bodies\n", iters);
        // passes per body if fp16 32x32x16 = 8, fp8 32x32x64 = 16, fp6 / fp4 = 8
        rate<0>("9 fp16 (strict: 3 per product)", 2 * 9 * 8, iters, dseed, dsink, dclk);
        rate<5>("6 fp16", 2 * 6 * 8, iters, dseed, dsink, dclk);
        rate<1>("6 fp16 + 3 fp6 (mx6)", 2 * (6 * 8 + 3 * 8), iters, dseed, dsink, dclk);
        rate<2>("6 fp16 + 3 fp8 (mx8)", 2 * (6 * 8 + 3 * 16), iters, dseed, dsink, dclk);
        rate<3>("9 fp6", 2 * 9 * 8, iters, dseed, dsink, dclk);
        rate<4>("9 fp8", 2 * 9 * 16, iters, dseed, dsink, dclk);
        rate<6>("9 fp4", 2 * 9 * 8, iters, dseed, dsink, dclk);
    }
    return 0;
}
