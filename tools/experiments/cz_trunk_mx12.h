// cz_trunk_mx12.h — round-6 experiment: k_trunk_mx_c128 (csrc/cz_trunk_mx.h: read that header for the arithmetic, the LDS layout and
// the slab format, all unchanged) with TWELVE waves per workgroup — three per SIMD — and two cell tiles per wave instead of eight
// waves with three: wave = 4 wr + ct, cell group wr = 0..2 (tiles 2 wr, 2 wr + 1), channel tile ct.  The same MFMAs per slab and
// SIMD (3 x 6 instead of 2 x 9), a third wave to cover the other two's waits; the price: every channel tile's weights are read by
// three cell groups instead of two (+9 % LDS read instructions), sixteen 1 KB DMA pieces over twelve waves (waves 0-3 issue two).
// Built by tools/experiments/mx_ablate.sh mx12=mx12 into a library that launches it with CCHESS_MX_KERNEL=12.
#pragma once
#include <type_traits>
#include "cz_trunk_mx.h"

namespace czconv {

#include "cz_trunk_mx12_asm.inc"

constexpr int MX12_THREADS = 768;

// planes [B][90][16] fp16 (0/1), w0 [9 taps][hi, lo][2 = ci/8][128 co][8] fp16 (the strict engine's first-layer pack),
// wpk [L][36 slabs][16384 B] (net.py: mx_pack_layer), bias / b0 fp32; out: trunk [B][90][128] FP32 or NULL; head_out [B][90][3].
__global__ __launch_bounds__(768, 3) void k_trunk_mx12_c128(const unsigned char *__restrict__ wpk,
                                                           const float *__restrict__ bias,
                                                           float *__restrict__ out,
                                                           const float *__restrict__ head_w,
                                                           const float *__restrict__ head_b,
                                                           float *__restrict__ head_out,
                                                           const uint16_t *__restrict__ planes,
                                                           const uint16_t *__restrict__ w0,
                                                           const float *__restrict__ b0,
                                                           int B, int nlayers,
                                                           const int *__restrict__ bcount,
                                                           unsigned long long *__restrict__ clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Geo = MXGeo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, ct = wave & 3;      // wr = 0..2
    const int l31 = lane & 31, khalf = lane >> 5;
    const int pos0 = blockIdx.x * Geo::P;
    if (bcount) {
        const int live = *bcount;
        B = live < B ? live : B;
    }
    if (pos0 >= B) return;
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (clk) { clk_c0 = __builtin_readcyclecounter(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    const int npos = (B - pos0) < Geo::P ? (B - pos0) : Geo::P;
    const int nrows = npos * 90;
    const int nslabs = nlayers * Geo::SLABS_PER_LAYER;
    auto lds_addr = [](int row_byte_off, int c) { return row_byte_off + ((c ^ ((row_byte_off >> 8) & 15)) << 4); };
    const unsigned voff0 = (unsigned)tid << 4, voff1 = voff0 + 12288u;      // voff1: waves 0-3 only (pieces 12..15)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto lds_row_of = [](int natural) {   // natural = p * 90 + y * 10 + x  ->  rank-major row 20 y + 10 p + x
        const int p = natural / 90, c = natural - p * 90, y = c / 10, x = c - y * 10;
        return 20 * y + 10 * p + x;
    };

    auto dma_slab = [&](int slab) {   // prologue only; the loop issues its DMAs from the slab asm
        const unsigned char *src = wpk + (size_t)slab * Geo::SLAB_BYTES;
        unsigned char *dst = smem + Geo::W_OFF + ((unsigned)slab & 3u) * Geo::SLAB_BYTES + (wave_u << 10);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + voff0),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        if (wave_u < 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + voff1),
                                             (__attribute__((address_space(3))) void *)(dst + 12288), 16, 0, 0);
    };
    for (int q = 0; q < 3; ++q) dma_slab(q < nslabs ? q : nslabs - 1);
    {
        const uint4 *g = reinterpret_cast<const uint4 *>(planes + (size_t)pos0 * 90 * 16);
        for (int idx = tid; idx < Geo::ROWS * 2; idx += MX12_THREADS) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < nrows * 2) v = g[idx];
            *reinterpret_cast<uint4 *>(smem + Geo::PLANES_OFF + (idx << 4)) = v;
        }
    }
    // zero row of the hi halves; zero aliases 192 .. 223 of the X / Y planes; scale 127 (= 2^0) in the aliases of SC
    if (tid < 16) *reinterpret_cast<uint4 *>(smem + Geo::ZERO_OFF + (tid << 4)) = make_uint4(0, 0, 0, 0);
    if (tid < 256) *reinterpret_cast<uint4 *>(smem + Geo::X_OFF + (tid >> 5) * Geo::XPLANE + (192 + (tid & 31)) * 16) = make_uint4(0, 0, 0, 0);
    else *reinterpret_cast<uint2 *>(smem + Geo::Y_OFF + ((tid - 256) >> 5) * Geo::YPLANE + (192 + (tid & 31)) * 8) = make_uint2(0, 0);
    if (tid < 64) *reinterpret_cast<uint32_t *>(smem + Geo::S_OFF + (tid >> 5) * Geo::SPLANE + (192 + (tid & 31)) * 4) = 0x7f7f7f7fu;
    if (head_out && tid < 3 * 128 / 4)
        reinterpret_cast<float4 *>(smem + Geo::HEADW_OFF)[tid] = reinterpret_cast<const float4 *>(head_w)[tid];
    bf16x8 wf[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int part = 0; part < 2; ++part)
            wf[t][part] = *reinterpret_cast<const bf16x8 *>(w0 + ((size_t)(((t * 2 + part) * 2 + khalf) * 128 + ct * 32 + l31) << 3));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int rowk[2], tapmask[2], natb[2];   // rowk: the lane's LDS row in tile i (-1: padding row)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = 32 * (wr * 2 + i) + l31 - 12;
        const int kk = k < 0 ? 0 : k;
        const int h = kk / 20, rem = kk - h * 20, pp = rem / 10, w = rem - pp * 10;
        rowk[i] = k;
        natb[i] = (pp * 90 + h * 10 + w) * 32;
        int m = 0;
        for (int t = 0; t < 9; ++t) {
            const int y = h + t / 3 - 1, x = w + t % 3 - 1;
            if (k >= 0 && y >= 0 && y < 9 && x >= 0 && x < 10) m |= 1 << t;
        }
        tapmask[i] = m;
    }
    // per tap: ab / key = hi row address and swizzle key (cz_trunk_split.h); xr = khalf * XPLANE + 16 * (row or its zero alias):
    // X block at xr + quarter * 2 XPLANE, Y at Y_OFF + xr / 2 + ..., scale dword at S_OFF + xr / 4
    auto tap_addr = [&](int tap, int (&ab)[2], int (&key)[2], int (&xr)[2], int (&yr)[2]) {
        const int delta = (tap / 3 - 1) * 20 + (tap - (tap / 3) * 3 - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // a padding lane (rowk < 0, every tap masked) keeps its OWN virtual row: with row 0 for all twelve of them their zero-row
            // slot / alias entry collided with the real row lane 12 reads (8.5 % of the LDS cycles in bank conflicts, round 5)
            const int kk = rowk[i] + delta;
            const bool on = (tapmask[i] >> tap) & 1;
            const int rb = Geo::HI_OFF + kk * CV_ROWB;
            ab[i] = on ? rb : Geo::ZERO_OFF;
            key[i] = ((rb >> 8) & 15) ^ khalf;
            xr[i] = khalf * Geo::XPLANE + ((on ? kk : 192 + (kk & 31)) << 4);
            yr[i] = xr[i] >> 1;
        }
    };
    const int vb0 = Geo::W_OFF + khalf * 2048 + ((ct * 32 + l31) << 4);
    const int vy0 = Geo::W_OFF + 12288 + khalf * 1024 + ((ct * 32 + l31) << 3);
    const int vs0 = Geo::W_OFF + 14336 + khalf * 512 + ((ct * 32 + l31) << 2);
    int keep;

    int rkk[2];
    auto refresh_rk = [&]() {   // opaque copies: keeps the epilogue addresses out of registers across the slab loop
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rkk[i] = rowk[i] < 0 ? 0 : rowk[i];
            asm volatile("" : "+v"(rkk[i]));
        }
    };
    f32x16 xreg[2];   // block input x (fp32) at this lane's accumulator positions
    float4 bqn[4];   // the NEXT tower layer's bias at this lane's accumulator positions, requested before the epilogue of the layer in
                     // front of it (round 6: the four loads at the top of a layer cost their full L2 latency once per layer)
    auto load_bias = [&](const float *bl) {
        // the lane's offset is recomputed from an opaque copy of the thread index: kept across the slab loops as a 64-bit per-lane
        // pointer it would be parked in scratch and cost a dependent global round trip per layer, which is what this is here to remove
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        const int boff = ((t >> 6) & 3) * 32 + 4 * ((t & 63) >> 5);
#pragma unroll
        for (int q = 0; q < 4; ++q) bqn[q] = *reinterpret_cast<const float4 *>(bl + boff + 8 * q);
    };
    auto init_acc = [&](f32x16 (&acc)[2], const float *bl, bool add_x) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bq = bl ? *reinterpret_cast<const float4 *>(bl + ct * 32 + 8 * q + 4 * khalf) : bqn[q];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                acc[i][4 * q + 0] = bq.x + (add_x ? xreg[i][4 * q + 0] : 0.0f);
                acc[i][4 * q + 1] = bq.y + (add_x ? xreg[i][4 * q + 1] : 0.0f);
                acc[i][4 * q + 2] = bq.z + (add_x ? xreg[i][4 * q + 2] : 0.0f);
                acc[i][4 * q + 3] = bq.w + (add_x ? xreg[i][4 * q + 3] : 0.0f);
            }
        }
    };
    // epilogue of one tile: ReLU + clamp, hi = rn16(v) to the HI row, the fp6 block [hi | 2^11 (v - hi)] under the scale
    // 2^(exponent(max v) - 2) to the X / Y planes of group 2 ct + khalf, the scale byte to SC; keep_x: v is a block input
    auto store_tile = [&](f32x16 a, int i, bool keep_x) {
        f32x16 v;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = __builtin_amdgcn_fmed3f(a[r], 0.0f, 65504.0f);
        if (keep_x) xreg[i] = v;
        uint32_t pk[8];
        f32x16 hi, lo;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            pk[r >> 1] = pack_pair<true>(f32x2{v[r], v[r + 1]});
            const f32x2 u = unpack_pair<true>(pk[r >> 1]);
            hi[r] = u[0]; hi[r + 1] = u[1];
            lo[r] = (v[r] - u[0]) * 2048.0f; lo[r + 1] = (v[r + 1] - u[1]) * 2048.0f;
        }
        float m = v[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m = fmaxf(m, v[r]);
        int byte = (int)(__float_as_uint(m) >> 23) - 2;
        byte = byte < 1 ? 1 : byte;
        const u32x6 blk = mx_cvt6(hi, lo, __uint_as_float((uint32_t)byte << 23));
        const bool live = rowk[i] >= 0;
        if (live) {
            const int rowb = Geo::HI_OFF + rkk[i] * CV_ROWB;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n0 = ct * 32 + 8 * q + 4 * khalf;
                *reinterpret_cast<uint2 *>(smem + lds_addr(rowb, n0 >> 3) + ((n0 & 4) << 1)) = make_uint2(pk[2 * q], pk[2 * q + 1]);
            }
            const int g = 2 * ct + khalf;
            *reinterpret_cast<uint4 *>(smem + Geo::X_OFF + g * Geo::XPLANE + (rkk[i] << 4)) = make_uint4(blk[0], blk[1], blk[2], blk[3]);
            *reinterpret_cast<uint2 *>(smem + Geo::Y_OFF + g * Geo::YPLANE + (rkk[i] << 3)) = make_uint2(blk[4], blk[5]);
            smem[Geo::S_OFF + khalf * Geo::SPLANE + (rkk[i] << 2) + ct] = (unsigned char)byte;
        }
    };
    // the last layer: fp32 rows for the heads / the trunk dump, 16-byte chunk c of a row at c ^ (row & 31)
    auto store_tile_f32 = [&](f32x16 a, int i) {
        if (rowk[i] < 0) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 8 * ct + 2 * q + khalf;
            *reinterpret_cast<float4 *>(smem + rkk[i] * Geo::F32_ROWB + ((c ^ (rkk[i] & 31)) << 4)) =
                make_float4(fmaxf(a[4 * q + 0], 0.0f), fmaxf(a[4 * q + 1], 0.0f), fmaxf(a[4 * q + 2], 0.0f), fmaxf(a[4 * q + 3], 0.0f));
        }
    };

    {   // first layer: conv3x3(14 -> 128) + BN + ReLU; the planes are exact in 16 bits, the weights are hi + lo (two fp16 MFMAs)
        f32x16 acc[2];
        init_acc(acc, b0, false);
        load_bias(bias);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int shift = (t / 3 - 1) * 10 + (t % 3 - 1);
            bf16x8 af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int a = ((tapmask[i] >> t) & 1) ? Geo::PLANES_OFF + (natb[i] + shift * 32) : Geo::ZERO_OFF;
                af[i] = *reinterpret_cast<const bf16x8 *>(smem + a + khalf * 16);
            }
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i] = mfma_32x32x16<true>(wf[t][part], af[i], acc[i]);
        }
        refresh_rk();
#pragma unroll
        for (int i = 0; i < 2; ++i) store_tile(acc[i], i, true);
        __syncthreads();
    }

#define MX_OPERANDS(NAB, NKEY)                                                                                       \
            : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]),                                                                   \
              [a0h0] "+v"(fa.a[0]), [a0h1] "+v"(fa.a[1]), [w0] "+v"(fa.w),                                            \
              [a1h0] "+v"(fb.a[0]), [a1h1] "+v"(fb.a[1]), [w1] "+v"(fb.w),                                            \
              [sb0] "+v"(sb[0]), [sb1] "+v"(sb[1]),                                                                   \
              [t0] "=&v"(t0), [t1] "=&v"(t1), [ws] "=&v"(wsr), [keep] "=&s"(keep)                                      \
            : [ab0] "v"(ab[0]), [ab1] "v"(ab[1]), [key0] "v"(key[0]), [key1] "v"(key[1]),                              \
              [nab0] "v"(NAB[0]), [nab1] "v"(NAB[1]), [nkey0] "v"(NKEY[0]), [nkey1] "v"(NKEY[1]),                      \
              [xr0] "v"(xr[0]), [xr1] "v"(xr[1]), [yr0] "v"(yr[0]), [yr1] "v"(yr[1]), [vb] "v"(vb), [vy] "v"(vy), [vs] "v"(vs), \
              [vbn] "v"(vbn), [voff0] "v"(voff0), [voff1] "v"(voff1), [sbase] "s"(sbase), [ldst] "s"(ldst)
#define MX_CLOBBERS "memory", "scc", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", \
                    "v156", "v157", "v158", "v159", "v160", "v161"
#define MX_ARGS()                                                                                                    \
            const int slot = ((unsigned)g & 3u) << Geo::SLAB_SHIFT;                                                  \
            const int vb = vb0 + slot, vy = vy0 + slot, vs = vs0 + slot;                                             \
            const int vbn = vb0 + ((((unsigned)g + 1u) & 3u) << Geo::SLAB_SHIFT);                                    \
            const int gn = g + 3 < nslabs ? g + 3 : nslabs - 1;                                                      \
            const unsigned char *sbase = wpk + (size_t)gn * Geo::SLAB_BYTES;                                         \
            const int ldst = Geo::W_OFF + ((((unsigned)g + 3u) & 3u) << Geo::SLAB_SHIFT) + (wave_u << 10);
#define MX_RUN(ASMSTR, NAB, NKEY)                                                                                    \
        {                                                                                                            \
            MX_ARGS()                                                                                                \
            asm volatile(ASMSTR MX_OPERANDS(NAB, NKEY) : MX_CLOBBERS);                                               \
            ++g;                                                                                                     \
        }
#define MX_RUNV(ASMSTR, NAB, NKEY)   /* the same + the wave-uniform skip mask; clobbers VCC */                        \
        {                                                                                                            \
            MX_ARGS()                                                                                                \
            asm volatile(ASMSTR MX_OPERANDS(NAB, NKEY), [skipm] "s"(skipm) : MX_CLOBBERS, "vcc");                    \
            ++g;                                                                                                     \
        }

#ifdef MX_TIMING   /* diagnostic build (tools/mx_timing.py): where a workgroup's cycles go; the clock-probe buffer carries the sums */
    unsigned long long tm_loop = 0, tm_epi = 0, tm_set = 0, tm_t0 = __builtin_readcyclecounter();
    const unsigned long long tm_first = tm_t0 - clk_c0;
#undef MX_TM
#define MX_TM(acc) { const unsigned long long n_ = __builtin_readcyclecounter(); acc += n_ - tm_t0; tm_t0 = n_; }
#else
#undef MX_TM
#define MX_TM(acc)
#endif
    struct MxFrag { bf16x8 a[2], w; };
    int skipm;   // cell group 0 (waves 0..3): its first row tile skips the dy = -1 taps
    asm volatile("s_cmp_lt_u32 %1, 4\n\ts_cselect_b32 %0, 1, 0" : "=s"(skipm) : "s"(wave_u) : "scc");
    // two wave classes, a wave-uniform branch around the whole tower: waves 0-3 (cell group 0) issue two DMA pieces per slab and
    // skip their first tile's MFMAs in the dy = -1 taps; waves 4-11 issue one piece
#define B12(Q, NAB, NKEY) if constexpr (W0) MX_RUN(MX12_SLAB_P2_##Q, NAB, NKEY) else MX_RUN(MX12_SLAB_P1_##Q, NAB, NKEY)
#define B12S(Q, NAB, NKEY) if constexpr (W0) MX_RUNV(MX12_SKIP0_P2_##Q, NAB, NKEY) else MX_RUN(MX12_SLAB_P1_##Q, NAB, NKEY)
    auto tower = [&](auto w0c) {
    constexpr bool W0 = decltype(w0c)::value;
    int g = 0;
#pragma unroll 1
    for (int layer = 0; layer < nlayers; ++layer) {
        f32x16 acc[2];
        init_acc(acc, nullptr, (layer & 1) != 0);
        int ab[2], key[2], nab[2], nkey[2], xr[2], yr[2], nxr[2], nyr[2], t0, t1, wsr;
        int sb[2] = {0, 0};
        MxFrag fa, fb;
        tap_addr(0, ab, key, xr, yr);
        {   // the first slab's two fp16 operand sets (waited for by its steps A and B)
            const int vb = vb0 + (((unsigned)g & 3u) << Geo::SLAB_SHIFT);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa.a[i] = *reinterpret_cast<const bf16x8 *>(smem + ab[i] + ((0 ^ key[i]) << 4));
                fb.a[i] = *reinterpret_cast<const bf16x8 *>(smem + ab[i] + ((2 ^ key[i]) << 4));
            }
            fa.w = *reinterpret_cast<const bf16x8 *>(smem + vb);
            fb.w = *reinterpret_cast<const bf16x8 *>(smem + vb + 4096);
            asm volatile("" : "+v"(fa.a[0]), "+v"(fa.a[1]), "+v"(fa.w), "+v"(fb.a[0]), "+v"(fb.a[1]), "+v"(fb.w));
        }
        MX_TM(tm_set)
        int tap = 0;
#pragma unroll 1
        for (; tap < 3; ++tap) {   // dy = -1: cell group 0 branches around the MFMAs of its all-rank-0 row tile
            B12S(Q0, ab, key)
            B12S(Q1, ab, key)
            B12S(Q2, ab, key)
            tap_addr(tap + 1, nab, nkey, nxr, nyr);
            B12S(Q3, nab, nkey)
#pragma unroll
            for (int i = 0; i < 2; ++i) { ab[i] = nab[i]; key[i] = nkey[i]; xr[i] = nxr[i]; yr[i] = nyr[i]; }
        }
#pragma unroll 1
        for (; tap < 8; ++tap) {
            B12(Q0, ab, key)
            B12(Q1, ab, key)
            B12(Q2, ab, key)
            tap_addr(tap + 1, nab, nkey, nxr, nyr);
            B12(Q3, nab, nkey)
#pragma unroll
            for (int i = 0; i < 2; ++i) { ab[i] = nab[i]; key[i] = nkey[i]; xr[i] = nxr[i]; yr[i] = nyr[i]; }
        }
        B12(Q0, ab, key)      // tap 8; its last slab requests nothing
        B12(Q1, ab, key)
        B12(Q2, ab, key)
        B12(Q3_LAST, ab, key)
        const bool last = layer + 1 == nlayers;
        if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the fp32 rows reach into the weight ring
        else load_bias(bias + (layer + 1) * 128);
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // let the MFMAs retire (the compiler does not see them)
        MX_TM(tm_loop)
        refresh_rk();
        if (last) {
#pragma unroll
            for (int i = 0; i < 2; ++i) store_tile_f32(acc[i], i);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) store_tile(acc[i], i, (layer & 1) != 0);
        }
        __syncthreads();
        MX_TM(tm_epi)
    }
    };   // tower
    if (wave_u < 4) tower(std::integral_constant<bool, true>{});
    else tower(std::integral_constant<bool, false>{});
#undef B12
#undef B12S
    if (clk && tid == 0) {
        clk[blockIdx.x * 4 + 0] = clk_c0; clk[blockIdx.x * 4 + 1] = __builtin_readcyclecounter();
        clk[blockIdx.x * 4 + 2] = clk_r0; clk[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
#ifdef MX_TIMING
        clk[blockIdx.x * 4 + 0] = tm_first; clk[blockIdx.x * 4 + 1] = tm_loop; clk[blockIdx.x * 4 + 2] = tm_epi; clk[blockIdx.x * 4 + 3] = tm_set;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (out) {   // trunk activations as fp32, 4 channels per thread and step
        float4 *go = reinterpret_cast<float4 *>(out + (size_t)pos0 * 90 * 128);
        for (int idx = tid; idx < nrows * 32; idx += MX12_THREADS) {
            const int r = idx >> 5, c = idx & 31, k = lds_row_of(r);
            go[idx] = *reinterpret_cast<const float4 *>(smem + k * Geo::F32_ROWB + ((c ^ (k & 31)) << 4));
        }
    }
    if (head_out) {
        const float *hw = reinterpret_cast<const float *>(smem + Geo::HEADW_OFF);
        // one thread per board cell, all three head channels; chunks in a fixed order: a position's outputs do not depend on
        // the row / workgroup it lands on
        for (int r = tid; r < nrows; r += MX12_THREADS) {
            const int k = lds_row_of(r);
            float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
#pragma unroll 4
            for (int c = 0; c < 32; ++c) {
                const float4 e = *reinterpret_cast<const float4 *>(smem + k * Geo::F32_ROWB + ((c ^ (k & 31)) << 4));
                const float *w0_ = hw + c * 4, *w1_ = hw + 128 + c * 4, *w2_ = hw + 256 + c * 4;
                acc0 += e.x * w0_[0] + e.y * w0_[1] + e.z * w0_[2] + e.w * w0_[3];
                acc1 += e.x * w1_[0] + e.y * w1_[1] + e.z * w1_[2] + e.w * w1_[3];
                acc2 += e.x * w2_[0] + e.y * w2_[1] + e.z * w2_[2] + e.w * w2_[3];
            }
            float *o = head_out + ((size_t)pos0 * 90 + r) * 3;
            o[0] = fmaxf(acc0 + head_b[0], 0.f);
            o[1] = fmaxf(acc1 + head_b[1], 0.f);
            o[2] = fmaxf(acc2 + head_b[2], 0.f);
        }
    }
}
#undef MX_OPERANDS
#undef MX_CLOBBERS
#undef MX_ARGS
#undef MX_RUN
#undef MX_RUNV

}  // namespace czconv
