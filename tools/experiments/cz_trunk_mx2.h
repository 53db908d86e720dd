// cz_trunk_mx2.h — N1m2: k_trunk_mx_c128's arithmetic with a 3 x 2 register tile per wave and the K range split between the two
// waves of a pair (round 6).
//
// What limited k_trunk_mx_c128 (profiles/r05s_pmc_sq_mx.json, r05d_mx_ablation.txt): 9 MFMAs per slab body carry the body's whole
// fixed work — 12 ds_read_b128 + 4 ds_read_b64 + the scale dwords, 12 address VALU, two LDS-DMA pieces, one barrier —, i.e. 5 other
// instructions per MFMA on each of the two waves of a SIMD: the MFMA pipe is busy 56 % of the time, the LDS 59 %.  The fast engine's
// body (cz_conv_kernel.h) has 24 MFMAs behind the same fixed work and keeps the pipe 78 % busy.  A 3 x 2 tile halves the operand
// reads per MFMA (3 activation + 2 weight fragments feed 6 MFMAs instead of 3 + 1 feeding 3), but two positions have only
// 6 cell tiles x 4 channel tiles = 4 such tiles — 4 waves, one per SIMD, nothing to cover a wait.  So the K range is split:
//   wave = 4 wr + 2 cg + kp: cell group wr (3 cell tiles), channel tiles 2 cg + j (j = 0, 1), of which it FINALISES tile j = kp;
//   of a layer's 36 slabs it computes the 18 with index = kp (mod 2): 12 fp16 + 6 fp6 MFMAs per body.
//   At the end of a layer the two waves of a pair (wave ^ 1) swap the partial sums of the tile the OTHER one finalises through
//   the (then dead) activation planes: 3 tiles out, 3 tiles in, 48 registers each way.
// Per 18 MFMAs: 15 ds_read_b128 + 5 ds_read_b64 + 2-5 ds_read_b32 (was 24 + 8 + 2-8), 18 address VALU (was 24-30), 4 DMA pieces
// (as before), ONE barrier (was two): 3.1 other instructions per MFMA instead of 5.1.
// The weight ring holds two PAIRS of slabs.  Body q (slabs 2 q, 2 q + 1) reads pair q's fp6 blocks in step A and pair q + 1's fp16
// halves in steps B / C (operands are requested one body ahead); its barrier, at the top of step B behind `s_waitcnt vmcnt(0)
// lgkmcnt(0)`, publishes pair q + 1 and guarantees that nobody still reads pair q — whose buffers the four DMA pieces a wave then
// issues (pair q + 2) overwrite.
// Registers: 96 accumulators; the block input x of a residual block (48 fp32 registers per lane in k_trunk_mx_c128) does not fit
// beside them and lives in GLOBAL memory between the epilogue that produces it and the epilogue that adds it, two layers later
// (12 coalesced 1 KB stores / loads per wave: 96 KB per workgroup and block, L2 / MALL resident; the loads are issued before the
// exchange and have its four barriers to arrive).
// LDS layout, slab format, precision: exactly k_trunk_mx_c128's (cz_trunk_mx.h) — same packed weights (net.py: mx_pack_layers),
// same CPU emulation (tests/mxemu.py); only the order of the fp32 additions differs (two partial sums per accumulator).
#pragma once
#include <type_traits>
#include "cz_trunk_mx.h"

namespace czconv {

#include "cz_trunk_mx2_asm.inc"

constexpr int MX2_XBUF_FLOATS_PER_WG = 8 * 48 * 64;   // 8 waves x 3 tiles x 16 registers x 64 lanes (98,304 B)

// A thread's place in the workgroup, RECOMPUTED from an opaque copy of threadIdx.x wherever the code outside the slab loops needs
// it: the loops run on every register a wave of two per SIMD has, so whatever is merely live ACROSS them is parked in scratch
// and comes back as one dependent global-memory round trip per use — the first version of this kernel spent as long at the ends
// of a layer (60 serialised scratch reloads) as in its slab loops.
struct Mx2Th {
    int tid, lane, wave, wr, cg, kp, ct, l31, khalf;
    __device__ __forceinline__ static Mx2Th here() {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        Mx2Th g;
        g.tid = t; g.lane = t & 63; g.wave = t >> 6; g.wr = g.wave >> 2; g.cg = (g.wave >> 1) & 1; g.kp = g.wave & 1;
        g.ct = 2 * g.cg + g.kp; g.l31 = g.lane & 31; g.khalf = g.lane >> 5;
        return g;
    }
};

// planes / w0 / wpk / bias / b0 / out / head_out: as k_trunk_mx_c128; xbuf: [gridDim.x][MX2_XBUF_FLOATS_PER_WG] fp32 scratch.
__global__ __launch_bounds__(512, 2) void k_trunk_mx2_c128(const unsigned char *__restrict__ wpk,
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
                                                            unsigned long long *__restrict__ clk,
                                                            float *__restrict__ xbuf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Geo = MXGeo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, cg = (wave >> 1) & 1, kp = wave & 1;
    const int ct = 2 * cg + kp;                      // the channel tile this wave finalises (j = 0); j = 1 is ct ^ 1
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
    const int nslabs = nlayers * Geo::SLABS_PER_LAYER;      // even
    auto lds_addr = [](int row_byte_off, int c) { return row_byte_off + ((c ^ ((row_byte_off >> 8) & 15)) << 4); };
    const unsigned voff0 = (unsigned)tid << 4, voff1 = voff0 + 8192u;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int kp_u = wave_u & 1;
    auto lds_row_of = [](int natural) {   // natural = p * 90 + y * 10 + x  ->  rank-major row 20 y + 10 p + x
        const int p = natural / 90, c = natural - p * 90, y = c / 10, x = c - y * 10;
        return 20 * y + 10 * p + x;
    };
    float *xw = xbuf + (size_t)blockIdx.x * MX2_XBUF_FLOATS_PER_WG + (size_t)wave * (48 * 64) + lane * 4;   // + (4 i + q) * 256 floats

    auto dma_slab = [&](int slab) {   // prologue only; the loop issues its DMAs from the body asm
        const unsigned char *src = wpk + (size_t)slab * Geo::SLAB_BYTES;
        unsigned char *dst = smem + Geo::W_OFF + ((unsigned)slab & 3u) * Geo::SLAB_BYTES + (wave_u << 10);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + voff0),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + voff1),
                                         (__attribute__((address_space(3))) void *)(dst + Geo::THREADS * 16), 16, 0, 0);
    };
    for (int q = 0; q < 3; ++q) dma_slab(q);
    {
        const uint4 *g = reinterpret_cast<const uint4 *>(planes + (size_t)pos0 * 90 * 16);
        for (int idx = tid; idx < Geo::ROWS * 2; idx += Geo::THREADS) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < nrows * 2) v = g[idx];
            *reinterpret_cast<uint4 *>(smem + Geo::PLANES_OFF + (idx << 4)) = v;
        }
    }
    // zero row of the hi halves; zero aliases 192 .. 223 of the X / Y planes; scale 127 (= 2^0) in the aliases of SC (again after
    // every exchange, which borrows the planes)
    auto init_aliases = [&](int tid) {
        if (tid < 256) *reinterpret_cast<uint4 *>(smem + Geo::X_OFF + (tid >> 5) * Geo::XPLANE + (192 + (tid & 31)) * 16) = make_uint4(0, 0, 0, 0);
        else *reinterpret_cast<uint2 *>(smem + Geo::Y_OFF + ((tid - 256) >> 5) * Geo::YPLANE + (192 + (tid & 31)) * 8) = make_uint2(0, 0);
        if (tid < 64) *reinterpret_cast<uint32_t *>(smem + Geo::S_OFF + (tid >> 5) * Geo::SPLANE + (192 + (tid & 31)) * 4) = 0x7f7f7f7fu;
    };
    if (tid < 16) *reinterpret_cast<uint4 *>(smem + Geo::ZERO_OFF + (tid << 4)) = make_uint4(0, 0, 0, 0);
    init_aliases(tid);
    if (head_out && tid < 3 * 128 / 4)
        reinterpret_cast<float4 *>(smem + Geo::HEADW_OFF)[tid] = reinterpret_cast<const float4 *>(head_w)[tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int rowk[3], tapmask[3], natb[3];   // rowk: the lane's LDS row in tile i (-1: padding row)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int k = 32 * (wr * 3 + i) + l31 - 12;
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
    auto tap_addr = [&](int khalf, int tap, int (&ab)[3], int (&key)[3], int (&xr)[3], int (&yr)[3]) {   // as k_trunk_mx_c128
        const int delta = (tap / 3 - 1) * 20 + (tap - (tap / 3) * 3 - 1);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int kk = rowk[i] + delta;
            const bool on = (tapmask[i] >> tap) & 1;
            const int rb = Geo::HI_OFF + kk * CV_ROWB;
            ab[i] = on ? rb : Geo::ZERO_OFF;
            key[i] = ((rb >> 8) & 15) ^ khalf;
            xr[i] = khalf * Geo::XPLANE + ((on ? kk : 192 + (kk & 31)) << 4);
            yr[i] = xr[i] >> 1;
        }
    };
    // weight operand addresses inside a slab, channel tile j = 0 (2 cg); tile j = 1 is 512 / 256 / 128 bytes further (immediates)
    const int vb0 = Geo::W_OFF + khalf * 2048 + ((cg * 64 + l31) << 4);
    const int vy0 = Geo::W_OFF + 12288 + khalf * 1024 + ((cg * 64 + l31) << 3);
    const int vs0 = Geo::W_OFF + 14336 + khalf * 512 + ((cg * 64 + l31) << 2);
    int keep;

    int rkk[3];
    auto refresh_rk = [&]() {   // opaque copies: keeps the epilogue addresses out of registers across the slab loop
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            rkk[i] = rowk[i] < 0 ? 0 : rowk[i];
            asm volatile("" : "+v"(rkk[i]));
        }
    };
    // epilogue of one tile (k_trunk_mx_c128's): ReLU + clamp, hi = rn16(v) to the HI row, the fp6 block [hi | 2^11 (v - hi)] under the
    // scale 2^(exponent(max v) - 2) to the X / Y planes of group 2 ct + khalf, the scale byte to SC; keep_x: v is a block input and
    // goes to the workgroup's scratch
    auto store_tile = [&](const Mx2Th &th, float *xw, f32x16 a, int i, bool keep_x) {
        const int ct = th.ct, khalf = th.khalf;
        f32x16 v;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = __builtin_amdgcn_fmed3f(a[r], 0.0f, 65504.0f);
        if (keep_x) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4 *>(xw + (4 * i + q) * 256) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        }
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
    auto store_tile_f32 = [&](const Mx2Th &th, f32x16 a, int i) {
        const int ct = th.ct, khalf = th.khalf;
        if (rowk[i] < 0) return;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 8 * ct + 2 * q + khalf;
            *reinterpret_cast<float4 *>(smem + rkk[i] * Geo::F32_ROWB + ((c ^ (rkk[i] & 31)) << 4)) =
                make_float4(fmaxf(a[4 * q + 0], 0.0f), fmaxf(a[4 * q + 1], 0.0f), fmaxf(a[4 * q + 2], 0.0f), fmaxf(a[4 * q + 3], 0.0f));
        }
    };
    auto bias_acc = [&](const Mx2Th &th, f32x16 (&dst)[3], const float *bl) {   // accumulator registers 4 q + r = channel 32 ct + 8 q + 4 khalf + r
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bq = *reinterpret_cast<const float4 *>(bl + th.ct * 32 + 8 * q + 4 * th.khalf);
#pragma unroll
            for (int i = 0; i < 3; ++i) { dst[i][4 * q + 0] = bq.x; dst[i][4 * q + 1] = bq.y; dst[i][4 * q + 2] = bq.z; dst[i][4 * q + 3] = bq.w; }
        }
    };

    {   // first layer: conv3x3(14 -> 128) + BN + ReLU of the wave's three FINAL tiles (no K split: 18 k-steps of 16 padded channels);
        // the planes are exact in 16 bits, the weights are hi + lo (two fp16 MFMAs)
        f32x16 acc[3];
        const Mx2Th th = Mx2Th::here();
        bias_acc(th, acc, b0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int shift = (t / 3 - 1) * 10 + (t % 3 - 1);
            bf16x8 af[3], wf[2];
#pragma unroll
            for (int part = 0; part < 2; ++part)
                wf[part] = *reinterpret_cast<const bf16x8 *>(w0 + ((size_t)(((t * 2 + part) * 2 + khalf) * 128 + ct * 32 + l31) << 3));
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int a = ((tapmask[i] >> t) & 1) ? Geo::PLANES_OFF + (natb[i] + shift * 32) : Geo::ZERO_OFF;
                af[i] = *reinterpret_cast<const bf16x8 *>(smem + a + khalf * 16);
            }
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    acc[i] = mfma_32x32x16<true>(wf[part], af[i], acc[i]);
        }
        refresh_rk();
#pragma unroll
        for (int i = 0; i < 3; ++i) store_tile(th, xw, acc[i], i, true);
        __syncthreads();
        dma_slab(3);   // the planes are done with ring buffer 3: the second slab of pair 1 (published by body 0's barrier; a layer's
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // first body does not wait for vector memory, so it has to have landed here)
    }

#define MX2_OPERANDS()                                                                                               \
            /* accumulators and operand sets in FIXED registers: with six 16-register tuples tied through six asm statements */ \
            /* and two loops the allocator otherwise re-homes them between statements (copies, 200+ bytes of scratch) */         \
            : [c00] "+{v[0:15]}"(acc[0][0]), [c01] "+{v[16:31]}"(acc[0][1]), [c10] "+{v[32:47]}"(acc[1][0]),                   \
              [c11] "+{v[48:63]}"(acc[1][1]), [c20] "+{v[64:79]}"(acc[2][0]), [c21] "+{v[80:95]}"(acc[2][1]),                  \
              [a0h0] "+{v[96:99]}"(fa.a[0]), [a0h1] "+{v[100:103]}"(fa.a[1]), [a0h2] "+{v[104:107]}"(fa.a[2]),                 \
              [w00] "+{v[108:111]}"(fa.w[0]), [w01] "+{v[112:115]}"(fa.w[1]),                                                  \
              [a1h0] "+{v[116:119]}"(fb.a[0]), [a1h1] "+{v[120:123]}"(fb.a[1]), [a1h2] "+{v[124:127]}"(fb.a[2]),               \
              [w10] "+{v[128:131]}"(fb.w[0]), [w11] "+{v[132:135]}"(fb.w[1]),                                                  \
              [sb0] "+v"(sb[0]), [sb1] "+v"(sb[1]), [sb2] "+v"(sb[2]),                                                \
              [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [ws0] "=&v"(ws0), [ws1] "=&v"(ws1), [keep] "=&s"(keep)
#define MX2_INPUTS(NAB, NKEY)                                                                                        \
              [nab0] "v"(NAB[0]), [nab1] "v"(NAB[1]), [nab2] "v"(NAB[2]), [nkey0] "v"(NKEY[0]),                        \
              [nkey1] "v"(NKEY[1]), [nkey2] "v"(NKEY[2]), [xr0] "v"(xr[0]), [xr1] "v"(xr[1]), [xr2] "v"(xr[2]),        \
              [yr0] "v"(yr[0]), [yr1] "v"(yr[1]), [yr2] "v"(yr[2]), [vb] "v"(vb), [vy] "v"(vy), [vs] "v"(vs), [vbn] "v"(vbn), \
              [voff0] "v"(voff0), [sbase] "s"(sbase), [ldst] "s"(ldst), [cba] "s"(cba), [cbb] "s"(cbb), [xo] "s"(xo),  \
              [yo] "s"(yo), [kp8] "s"(kp8)
#define MX2_CLOBBERS "memory", "scc", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237",   \
                     "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", \
                     "v252", "v253", "v254", "v255"
    // HS: this body's quarter (kp or kp + 2); HN: the next body's
#define MX2_ARGS(HS, HN)                                                                                             \
            const int s_ = 2 * g + KP;                               /* this body's slab */                           \
            const int slot = ((unsigned)s_ & 3u) << Geo::SLAB_SHIFT;                                                 \
            const int vb = vb0 + slot, vy = vy0 + slot, vs = vs0 + slot;                                             \
            const int vbn = vb0 + ((((unsigned)s_ + 2u) & 3u) << Geo::SLAB_SHIFT);                                   \
            const int pn = 2 * g + 4 < nslabs ? 2 * g + 4 : nslabs - 2;   /* first slab of the pair after next */     \
            const unsigned char *sbase = wpk + (size_t)pn * Geo::SLAB_BYTES;                                         \
            const int ldst = Geo::W_OFF + (((2u * (unsigned)g + 4u) & 3u) << Geo::SLAB_SHIFT) + (wave_u << 10);      \
            const int cba = 4 * (HN), cbb = 4 * (HN) + 2;                                                            \
            const int xo = (HS) * 2 * Geo::XPLANE, yo = Geo::Y_OFF + (HS) * 2 * Geo::YPLANE;
#define MX2_RUN(ASMSTR, HS, HN, NAB, NKEY)                                                                           \
        {                                                                                                            \
            MX2_ARGS(HS, HN)                                                                                         \
            asm volatile(ASMSTR MX2_OPERANDS() : MX2_INPUTS(NAB, NKEY) : MX2_CLOBBERS);                              \
            ++g;                                                                                                     \
        }
#define MX2_RUNV(ASMSTR, HS, HN, NAB, NKEY)   /* the same + the wave-uniform skip mask; clobbers VCC */               \
        {                                                                                                            \
            MX2_ARGS(HS, HN)                                                                                         \
            asm volatile(ASMSTR MX2_OPERANDS() : MX2_INPUTS(NAB, NKEY), [skipm] "s"(skipm) : MX2_CLOBBERS, "vcc");   \
            ++g;                                                                                                     \
        }

    struct Mx2Frag { bf16x8 a[3], w[2]; };
    int skipm;     // cell group 0 (waves 0..3): its first row tile skips the dy = -1 taps
    asm volatile("s_cmp_lt_u32 %1, 4\n\ts_cselect_b32 %0, 1, 0" : "=s"(skipm) : "s"(wave_u) : "scc");
    // the tower, instantiated for both values of kp (a wave-uniform branch around the whole loop: nothing is live where the paths
    // join): channel tile j of the wave is 2 cg + j, the one it finalises j = KP
#ifdef MX2_TIMING   /* diagnostic build (tools/mx_timing.py): where a workgroup's cycles go; the clock-probe buffer carries the sums */
    unsigned long long tm_loop = 0, tm_xch = 0, tm_epi = 0, tm_t0 = 0, tm_x1 = 0, tm_x2 = 0, tm_x3 = 0, tm_x4 = 0;
#define MX2_TM(acc) { const unsigned long long n_ = __builtin_readcyclecounter(); acc += n_ - tm_t0; tm_t0 = n_; }
#else
#define MX2_TM(acc)
#endif
    auto tower = [&](auto kpc) {
    constexpr int KP = decltype(kpc)::value;
    constexpr int kp8 = KP * 8;
    int g = 0;     // pair (= body) counter over the whole tower
#pragma unroll 1
    for (int layer = 0; layer < nlayers; ++layer) {
        f32x16 acc[3][2];   // both start at zero: the layer's bias joins in the epilogue, with the partner's partial sums
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        int ab[3], key[3], nab[3], nkey[3], xr[3], yr[3], nxr[3], nyr[3], t0, t1, t2, t3, ws0, ws1;
        // per layer: what the addresses are made of is made opaque, or the compiler computes the (layer-invariant) addresses of
        // every tap once, before the layer loop, and parks them in scratch
        asm volatile("" : "+v"(rowk[0]), "+v"(rowk[1]), "+v"(rowk[2]), "+v"(tapmask[0]), "+v"(tapmask[1]), "+v"(tapmask[2]));
        const int kh0 = Mx2Th::here().khalf;
        int sb[3] = {0, 0, 0};
        Mx2Frag fa, fb;
        tap_addr(kh0, 0, ab, key, xr, yr);
        {   // the first body's two fp16 operand sets (quarter kp of tap 0)
            const int vb = vb0 + (((unsigned)(2 * g + KP) & 3u) << Geo::SLAB_SHIFT);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                fa.a[i] = *reinterpret_cast<const bf16x8 *>(smem + ab[i] + (((4 * KP + 0) ^ key[i]) << 4));
                fb.a[i] = *reinterpret_cast<const bf16x8 *>(smem + ab[i] + (((4 * KP + 2) ^ key[i]) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fa.w[j] = *reinterpret_cast<const bf16x8 *>(smem + vb + j * 512);
                fb.w[j] = *reinterpret_cast<const bf16x8 *>(smem + vb + j * 512 + 4096);
            }
            asm volatile("" : "+v"(fa.a[0]), "+v"(fa.a[1]), "+v"(fa.a[2]), "+v"(fa.w[0]), "+v"(fa.w[1]),
                              "+v"(fb.a[0]), "+v"(fb.a[1]), "+v"(fb.a[2]), "+v"(fb.w[0]), "+v"(fb.w[1]));
        }
        MX2_TM(tm_epi)
        // tap 0, peeled: the layer's first body does not wait for vector memory (see slabMX2)
        MX2_RUNV(MX2_SKIP0_A_FIRST, KP, KP + 2, ab, key)
        tap_addr(kh0, 1, nab, nkey, nxr, nyr);
        MX2_RUNV(MX2_SKIP0_B, KP + 2, KP, nab, nkey)
#pragma unroll
        for (int i = 0; i < 3; ++i) { ab[i] = nab[i]; key[i] = nkey[i]; xr[i] = nxr[i]; yr[i] = nyr[i]; }
        int tap = 1;
#pragma unroll 1
        for (; tap < 3; ++tap) {   // dy = -1: cell group 0 branches around the MFMAs of its all-rank-0 row tile
            MX2_RUNV(MX2_SKIP0_A, KP, KP + 2, ab, key)
            tap_addr(kh0, tap + 1, nab, nkey, nxr, nyr);
            MX2_RUNV(MX2_SKIP0_B, KP + 2, KP, nab, nkey)
#pragma unroll
            for (int i = 0; i < 3; ++i) { ab[i] = nab[i]; key[i] = nkey[i]; xr[i] = nxr[i]; yr[i] = nyr[i]; }
        }
#pragma unroll 1
        for (; tap < 8; ++tap) {
            MX2_RUN(MX2_BODY_A, KP, KP + 2, ab, key)
            tap_addr(kh0, tap + 1, nab, nkey, nxr, nyr);
            MX2_RUN(MX2_BODY_B, KP + 2, KP, nab, nkey)
#pragma unroll
            for (int i = 0; i < 3; ++i) { ab[i] = nab[i]; key[i] = nkey[i]; xr[i] = nxr[i]; yr[i] = nyr[i]; }
        }
        // tap 8: the layer's last body requests nothing
        MX2_RUN(MX2_BODY_A, KP, KP + 2, ab, key)
        MX2_RUN(MX2_BODY_B_LAST, KP + 2, KP, ab, key)
        // let the MFMAs retire (the compiler does not see them), the DMAs land (the last layer's fp32 rows reach into the weight
        // ring).  Behind the last body's barrier (which waited for every LDS read of its wave) nobody reads the activation planes any
        // more: they are the exchange area from here on, without another barrier
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        MX2_TM(tm_loop)
        const bool last = layer + 1 == nlayers;
        const bool add_x = (layer & 1) != 0;
        const Mx2Th th = Mx2Th::here();      // everything from here to the end of the layer is addressed from this copy
        float *xw = xbuf + (size_t)blockIdx.x * MX2_XBUF_FLOATS_PER_WG + (size_t)th.wave * (48 * 64) + th.lane * 4;
        float4 bq[4];                        // the layer's bias at this lane's accumulator positions (channel 32 ct + 8 q + 4 khalf + r)
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4 *>(bias + layer * 128 + th.ct * 32 + 8 * q + 4 * th.khalf);
        auto load_x = [&](f32x16 &dst, int i) {   // the block input, written two layers ago: in flight during the exchange
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4 *>(xw + (4 * i + q) * 256);
                dst[4 * q] = t.x; dst[4 * q + 1] = t.y; dst[4 * q + 2] = t.z; dst[4 * q + 3] = t.w;
            }
        };
        // exchange: the partial sums of the tile the partner finalises (j = 1 - KP) go to it (wave ^ 1), its partial sums of MY tile
        // come back.  4 KB per wave and cell tile: tiles 0, 1 at once (64 KB), then tile 2
        unsigned char *exw = smem + th.wave * 4096 + th.lane * 16, *exr = smem + (th.wave ^ 1) * 4096 + th.lane * 16;
        f32x16 xin[3];
#ifndef MX2_ABLATE_NO_EXCHANGE   /* timing ablation (tools/experiments/mx_ablate.sh; wrong results) */
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4 *>(exw + i * 32768 + q * 1024) = make_float4(acc[i][1 - KP][4 * q], acc[i][1 - KP][4 * q + 1], acc[i][1 - KP][4 * q + 2], acc[i][1 - KP][4 * q + 3]);
        if (add_x) { load_x(xin[0], 0); load_x(xin[1], 1); }    // into the registers the two tiles just left
        MX2_TM(tm_x1)
        __syncthreads();
        MX2_TM(tm_x2)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = *reinterpret_cast<const float4 *>(exr + i * 32768 + q * 1024);
                acc[i][KP][4 * q] += t.x; acc[i][KP][4 * q + 1] += t.y; acc[i][KP][4 * q + 2] += t.z; acc[i][KP][4 * q + 3] += t.w;
            }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(exw + q * 1024) = make_float4(acc[2][1 - KP][4 * q], acc[2][1 - KP][4 * q + 1], acc[2][1 - KP][4 * q + 2], acc[2][1 - KP][4 * q + 3]);
        MX2_TM(tm_x3)
        if (add_x) load_x(xin[2], 2);
        __syncthreads();
        MX2_TM(tm_x4)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 t = *reinterpret_cast<const float4 *>(exr + q * 1024);
            acc[2][KP][4 * q] += t.x; acc[2][KP][4 * q + 1] += t.y; acc[2][KP][4 * q + 2] += t.z; acc[2][KP][4 * q + 3] += t.w;
        }
#else
        (void)exw; (void)exr;
        if (add_x) { load_x(xin[0], 0); load_x(xin[1], 1); load_x(xin[2], 2); }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][KP][r] += acc[i][1 - KP][r];
#endif
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[i][KP][4 * q] += bq[q].x; acc[i][KP][4 * q + 1] += bq[q].y; acc[i][KP][4 * q + 2] += bq[q].z; acc[i][KP][4 * q + 3] += bq[q].w;
            }
        if (add_x) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][KP][r] += xin[i][r];
        }
        __syncthreads();
        MX2_TM(tm_xch)
        refresh_rk();
        if (last) {
#pragma unroll
            for (int i = 0; i < 3; ++i) store_tile_f32(th, acc[i][KP], i);
        } else {
            init_aliases(th.tid);
#pragma unroll
            for (int i = 0; i < 3; ++i) store_tile(th, xw, acc[i][KP], i, add_x);
        }
        __syncthreads();
    }
    };   // tower
#ifdef MX2_TIMING
    tm_t0 = __builtin_readcyclecounter();
    const unsigned long long tm_first = tm_t0 - clk_c0;
#endif
    if (kp_u) tower(std::integral_constant<int, 1>{});
    else tower(std::integral_constant<int, 0>{});
    if (clk && tid == 0) {
        clk[blockIdx.x * 4 + 0] = clk_c0; clk[blockIdx.x * 4 + 1] = __builtin_readcyclecounter();
        clk[blockIdx.x * 4 + 2] = clk_r0; clk[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
#ifdef MX2_TIMING
        clk[blockIdx.x * 4 + 0] = tm_first; clk[blockIdx.x * 4 + 1] = tm_loop; clk[blockIdx.x * 4 + 2] = tm_xch; clk[blockIdx.x * 4 + 3] = tm_epi;
        unsigned long long *c2 = clk + (size_t)(gridDim.x + blockIdx.x) * 4;     // tools/mx_timing.py arms a buffer of twice the grid
        c2[0] = tm_x1; c2[1] = tm_x2; c2[2] = tm_x3; c2[3] = tm_x4;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (out) {   // trunk activations as fp32, 4 channels per thread and step
        float4 *go = reinterpret_cast<float4 *>(out + (size_t)pos0 * 90 * 128);
        for (int idx = tid; idx < nrows * 32; idx += Geo::THREADS) {
            const int r = idx >> 5, c = idx & 31, k = lds_row_of(r);
            go[idx] = *reinterpret_cast<const float4 *>(smem + k * Geo::F32_ROWB + ((c ^ (k & 31)) << 4));
        }
    }
    if (head_out) {
        const float *hw = reinterpret_cast<const float *>(smem + Geo::HEADW_OFF);
        // one thread per board cell, all three head channels; chunks in a fixed order: a position's outputs do not depend on
        // the row / workgroup it lands on
        for (int r = tid; r < nrows; r += Geo::THREADS) {
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
#undef MX2_OPERANDS
#undef MX2_INPUTS
#undef MX2_CLOBBERS
#undef MX2_ARGS
#undef MX2_RUN
#undef MX2_RUNV

}  // namespace czconv
