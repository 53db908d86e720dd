// cz_trunk_split.h — N1s: the strict-precision net trunk.  k_tower8_c128 (cz_conv_kernel.h) rounds every weight and every
// stored activation to ONE 16-bit float: 11 (fp16) / 8 (bf16) significant bits per operand, |dlogit| 1.5e-2 against the
// reference's fp32 sess.run (policy_value_network.py:202-214) on trained-like weights at 7 blocks, 3.5e-2 at 19 — outside
// north_star's 1e-3.  The decomposition (tools/precision_decomposition.py) says weights and activations contribute about
// equally and that keeping only one of them exact buys nothing, so here BOTH are carried as an unevaluated sum of two
// 16-bit floats, x = hi + lo with hi = rn16(x), lo = rn16(x - hi) (22 / 16 significant bits), and a product is three MFMAs
//     a*w  ~=  a_hi*w_hi + a_hi*w_lo + a_lo*w_hi          (a_lo*w_lo < 2^-22 |a*w|: below the fp32 accumulator's resolution)
// accumulated in fp32 in the same accumulator.  Measured against the fp64 graph: fp16 halves 1.8e-5 (7 blocks) / 1.0e-4
// (19 blocks) on trained-like weights, bf16 halves 1.8e-4 / 4.8e-4.
//
// Same one-launch structure as k_tower8_c128 (first conv + all residual blocks + head 1x1 convs, activations resident in
// LDS, weights streamed L2 -> LDS by LDS-DMA through a ring of four 16 KB slabs, hand-scheduled slab body from
// tools/gen_tower_asm.py), re-tiled for twice the bytes per value:
//   positions  2 per workgroup (180 cells = 6 tiles of 32, 12 dead rows), one workgroup per CU; activations 2 x 45 KB:
//              the hi halves at LDS rows 0..179, one zero row, the lo halves LO_OFF = 181 rows behind (the lo half of a
//              cell is read with the ds_read's immediate offset from the hi half's address: one address per cell and tap;
//              its chunk swizzle is its hi row's), the lo zero row behind them so that a masked tap reads zeros in both.
//   waves      8 = 2 (cell groups of 3 tiles) x 4 (channel tiles of 32); 3 accumulators each; per k-step 8 ds_read_b128
//              (3 x hi/lo activation fragments + the weight tile's hi/lo) feed 9 MFMAs.
//   weights    slab = 32 input channels of one tap: [hi: 4 x 128 co x 8][lo: the same] = 16 KB, 36 slabs per layer,
//              2 k-steps per slab, one barrier per slab, 2 DMA pieces per wave and slab (vmcnt(2)).
//   epilogue   acc -> clamp to [0, 65504] (ReLU; fp16 cannot overflow to inf) -> hi = rn16(v), lo = rn16(v - hi) -> LDS in
//              place; the block input x stays in registers as (hi, lo) pairs and is folded into the accumulator
//              initialisation of the block's second conv.
//   first conv the input planes are 0/1: exact in 16 bits, so only the weights are split (2 MFMAs per tap and tile).
//   heads      the two 1x1 convs read hi + lo in fp32.
// Roofline: MFMA; algorithmic flops are those of the fp32 graph (the 3x is the price of the precision, not useful work).
#pragma once
#include "cz_conv_kernel.h"

namespace czconv {

#include "cz_trunk_split_asm.inc"

struct XSGeo {
    static constexpr int P = 2;
    static constexpr int ROWS = P * 90;                        // 180 cells -> 6 row tiles of 32
    static constexpr int THREADS = 512;                        // 8 waves: 2 cell groups x 4 channel tiles
    static constexpr int ZERO_OFF = ROWS * CV_ROWB;            // 46,080: the zero row of the hi half
    static constexpr int LO_OFF = ZERO_OFF + CV_ROWB;          // 46,336: lo half of row r at LO_OFF + r * 256; its zero row at ZERO_OFF + LO_OFF
    static constexpr int W_OFF = 2 * LO_OFF;                   // 92,672
    static constexpr int SLAB_BYTES = 2 * 32 * 128 * 2;        // 32 input channels of one tap, hi then lo: 16 KB
    static constexpr int SLAB_SHIFT = 14;
    static constexpr int SLABS_PER_LAYER = 36;
    static constexpr int LDS_BYTES = W_OFF + TW_NBUF * SLAB_BYTES;            // 158,208
    static constexpr int PLANES_OFF = W_OFF + 3 * SLAB_BYTES;  // the input planes (32 B per cell) borrow ring buffer 3
    static constexpr int HEADW_OFF = LDS_BYTES;                // head 1x1 conv weights [3][128] f32
    static constexpr int LDS_TOTAL = LDS_BYTES + 3 * 128 * 4;  // 159,744 of the CU's 163,840
};
static_assert(XSGeo::LO_OFF == 181 * 256, "tools/gen_tower_asm.py: XS_LO_OFF");
constexpr int XS_P = XSGeo::P, XS_THREADS = XSGeo::THREADS, XS_LDS_BYTES = XSGeo::LDS_TOTAL;

struct XsFrag { bf16x8 ah[3], al[3], wh, wl; };

// the eight fragment reads of one k-step, no MFMAs (layer prologue)
#define XS_LOADSET(CA, OB, OBL, X, AB, KEY, VB)                                                          \
    asm volatile(                                                                                    \
        "v_xor_b32 %[t0], " #CA ", %[k0]\n\t"                                                        \
        "v_xor_b32 %[t1], " #CA ", %[k1]\n\t"                                                        \
        "v_xor_b32 %[t2], " #CA ", %[k2]\n\t"                                                        \
        "v_lshl_add_u32 %[t0], %[t0], 4, %[b0]\n\t"                                                  \
        "v_lshl_add_u32 %[t1], %[t1], 4, %[b1]\n\t"                                                  \
        "v_lshl_add_u32 %[t2], %[t2], 4, %[b2]\n\t"                                                  \
        "ds_read_b128 %[xh0], %[t0]\n\t"                                                             \
        "ds_read_b128 %[xl0], %[t0] offset:46336\n\t"                                                \
        "ds_read_b128 %[xh1], %[t1]\n\t"                                                             \
        "ds_read_b128 %[xl1], %[t1] offset:46336\n\t"                                                \
        "ds_read_b128 %[xh2], %[t2]\n\t"                                                             \
        "ds_read_b128 %[xl2], %[t2] offset:46336\n\t"                                                \
        "ds_read_b128 %[xwh], %[vb] offset:" #OB "\n\t"                                              \
        "ds_read_b128 %[xwl], %[vb] offset:" #OBL "\n\t"                                        \
        : [xh0] "=&v"(X.ah[0]), [xh1] "=&v"(X.ah[1]), [xh2] "=&v"(X.ah[2]), [xl0] "=&v"(X.al[0]),     \
          [xl1] "=&v"(X.al[1]), [xl2] "=&v"(X.al[2]), [xwh] "=&v"(X.wh), [xwl] "=&v"(X.wl),           \
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)                                              \
        : [k0] "v"(KEY[0]), [k1] "v"(KEY[1]), [k2] "v"(KEY[2]), [b0] "v"(AB[0]), [b1] "v"(AB[1]),      \
          [b2] "v"(AB[2]), [vb] "v"(VB)                                                               \
        : "memory")

// x = hi + lo: the two 16-bit halves of four fp32 values (channels n0 .. n0+3 of one cell), packed like the LDS cell
struct XsQuad { uint32_t h01, h23, l01, l23; };

template <bool F16> __device__ __forceinline__ XsQuad xs_split4(float v0, float v1, float v2, float v3) {
    // ReLU (and, for fp16, a clamp at the largest finite half: an activation beyond 65504 would otherwise become inf in hi
    // and NaN in lo) BEFORE the split: a negative value has hi = lo = 0
    const float top = F16 ? 65504.0f : 3.3895314e38f;
    v0 = __builtin_amdgcn_fmed3f(v0, 0.0f, top); v1 = __builtin_amdgcn_fmed3f(v1, 0.0f, top);
    v2 = __builtin_amdgcn_fmed3f(v2, 0.0f, top); v3 = __builtin_amdgcn_fmed3f(v3, 0.0f, top);
    XsQuad q;
    q.h01 = pack_pair<F16>(f32x2{v0, v1});
    q.h23 = pack_pair<F16>(f32x2{v2, v3});
    const f32x2 a = unpack_pair<F16>(q.h01), b = unpack_pair<F16>(q.h23);
    q.l01 = pack_pair<F16>(f32x2{v0 - a[0], v1 - a[1]});   // v - hi is exact in fp32
    q.l23 = pack_pair<F16>(f32x2{v2 - b[0], v3 - b[1]});
    return q;
}

// planes [B][90][16] (0/1 in the operand type), w0 [9 taps][hi, lo][2 = ci/8][128 co][8], wpk [L][36 slabs][hi, lo][4 = ci/8][128 co][8],
// bias / b0 fp32; out: trunk [B][90][128] FP32 (hi + lo) or NULL; head_out [B][90][3] fp32 (post-ReLU head conv outputs) or NULL.
template <bool F16>
__global__ __launch_bounds__(512, 2) void k_trunk_split_c128(const uint16_t *__restrict__ wpk,
                                                              const float *__restrict__ bias,
                                                              float *__restrict__ out,
                                                              const float *__restrict__ head_w,
                                                              const float *__restrict__ head_b,
                                                              float *__restrict__ head_out,
                                                              const uint16_t *__restrict__ planes,
                                                              const uint16_t *__restrict__ w0,
                                                              const float *__restrict__ b0,
                                                              int B, int nlayers,
                                                              const int *__restrict__ bcount,    // device row count or NULL
                                                              unsigned long long *__restrict__ clk) {   // clock probe [grid][4] or NULL (cz_set_clock_probe)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Geo = XSGeo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, ct = wave & 3;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int pos0 = blockIdx.x * Geo::P;
    if (bcount) {
        const int live = *bcount;
        B = live < B ? live : B;
    }
    if (pos0 >= B) return;
    // measurement hook (bench.py roofline.effective_clock_GHz): workgroup lifetime in shader-clock cycles (s_memtime) and in
    // the constant 100 MHz reference clock (s_memrealtime); four scalar registers, one uniform branch when the probe is off
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (clk) { clk_c0 = __builtin_readcyclecounter(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    const int npos = (B - pos0) < Geo::P ? (B - pos0) : Geo::P;
    const int nrows = npos * 90;
    const int nslabs = nlayers * Geo::SLABS_PER_LAYER;
    auto lds_addr = [](int row_byte_off, int c) { return row_byte_off + ((c ^ ((row_byte_off >> 8) & 15)) << 4); };
    const unsigned voff0 = (unsigned)tid << 4, voff1 = voff0 + (unsigned)Geo::THREADS * 16u;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // Cell order inside the workgroup (as in k_tower8_c128): LDS row k of (position p, cell y * 10 + x) is 20 y + 10 p + x, and the
    // 192 GEMM rows are the 12 padding rows FIRST, then k = 0 .. 179: the first row tile holds nothing but padding and the rank-0
    // cells of both positions, whose dy = -1 taps are off the board — cell group 0 (waves 0..3, one per SIMD) does not issue
    // that tile's three MFMAs per k-step in taps 0..2 (XS_SKIP0 bodies below)
    auto lds_row_of = [](int natural) {   // natural = p * 90 + y * 10 + x
        const int p = natural / 90, c = natural - p * 90, y = c / 10, x = c - y * 10;
        return 20 * y + 10 * p + x;
    };

    auto dma_slab = [&](int slab) {   // prologue only; the loop issues its DMAs from the slab asm
        const unsigned char *src = reinterpret_cast<const unsigned char *>(wpk) + (size_t)slab * Geo::SLAB_BYTES;
        unsigned char *dst = smem + Geo::W_OFF + ((unsigned)slab & 3u) * Geo::SLAB_BYTES + (wave_u << 10);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + voff0),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + voff1),
                                         (__attribute__((address_space(3))) void *)(dst + Geo::THREADS * 16), 16, 0, 0);
    };
    for (int q = 0; q < 3; ++q) dma_slab(q < nslabs ? q : nslabs - 1);
    {
        const uint4 *g = reinterpret_cast<const uint4 *>(planes + (size_t)pos0 * 90 * 16);
        for (int idx = tid; idx < Geo::ROWS * 2; idx += Geo::THREADS) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < nrows * 2) v = g[idx];
            *reinterpret_cast<uint4 *>(smem + Geo::PLANES_OFF + (idx << 4)) = v;
        }
    }
    if (tid < 16) {
        *reinterpret_cast<uint4 *>(smem + Geo::ZERO_OFF + (tid << 4)) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4 *>(smem + Geo::ZERO_OFF + Geo::LO_OFF + (tid << 4)) = make_uint4(0, 0, 0, 0);
    }
    if (head_out && tid < 3 * 128 / 4)
        reinterpret_cast<float4 *>(smem + Geo::HEADW_OFF)[tid] = reinterpret_cast<const float4 *>(head_w)[tid];
    // first-layer weights (hi, lo) of this wave's channel tile: requested before the wait below
    bf16x8 wf[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int part = 0; part < 2; ++part)
            wf[t][part] = *reinterpret_cast<const bf16x8 *>(w0 + ((size_t)(((t * 2 + part) * 2 + khalf) * 128 + ct * 32 + l31) << 3));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int rowb[3], tapmask[3], natb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int k = 32 * (wr * 3 + i) + l31 - 12;       // LDS row; k < 0: one of the 12 padding rows
        const int kk = k < 0 ? 0 : k;
        const int h = kk / 20, rem = kk - h * 20, pp = rem / 10, w = rem - pp * 10;
        rowb[i] = k * CV_ROWB;   // a padding lane (k < 0, every tap masked) keeps its own virtual row for the zero row's slot: with row 0 for all
                                 // twelve their slot collided with the real row lane 12 reads (10.6 % of the LDS cycles in bank conflicts)
        natb[i] = (pp * 90 + h * 10 + w) * 32;           // the cell's 32 bytes of input planes (natural order)
        int m = 0;
        for (int t = 0; t < 9; ++t) {
            const int y = h + t / 3 - 1, x = w + t % 3 - 1;
            if (k >= 0 && y >= 0 && y < 9 && x >= 0 && x < 10) m |= 1 << t;
        }
        tapmask[i] = m;
    }
    auto tap_addr = [&](int tap, int (&ab)[3], int (&key)[3]) {
        const int delta = ((tap / 3 - 1) * 20 + (tap - (tap / 3) * 3 - 1)) * CV_ROWB;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ab[i] = ((tapmask[i] >> tap) & 1) ? rowb[i] + delta : Geo::ZERO_OFF;
            key[i] = (((rowb[i] + delta) >> 8) & 15) ^ khalf;   // a masked lane keeps its real neighbour's slot (cz_conv_kernel.h)
        }
    };
    const int vb0 = Geo::W_OFF + khalf * 2048 + ((ct * 32 + l31) << 4);
    int keep;

    int rb[3];
    auto refresh_rb = [&]() {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = 32 * (wr * 3 + i) + l31 - 12;
            rb[i] = (k < 0 ? 0 : k) * CV_ROWB;
            asm volatile("" : "+v"(rb[i]));
        }
    };
    // hi half of (cell of tile i, channels n0 .. n0+3); the lo half is LO_OFF bytes behind
    auto cell_ptr = [&](int i, int q, bool &live) -> unsigned char * {
        live = 32 * (wr * 3 + i) + l31 >= 12;
        const int n0 = ct * 32 + 8 * q + 4 * khalf;
        return smem + lds_addr(rb[i], n0 >> 3) + ((n0 & 4) << 1);
    };
    XsQuad xreg[3][4];   // block input x at this lane's accumulator positions
    auto init_acc = [&](f32x16 (&acc)[3], const float *bl, bool add_x) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bq = *reinterpret_cast<const float4 *>(bl + ct * 32 + 8 * q + 4 * khalf);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                float a0 = bq.x, a1 = bq.y, a2 = bq.z, a3 = bq.w;
                if (add_x) {
                    const XsQuad x = xreg[i][q];
                    const f32x2 h0 = unpack_pair<F16>(x.h01), h1 = unpack_pair<F16>(x.h23);
                    const f32x2 l0 = unpack_pair<F16>(x.l01), l1 = unpack_pair<F16>(x.l23);
                    a0 += h0[0] + l0[0]; a1 += h0[1] + l0[1]; a2 += h1[0] + l1[0]; a3 += h1[1] + l1[1];
                }
                acc[i][4 * q + 0] = a0; acc[i][4 * q + 1] = a1; acc[i][4 * q + 2] = a2; acc[i][4 * q + 3] = a3;
            }
        }
    };
    auto store_layer = [&](f32x16 (&acc)[3]) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bool live;
                unsigned char *cell = cell_ptr(i, q, live);
                const XsQuad s = xs_split4<F16>(acc[i][4 * q + 0], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
                if (live) {
                    *reinterpret_cast<uint2 *>(cell) = make_uint2(s.h01, s.h23);
                    *reinterpret_cast<uint2 *>(cell + Geo::LO_OFF) = make_uint2(s.l01, s.l23);
                }
            }
    };

    {   // first layer: conv3x3(14 -> 128) + BN + ReLU, one k-step per tap; the planes are exact, only the weights are split
        f32x16 acc[3];
        init_acc(acc, b0, false);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int shift = (t / 3 - 1) * 10 + (t % 3 - 1);
            bf16x8 af[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int a = ((tapmask[i] >> t) & 1) ? Geo::PLANES_OFF + (natb[i] + shift * 32) : Geo::ZERO_OFF;
                af[i] = *reinterpret_cast<const bf16x8 *>(smem + a + khalf * 16);
            }
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    acc[i] = mfma_32x32x16<F16>(wf[t][part], af[i], acc[i]);
        }
        refresh_rb();
        store_layer(acc);    // U is not read by the first conv: no barrier needed in front
        __syncthreads();
    }

#define XS_SLAB(ASMSTR, NAB, NKEY)                                                                               \
        asm volatile(ASMSTR                                                                                      \
            : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]),                                             \
              [f0ah0] "+v"(f0.ah[0]), [f0ah1] "+v"(f0.ah[1]), [f0ah2] "+v"(f0.ah[2]),                             \
              [f0al0] "+v"(f0.al[0]), [f0al1] "+v"(f0.al[1]), [f0al2] "+v"(f0.al[2]),                             \
              [f0wh] "+v"(f0.wh), [f0wl] "+v"(f0.wl),                                                            \
              [f1ah0] "=&v"(f1.ah[0]), [f1ah1] "=&v"(f1.ah[1]), [f1ah2] "=&v"(f1.ah[2]),                          \
              [f1al0] "=&v"(f1.al[0]), [f1al1] "=&v"(f1.al[1]), [f1al2] "=&v"(f1.al[2]),                          \
              [f1wh] "=&v"(f1.wh), [f1wl] "=&v"(f1.wl),                                                          \
              [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [keep] "=&s"(keep)                                     \
            : [ab0] "v"(ab[0]), [ab1] "v"(ab[1]), [ab2] "v"(ab[2]), [key0] "v"(key[0]), [key1] "v"(key[1]),          \
              [key2] "v"(key[2]), [nab0] "v"(NAB[0]), [nab1] "v"(NAB[1]), [nab2] "v"(NAB[2]), [nkey0] "v"(NKEY[0]),   \
              [nkey1] "v"(NKEY[1]), [nkey2] "v"(NKEY[2]), [vb] "v"(vb), [vbn] "v"(vbn), [voff0] "v"(voff0),          \
              [voff1] "v"(voff1), [sbase] "s"(sbase), [ldst] "s"(ldst)                                                \
            : "memory", "scc")
#define XS_SLABV(ASMSTR, NAB, NKEY)   /* the same operands + the wave-uniform skip mask; clobbers VCC */                 \
        asm volatile(ASMSTR                                                                                      \
            : [c0] "+v"(acc[0]), [c1] "+v"(acc[1]), [c2] "+v"(acc[2]),                                             \
              [f0ah0] "+v"(f0.ah[0]), [f0ah1] "+v"(f0.ah[1]), [f0ah2] "+v"(f0.ah[2]),                             \
              [f0al0] "+v"(f0.al[0]), [f0al1] "+v"(f0.al[1]), [f0al2] "+v"(f0.al[2]),                             \
              [f0wh] "+v"(f0.wh), [f0wl] "+v"(f0.wl),                                                            \
              [f1ah0] "=&v"(f1.ah[0]), [f1ah1] "=&v"(f1.ah[1]), [f1ah2] "=&v"(f1.ah[2]),                          \
              [f1al0] "=&v"(f1.al[0]), [f1al1] "=&v"(f1.al[1]), [f1al2] "=&v"(f1.al[2]),                          \
              [f1wh] "=&v"(f1.wh), [f1wl] "=&v"(f1.wl),                                                          \
              [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [keep] "=&s"(keep)                                     \
            : [ab0] "v"(ab[0]), [ab1] "v"(ab[1]), [ab2] "v"(ab[2]), [key0] "v"(key[0]), [key1] "v"(key[1]),          \
              [key2] "v"(key[2]), [nab0] "v"(NAB[0]), [nab1] "v"(NAB[1]), [nab2] "v"(NAB[2]), [nkey0] "v"(NKEY[0]),   \
              [nkey1] "v"(NKEY[1]), [nkey2] "v"(NKEY[2]), [vb] "v"(vb), [vbn] "v"(vbn), [voff0] "v"(voff0),          \
              [voff1] "v"(voff1), [sbase] "s"(sbase), [ldst] "s"(ldst), [skipm] "s"(skipm)                            \
            : "memory", "vcc", "scc")
#define XS_ARGS()                                                                                               \
            const int vb = vb0 + (((unsigned)g & 3u) << Geo::SLAB_SHIFT), vbn = vb0 + ((((unsigned)g + 1u) & 3u) << Geo::SLAB_SHIFT); \
            const int gn = g + 3 < nslabs ? g + 3 : nslabs - 1;                                                 \
            const unsigned char *sbase = reinterpret_cast<const unsigned char *>(wpk) + (size_t)gn * Geo::SLAB_BYTES; \
            const int ldst = Geo::W_OFF + ((((unsigned)g + 3u) & 3u) << Geo::SLAB_SHIFT) + (wave_u << 10);
#define XS_RUNV(BF, HF, NAB, NKEY)                                                                              \
        {                                                                                                       \
            XS_ARGS()                                                                                           \
            if constexpr (F16) { XS_SLABV(HF, NAB, NKEY); } else { XS_SLABV(BF, NAB, NKEY); }                   \
            ++g;                                                                                                \
        }
#define XS_RUN(BF, HF, NAB, NKEY)                                                                               \
        {                                                                                                       \
            const int vb = vb0 + (((unsigned)g & 3u) << Geo::SLAB_SHIFT), vbn = vb0 + ((((unsigned)g + 1u) & 3u) << Geo::SLAB_SHIFT); \
            const int gn = g + 3 < nslabs ? g + 3 : nslabs - 1;                                                 \
            const unsigned char *sbase = reinterpret_cast<const unsigned char *>(wpk) + (size_t)gn * Geo::SLAB_BYTES; \
            const int ldst = Geo::W_OFF + ((((unsigned)g + 3u) & 3u) << Geo::SLAB_SHIFT) + (wave_u << 10);      \
            if constexpr (F16) { XS_SLAB(HF, NAB, NKEY); } else { XS_SLAB(BF, NAB, NKEY); }                     \
            ++g;                                                                                                \
        }

    int g = 0;
    int skipm;   // cell group 0 (waves 0..3): its first row tile skips the dy = -1 taps.  Defined by scalar asm so that it IS a scalar register
    asm volatile("s_cmp_lt_u32 %1, 4\n\ts_cselect_b32 %0, 1, 0" : "=s"(skipm) : "s"(wave_u) : "scc");
#pragma unroll 1
    for (int layer = 0; layer < nlayers; ++layer) {
        f32x16 acc[3];
        refresh_rb();
        if (!(layer & 1)) {   // first conv of a block: remember x, start from the bias
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bool live;
                    const unsigned char *cell = cell_ptr(i, q, live);
                    const uint2 h = *reinterpret_cast<const uint2 *>(cell), l = *reinterpret_cast<const uint2 *>(cell + Geo::LO_OFF);
                    xreg[i][q] = XsQuad{h.x, h.y, l.x, l.y};
                }
            init_acc(acc, bias + layer * 128, false);
        } else {
            init_acc(acc, bias + layer * 128, true);
        }
        int ab[3], key[3], nab[3], nkey[3], t0, t1, t2;
        XsFrag f0, f1;
        tap_addr(0, ab, key);
        {
            const int vb = vb0 + (((unsigned)g & 3u) << Geo::SLAB_SHIFT);
            XS_LOADSET(0, 0, 8192, f0, ab, key, vb);   // waited for by the first k-step itself
        }
        // taps 0..2 (dy = -1): one wave of every SIMD (cell group 0) branches around the MFMAs of its all-rank-0 row tile
        // (1/6 of the SIMD's MFMAs in these twelve slabs, 5.6 % of a layer's; adding 0 * w is exact: outputs bit-identical)
        int tap = 0;
#pragma unroll 1
        for (; tap < 3; ++tap) {
            XS_RUNV(XS_SKIP0_ASM_Q0, XSF_SKIP0_ASM_Q0, ab, key)
            XS_RUNV(XS_SKIP0_ASM_Q1, XSF_SKIP0_ASM_Q1, ab, key)
            XS_RUNV(XS_SKIP0_ASM_Q2, XSF_SKIP0_ASM_Q2, ab, key)
            tap_addr(tap + 1, nab, nkey);
            XS_RUNV(XS_SKIP0_ASM_Q3, XSF_SKIP0_ASM_Q3, nab, nkey)
#pragma unroll
            for (int i = 0; i < 3; ++i) { ab[i] = nab[i]; key[i] = nkey[i]; }
        }
#pragma unroll 1
        for (; tap < 9; ++tap) {
            XS_RUN(XS_SLAB_ASM_Q0, XSF_SLAB_ASM_Q0, ab, key)      // four 16 KB slabs (hi + lo of 32 input channels) per tap
            XS_RUN(XS_SLAB_ASM_Q1, XSF_SLAB_ASM_Q1, ab, key)
            XS_RUN(XS_SLAB_ASM_Q2, XSF_SLAB_ASM_Q2, ab, key)
            tap_addr(tap + 1, nab, nkey);
            XS_RUN(XS_SLAB_ASM_Q3, XSF_SLAB_ASM_Q3, nab, nkey)
#pragma unroll
            for (int i = 0; i < 3; ++i) { ab[i] = nab[i]; key[i] = nkey[i]; }
        }
        // the last k-step prefetched garbage for a non-existent next slab; drain it, let the MFMAs retire, and make
        // sure every wave is done reading the activations before anyone overwrites them in place
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        __syncthreads();
        refresh_rb();
        store_layer(acc);
        __syncthreads();
    }
    if (clk && tid == 0) {   // the layers are done; the epilogue below (trunk dump / head convs) is ~1 % of the workgroup's life
        clk[blockIdx.x * 4 + 0] = clk_c0; clk[blockIdx.x * 4 + 1] = __builtin_readcyclecounter();
        clk[blockIdx.x * 4 + 2] = clk_r0; clk[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (out) {   // trunk activations as fp32 = hi + lo, 8 channels per thread and step
        float4 *go = reinterpret_cast<float4 *>(out + (size_t)pos0 * 90 * 128);
        for (int idx = tid; idx < nrows * 16; idx += Geo::THREADS) {
            const int r = idx >> 4, c = idx & 15;
            const int a = lds_addr(lds_row_of(r) * CV_ROWB, c);
            const uint4 h = *reinterpret_cast<const uint4 *>(smem + a), l = *reinterpret_cast<const uint4 *>(smem + a + Geo::LO_OFF);
            const f32x2 h0 = unpack_pair<F16>(h.x), h1 = unpack_pair<F16>(h.y), h2 = unpack_pair<F16>(h.z), h3 = unpack_pair<F16>(h.w);
            const f32x2 l0 = unpack_pair<F16>(l.x), l1 = unpack_pair<F16>(l.y), l2 = unpack_pair<F16>(l.z), l3 = unpack_pair<F16>(l.w);
            go[idx * 2 + 0] = make_float4(h0[0] + l0[0], h0[1] + l0[1], h1[0] + l1[0], h1[1] + l1[1]);
            go[idx * 2 + 1] = make_float4(h2[0] + l2[0], h2[1] + l2[1], h3[0] + l3[0], h3[1] + l3[1]);
        }
    }
    if (head_out) {
        const float *hw = reinterpret_cast<const float *>(smem + Geo::HEADW_OFF);   // staged at the prologue
        // one thread per board cell, all three head channels; chunks in a fixed order, eight products left to right: a
        // position's outputs do not depend on the row / workgroup it lands on
        for (int r = tid; r < nrows; r += Geo::THREADS) {
            const int rowoff = lds_row_of(r) * CV_ROWB, key = (rowoff >> 8) & 15;
            float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
#pragma unroll 4
            for (int c = 0; c < 16; ++c) {
                const int p = c ^ key;
                const uint4 h = *reinterpret_cast<const uint4 *>(smem + rowoff + (p << 4));
                const uint4 l = *reinterpret_cast<const uint4 *>(smem + rowoff + (p << 4) + Geo::LO_OFF);
                const f32x2 h0 = unpack_pair<F16>(h.x), h1 = unpack_pair<F16>(h.y), h2 = unpack_pair<F16>(h.z), h3 = unpack_pair<F16>(h.w);
                const f32x2 l0 = unpack_pair<F16>(l.x), l1 = unpack_pair<F16>(l.y), l2 = unpack_pair<F16>(l.z), l3 = unpack_pair<F16>(l.w);
                const float e[8] = {h0[0] + l0[0], h0[1] + l0[1], h1[0] + l1[0], h1[1] + l1[1],
                                    h2[0] + l2[0], h2[1] + l2[1], h3[0] + l3[0], h3[1] + l3[1]};
                const float *w0_ = hw + c * 8, *w1_ = hw + 128 + c * 8, *w2_ = hw + 256 + c * 8;
                acc0 += e[0] * w0_[0] + e[1] * w0_[1] + e[2] * w0_[2] + e[3] * w0_[3] + e[4] * w0_[4] + e[5] * w0_[5] + e[6] * w0_[6] + e[7] * w0_[7];
                acc1 += e[0] * w1_[0] + e[1] * w1_[1] + e[2] * w1_[2] + e[3] * w1_[3] + e[4] * w1_[4] + e[5] * w1_[5] + e[6] * w1_[6] + e[7] * w1_[7];
                acc2 += e[0] * w2_[0] + e[1] * w2_[1] + e[2] * w2_[2] + e[3] * w2_[3] + e[4] * w2_[4] + e[5] * w2_[5] + e[6] * w2_[6] + e[7] * w2_[7];
            }
            float *o = head_out + ((size_t)pos0 * 90 + r) * 3;
            o[0] = fmaxf(acc0 + head_b[0], 0.f);
            o[1] = fmaxf(acc1 + head_b[1], 0.f);
            o[2] = fmaxf(acc2 + head_b[2], 0.f);
        }
    }
}
#undef XS_SLAB
#undef XS_SLABV
#undef XS_ARGS
#undef XS_RUNV
#undef XS_RUN
#undef XS_LOADSET

}  // namespace czconv
