// cz_conv.hip — N1: the residual tower's 3x3 convolution as a fused MFMA implicit GEMM (gfx950).
//
// Replaces, per layer, tf.layers.conv2d(128, 3, 'SAME') + batch_norm(no affine) [+ residual add]
// + ReLU of the reference (policy_value_network.py:45-47, 151-162).  BN is folded into the packed
// weights/bias on the host (net.py), so one launch = one conv layer end to end.
//
// GEMM view (per layer):  M = B*90 board cells, N = 128 output channels, K = 9 taps * 128 channels.
//   - a workgroup owns P whole positions (P = 2: 180 rows -> 6 row tiles of 32, the last 12 rows are
//     padding): their bf16 activations (45 KB) are loaded ONCE into LDS and stay there for all 9
//     taps; the im2col shift of a tap is an LDS address offset, out-of-board taps read a zero row.
//     Two workgroups (77 KB of LDS each) share a CU so one's load/store phases hide under the
//     other's MFMA loop.
//   - the weight matrix (288 KB/layer, L2 resident) is streamed through two 16 KB LDS slabs
//     (64 input channels of one tap), prefetched to registers one slab ahead (issue-early /
//     write-late) so L2 latency hides under the MFMAs.
//   - 2P waves = P (row groups of 3 tiles) x 2 (column groups of 2 tiles); each wave keeps
//     3x2 accumulators of v_mfma_f32_32x32x16_bf16 (96 regs), A/B fragments by ds_read_b128.
//   - LDS rows are 256 B (128 bf16): the 16-byte chunk c of row r lives at chunk c ^ (r & 15), which
//     makes the column-slice fragment reads bank-conflict free (guide T2).
//   - epilogue: + bias, + residual, ReLU, bf16, staged through LDS and written as full 256-byte rows.
// Roofline: MFMA-bound; algorithmic flops per launch = 2 * B*90 * 1152 * 128.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace czconv {


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef CZ_CONV_P
#define CZ_CONV_P 4
#endif
#ifndef CZ_CONV_WAVES_PER_EU
#define CZ_CONV_WAVES_PER_EU 2
#endif
// ablation switches for tools/conv_ubench.hip (all 0 in the product build)
#ifndef CZ_ABL
#define CZ_ABL 0
#endif
#define CZ_ABL_NO_ACTLOAD 1
#define CZ_ABL_NO_EPILOGUE 2
#define CZ_ABL_NO_MFMA 4
#define CZ_ABL_NO_LDSREAD 8
#define CZ_ABL_NO_WSTREAM 16
// CZ_TTRACE: tools/tower_ubench.hip only — s_memtime stamps of one workgroup into `out` (clobbers it)
#ifndef CZ_TTRACE
#define CZ_TTRACE 0
#endif
// tower ablations: 1 = no MFMA, 2 = no fragment reads, 4 = no weight stream, 8 = no layer epilogue
#ifndef CZ_TABL
#define CZ_TABL 0
#endif
constexpr int CV_P = CZ_CONV_P;         // positions per workgroup (2 -> two workgroups share a CU)
constexpr int CV_ROWS = CV_P * 90;      // 180
constexpr int CV_RT = 3;                // row tiles per wave
constexpr int CV_CT = 2;                // col tiles per wave
constexpr int CV_ROWB = 256;            // bytes per LDS activation row
constexpr int CV_ACT_BYTES = (CV_ROWS + 1) * CV_ROWB;  // + zero row
constexpr int CV_SLAB_BYTES = 128 * 128 * 2;            // 32 KB: one whole tap
constexpr int CV_LDS_BYTES = CV_ACT_BYTES + 2 * CV_SLAB_BYTES;
constexpr int CV_THREADS = 128 * CV_P;  // (CV_P/2 * 2) row groups x 2 column groups of waves
constexpr int CV_PRE = (CV_SLAB_BYTES / 16) / CV_THREADS;  // uint4 prefetch registers per thread

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {  // round to nearest even; NaN stays NaN
    const uint32_t u = __float_as_uint(f);
    const uint32_t r = u + 0x7FFFu + ((u >> 16) & 1u);
    const bool nan = (u & 0x7FFFFFFFu) > 0x7F800000u;
    return (uint16_t)((nan ? (u | 0x400000u) : r) >> 16);
}

// in/out/res: [B][90][128] bf16 (NHWC with H*W = 90).  wpk: [9 taps][16 kchunks][128 n][8] bf16.
__global__ __launch_bounds__(CV_THREADS, CZ_CONV_WAVES_PER_EU) void k_conv3x3_c128(const uint16_t *__restrict__ in,
                                                             const uint16_t *__restrict__ wpk,
                                                             const float *__restrict__ bias,
                                                             const uint16_t *__restrict__ res,
                                                             uint16_t *__restrict__ out, int B, int relu) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *act = smem;
    unsigned char *wbuf = smem + CV_ACT_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int pos0 = blockIdx.x * CV_P;
    const int npos = (B - pos0) < CV_P ? (B - pos0) : CV_P;
    const int nrows = npos * 90;

    // ---- stage activations: contiguous 16-byte chunks, swizzled rows ----
    {
        const uint4 *g = reinterpret_cast<const uint4 *>(in + (size_t)pos0 * 90 * 128);
        for (int idx = tid; idx < CV_ROWS * 16; idx += CV_THREADS) {
            const int r = idx >> 4, c = idx & 15;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (!(CZ_ABL & CZ_ABL_NO_ACTLOAD) && r < nrows) v = g[idx];
            *reinterpret_cast<uint4 *>(act + r * CV_ROWB + ((c ^ (r & 15)) << 4)) = v;
        }
        if (tid < 16) *reinterpret_cast<uint4 *>(act + CV_ROWS * CV_ROWB + (tid << 4)) = make_uint4(0, 0, 0, 0);
    }
    // ---- slab 0 of the weights ----
    const uint4 *wg = reinterpret_cast<const uint4 *>(wpk);
    // prefetch registers as named scalars (an indexed array here ends up in scratch memory)
    static_assert(CV_PRE == 4, "slab prefetch is written for 4 x 16 B per thread");
#define CV_LD(q) uint4 pre##q = wg[tid + q * CV_THREADS];
#define CV_ST(q, base) *reinterpret_cast<uint4 *>((base) + ((tid + q * CV_THREADS) << 4)) = pre##q;
    CV_LD(0) CV_LD(1) CV_LD(2) CV_LD(3)
    CV_ST(0, wbuf) CV_ST(1, wbuf) CV_ST(2, wbuf) CV_ST(3, wbuf)
    __syncthreads();

    // per-lane geometry of the 3 row tiles this wave owns
    int hh[CV_RT], ww[CV_RT], rown[CV_RT];
#pragma unroll
    for (int i = 0; i < CV_RT; ++i) {
        const int r = 32 * (wr * CV_RT + i) + l31;
        rown[i] = r;
        const int pix = r % 90;
        hh[i] = pix / 10;
        ww[i] = pix - hh[i] * 10;
        if (r >= CV_ROWS) hh[i] = -100;  // padding rows: every tap invalid -> zero row
    }
    const int bcol = (wc * 64 + l31) << 4;  // byte offset of this lane's column inside a kchunk row group

    f32x16 acc[CV_RT][CV_CT];
#pragma unroll
    for (int i = 0; i < CV_RT; ++i)
#pragma unroll
        for (int j = 0; j < CV_CT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll 1
    for (int s = 0; s < 9; ++s) {
        const int tap = s;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        if (!(CZ_ABL & CZ_ABL_NO_WSTREAM) && s + 1 < 9) {  // issue-early: next slab to registers
            const uint4 *wn = wg + (s + 1) * (CV_SLAB_BYTES / 16) + tid;
            pre0 = wn[0]; pre1 = wn[CV_THREADS]; pre2 = wn[2 * CV_THREADS]; pre3 = wn[3 * CV_THREADS];
        }
        int abase[CV_RT], asw[CV_RT];
#pragma unroll
        for (int i = 0; i < CV_RT; ++i) {
            const int y = hh[i] + dy, x = ww[i] + dx;
            const bool ok = (y >= 0) && (y < 9) && (x >= 0) && (x < 10);
            const int rr = ok ? rown[i] + dy * 10 + dx : CV_ROWS;
            abase[i] = rr * CV_ROWB;
            asw[i] = rr & 15;
        }
        const unsigned char *wb = wbuf + (s & 1) * CV_SLAB_BYTES;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int c = kk * 2 + khalf;  // 16-byte chunk (8 channels) of the activation row
            bf16x8 a[CV_RT], b[CV_CT];
#pragma unroll
            for (int i = 0; i < CV_RT; ++i) {
                if (CZ_ABL & CZ_ABL_NO_LDSREAD) { for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(float)(abase[i] + e + c); }
                else a[i] = *reinterpret_cast<const bf16x8 *>(act + abase[i] + ((c ^ asw[i]) << 4));
            }
#pragma unroll
            for (int j = 0; j < CV_CT; ++j) {
                if (CZ_ABL & CZ_ABL_NO_LDSREAD) { for (int e = 0; e < 8; ++e) b[j][e] = (__bf16)(float)(bcol + e + kk + j); }
                else b[j] = *reinterpret_cast<const bf16x8 *>(wb + (kk * 2 + khalf) * 2048 + bcol + j * 512);
            }
#pragma unroll
            for (int i = 0; i < CV_RT; ++i)
#pragma unroll
                for (int j = 0; j < CV_CT; ++j) {
                    if (CZ_ABL & CZ_ABL_NO_MFMA) { asm volatile("" :: "v"(a[i]), "v"(b[j])); acc[i][j][0] += 1.0f; }
                    // weights as the MFMA "A" operand: D[channel][cell] -> a lane owns 4 consecutive channels
                    // of one board cell per accumulator quad, which the epilogue packs into 8-byte stores
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
                }
        }
        if (!(CZ_ABL & CZ_ABL_NO_WSTREAM) && s + 1 < 9) {  // write-late: the other slab buffer was last read in iteration s-1
            unsigned char *nb = wbuf + ((s + 1) & 1) * CV_SLAB_BYTES;
            CV_ST(0, nb) CV_ST(1, nb) CV_ST(2, nb) CV_ST(3, nb)
        }
        __syncthreads();
    }

    if (CZ_ABL & CZ_ABL_NO_EPILOGUE) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < CV_RT; ++i)
#pragma unroll
            for (int j = 0; j < CV_CT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) t += acc[i][j][e];
        if (t == 12345.678f) out[tid] = 1;  // keeps the accumulators live without the store pass
        return;
    }
    // ---- epilogue: bias (+ residual) (+ ReLU) -> bf16, through LDS, full-row stores ----
    // After the last barrier nobody reads `act` any more: reuse it as the [360][128] bf16 output tile
    // (same swizzle, so the row-wise read-back below is conflict free as well).
#pragma unroll
    for (int i = 0; i < CV_RT; ++i) {
        const int r = 32 * (wr * CV_RT + i) + l31;  // board cell (row of the NHWC matrix) this lane owns
        if (r < CV_ROWS) {
#pragma unroll
            for (int j = 0; j < CV_CT; ++j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n0 = wc * 64 + j * 32 + 8 * q + 4 * khalf;  // 4 consecutive output channels
                    const float4 bq = *reinterpret_cast<const float4 *>(bias + n0);
                    float v0 = acc[i][j][4 * q + 0] + bq.x, v1 = acc[i][j][4 * q + 1] + bq.y;
                    float v2 = acc[i][j][4 * q + 2] + bq.z, v3 = acc[i][j][4 * q + 3] + bq.w;
                    if (relu && !res) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                    uint2 pk;
                    pk.x = (uint32_t)f32_to_bf16(v0) | ((uint32_t)f32_to_bf16(v1) << 16);
                    pk.y = (uint32_t)f32_to_bf16(v2) | ((uint32_t)f32_to_bf16(v3) << 16);
                    *reinterpret_cast<uint2 *>(act + r * CV_ROWB + (((n0 >> 3) ^ (r & 15)) << 4) + ((n0 & 4) << 1)) = pk;
                }
            }
        }
    }
    __syncthreads();
    {
        uint4 *go = reinterpret_cast<uint4 *>(out + (size_t)pos0 * 90 * 128);
        const uint4 *gr = res ? reinterpret_cast<const uint4 *>(res + (size_t)pos0 * 90 * 128) : nullptr;
        for (int idx = tid; idx < nrows * 16; idx += CV_THREADS) {
            const int r = idx >> 4, c = idx & 15;
            uint4 v = *reinterpret_cast<const uint4 *>(act + r * CV_ROWB + ((c ^ (r & 15)) << 4));
            if (gr) {
                const uint4 q = gr[idx];
                uint32_t vv[4] = {v.x, v.y, v.z, v.w}, qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float lo = bf16_to_f32((uint16_t)(vv[t] & 0xFFFF)) + bf16_to_f32((uint16_t)(qq[t] & 0xFFFF));
                    float hi = bf16_to_f32((uint16_t)(vv[t] >> 16)) + bf16_to_f32((uint16_t)(qq[t] >> 16));
                    if (relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
                    vv[t] = (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
                }
                v = make_uint4(vv[0], vv[1], vv[2], vv[3]);
            }
            go[idx] = v;
        }
    }
}


// =================================================================================================
// k_tower_c128: the WHOLE residual tower in one launch.
//
// A workgroup keeps the activations of TW_P = 2 positions in LDS for all 2*nblocks conv layers:
//   U (block input x, later block output y, in place) and V (the mid activation t), 45 KB each.
// Nothing but the first input and the last output touches HBM, so the kernel is bound by MFMA issue,
// LDS and the L2->LDS weight stream, not by HBM (a per-layer kernel moves 69 KB/position/layer, which
// is balanced against the MFMA peak even when perfectly overlapped).
//   waves   4 = 2 (row groups of 3 tiles) x 2 (column groups of 2 tiles), one per SIMD; 3x2
//           accumulators of v_mfma_f32_32x32x16_bf16 each.
//   weights stream L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR staging) in 16 KB slabs
//           (64 input channels of one tap) through a ring of FOUR buffers: the DMA of slab g+3 is
//           issued right after the barrier in the middle of slab g (its buffer was last read in slab
//           g-1), and that barrier — preceded by a counted s_waitcnt vmcnt(4) — publishes slab g+1,
//           whose DMA has been in flight for two slabs.  No wave waits at a slab boundary and the
//           stream runs across layer boundaries.
//   frags   A/B fragments rotate through four register sets, fetched two k-steps (12 MFMAs) ahead of
//           their use, also across slab boundaries.
//   epilogue per layer: + bias [+ x from U, same cells] -> ReLU -> bf16 (v_cvt_pk_bf16_f32), 8-byte
//           LDS accesses, nothing leaves the CU.
// Roofline: MFMA; algorithmic flops per launch = nblocks * 2 * (2 * B*90 * 1152 * 128).
// =================================================================================================
constexpr int TW_P = 2;
constexpr int TW_ROWS = TW_P * 90;                 // 180 board cells, 6 row tiles of 32 (12 rows padding)
constexpr int TW_THREADS = 256;
constexpr int TW_BUF_BYTES = TW_ROWS * CV_ROWB;    // 46080
constexpr int TW_ZERO_OFF = 2 * TW_BUF_BYTES;      // one zero row shared by U and V
constexpr int TW_W_OFF = TW_ZERO_OFF + CV_ROWB;
constexpr int TW_SLAB_BYTES = 64 * 128 * 2;        // half a tap: 16 KB
constexpr int TW_NBUF = 4;
constexpr int TW_LDS_BYTES = TW_W_OFF + TW_NBUF * TW_SLAB_BYTES;   // 157,952
constexpr int TW_SLAB_U4 = TW_SLAB_BYTES / 16;     // 1024 uint4 per slab, 4 per thread

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {  // one v_cvt_pk_bf16_f32 (RNE)
    f32x2 v; v[0] = lo; v[1] = hi;
    bf16x2 b = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<uint32_t *>(&b);
}

struct TwFrag { bf16x8 a[CV_RT]; bf16x8 b[CV_CT]; };

__global__ __launch_bounds__(TW_THREADS, 1) void k_tower_c128(const uint16_t *__restrict__ in,
                                                              const uint16_t *__restrict__ wpk,   // [L][9][16][128][8]
                                                              const float *__restrict__ bias,     // [L][128]
                                                              uint16_t *__restrict__ out, int B, int nlayers) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *wbuf = smem + TW_W_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int pos0 = blockIdx.x * TW_P;
    const int npos = (B - pos0) < TW_P ? (B - pos0) : TW_P;
    const int nrows = npos * 90;
    const int nslabs = nlayers * 18;
    const uint4 *wg = reinterpret_cast<const uint4 *>(wpk) + tid;

    // a wave-instruction of LDS-DMA moves 1 KB to [wave-uniform LDS base + lane*16]; 4 per thread and slab
    auto dma_slab = [&](int slab) {
        if (CZ_TABL & 4) return;
        const uint4 *src = wg + (size_t)slab * TW_SLAB_U4;
        unsigned char *dst = wbuf + (slab % TW_NBUF) * TW_SLAB_BYTES + (wave << 10);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + q * TW_THREADS),
                                             (__attribute__((address_space(3))) void *)(dst + q * 4096), 16, 0, 0);
    };
    for (int q = 0; q < 3 && q < nslabs; ++q) dma_slab(q);
    {   // stage x into U (swizzled rows), clear the zero row
        const uint4 *g = reinterpret_cast<const uint4 *>(in + (size_t)pos0 * 90 * 128);
        for (int idx = tid; idx < TW_ROWS * 16; idx += TW_THREADS) {
            const int r = idx >> 4, c = idx & 15;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < nrows) v = g[idx];
            *reinterpret_cast<uint4 *>(smem + r * CV_ROWB + ((c ^ (r & 15)) << 4)) = v;
        }
        if (tid < 16) *reinterpret_cast<uint4 *>(smem + TW_ZERO_OFF + (tid << 4)) = make_uint4(0, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int hh[CV_RT], ww[CV_RT], rown[CV_RT];
#pragma unroll
    for (int i = 0; i < CV_RT; ++i) {
        const int r = 32 * (wr * CV_RT + i) + l31;
        rown[i] = r;
        const int pix = r % 90;
        hh[i] = pix / 10;
        ww[i] = pix - hh[i] * 10;
        if (r >= TW_ROWS) hh[i] = -100;
    }
    const int bcol = (wc * 64 + l31) << 4;

    // activation-row addressing of one tap: out-of-board taps read the zero row
    auto tap_addr = [&](int tap, int src_off, int (&abase)[CV_RT], int (&asw)[CV_RT]) {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
        for (int i = 0; i < CV_RT; ++i) {
            const int y = hh[i] + dy, x = ww[i] + dx;
            const bool ok = (y >= 0) && (y < 9) && (x >= 0) && (x < 10);
            const int rr = rown[i] + dy * 10 + dx;
            abase[i] = ok ? src_off + rr * CV_ROWB : TW_ZERO_OFF;
            asw[i] = ok ? (rr & 15) : 0;
        }
    };
    // fragments of k-step kk (0..3) of a slab: 8 channels per lane from 3 cells + 2 weight columns
    auto load_frag = [&](TwFrag &f, const int (&abase)[CV_RT], const int (&asw)[CV_RT], int half, int kk,
                         const unsigned char *wb) {
        const int c = half * 8 + kk * 2 + khalf;
        if (CZ_TABL & 2) {
#pragma unroll
            for (int i = 0; i < CV_RT; ++i) asm volatile("" : "+v"(f.a[i]) : "v"(abase[i] + c));
#pragma unroll
            for (int j = 0; j < CV_CT; ++j) asm volatile("" : "+v"(f.b[j]) : "v"(bcol + kk));
            return;
        }
#pragma unroll
        for (int i = 0; i < CV_RT; ++i)
            f.a[i] = *reinterpret_cast<const bf16x8 *>(smem + abase[i] + ((c ^ asw[i]) << 4));
#pragma unroll
        for (int j = 0; j < CV_CT; ++j)
            f.b[j] = *reinterpret_cast<const bf16x8 *>(wb + (kk * 2 + khalf) * 2048 + bcol + j * 512);
    };

#if CZ_TTRACE
    unsigned long long *trace = reinterpret_cast<unsigned long long *>(out) + (size_t)(blockIdx.x == 1500 ? 0 : (1 << 20));
    const bool tr = (blockIdx.x == 1500) && tid == 0;
#define TW_STAMP(slot) if (tr) trace[slot] = __builtin_amdgcn_s_memtime();
#else
#define TW_STAMP(slot)
#endif
    int g = 0;  // running slab index over all layers
#pragma unroll 1
    for (int layer = 0; layer < nlayers; ++layer) {
        const int src_off = (layer & 1) ? TW_BUF_BYTES : 0;   // even layers read U write V, odd read V write U
        const int dst_off = (layer & 1) ? 0 : TW_BUF_BYTES;
        f32x16 acc[CV_RT][CV_CT];
#pragma unroll
        for (int i = 0; i < CV_RT; ++i)
#pragma unroll
            for (int j = 0; j < CV_CT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        auto mma = [&](const TwFrag &f) {
#pragma unroll
            for (int i = 0; i < CV_RT; ++i)
#pragma unroll
                for (int j = 0; j < CV_CT; ++j) {
                    if (CZ_TABL & 1) { asm volatile("" :: "v"(f.a[i]), "v"(f.b[j])); acc[i][j][0] += 1.0f; }
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[j], f.a[i], acc[i][j], 0, 0, 0);
                }
        };
        // this layer's bias for the 2 x 4 channel quads a lane owns: in flight during the whole main loop
        float4 breg[CV_CT][4];
#pragma unroll
        for (int j = 0; j < CV_CT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                breg[j][q] = *reinterpret_cast<const float4 *>(bias + layer * 128 + wc * 64 + j * 32 + 8 * q + 4 * khalf);
        int abase[CV_RT], asw[CV_RT];
        TwFrag f0 = {}, f1 = {}, f2 = {}, f3 = {};
        tap_addr(0, src_off, abase, asw);
        {   // the only exposed fragment loads of the layer
            const unsigned char *wb0 = wbuf + (g % TW_NBUF) * TW_SLAB_BYTES;
            load_frag(f0, abase, asw, 0, 0, wb0);
            load_frag(f1, abase, asw, 0, 1, wb0);
        }

#pragma unroll 1
        for (int s = 0; s < 18; ++s, ++g) {
            const int half = s & 1;
            const unsigned char *wb = wbuf + (g % TW_NBUF) * TW_SLAB_BYTES;
            TW_STAMP(8 + g * 4 + 0)
            load_frag(f2, abase, asw, half, 2, wb);
            mma(f0);
            load_frag(f3, abase, asw, half, 3, wb);
            mma(f1);
            TW_STAMP(8 + g * 4 + 1)
            // middle of the slab.  Slab g+1 (DMA issued two barriers ago) must have landed for this wave:
            // at most the 4 DMA instructions of slab g+2 may still be in flight.  The barrier publishes
            // slab g+1 and proves every wave has left slab g-1, whose buffer the DMA of slab g+3 refills.
            // No lgkmcnt drain: the fragment reads in flight belong to the current slab.
            // Everything below is branch-free on purpose (see the note on MFMAs under branches; a branch
            // here also lets the compiler sink the next slab's fragment reads to the top of the loop).
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            TW_STAMP(8 + g * 4 + 2)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            TW_STAMP(8 + g * 4 + 3)
            {
                const int gn = g + 3 < nslabs ? g + 3 : nslabs - 1;   // past the end: re-fetch the last slab (unused)
                const uint4 *src = wg + (size_t)gn * TW_SLAB_U4;
                unsigned char *dst = wbuf + ((g + 3) % TW_NBUF) * TW_SLAB_BYTES + (wave << 10);
                if (!(CZ_TABL & 4)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + q * TW_THREADS),
                                                         (__attribute__((address_space(3))) void *)(dst + q * 4096), 16, 0, 0);
                }
            }
            // fragments of the next slab's first two k-steps (its buffer was just published); after the last
            // slab of a layer they fetch in-bounds garbage that the next layer's prologue overwrites.
            {
                const unsigned char *nb = wbuf + ((g + 1) % TW_NBUF) * TW_SLAB_BYTES;
                tap_addr((s + 1) >> 1, src_off, abase, asw);
                load_frag(f0, abase, asw, half ^ 1, 0, nb);
                mma(f2);
                load_frag(f1, abase, asw, half ^ 1, 1, nb);
                mma(f3);
            }
        }
        // layer epilogue, entirely in LDS.  Odd layers (second conv of a block) add the block input x,
        // which still sits in U at exactly the cells this lane is about to overwrite.
        TW_STAMP(4096 + layer * 2)
#pragma unroll
        for (int i = 0; i < ((CZ_TABL & 8) ? 0 : CV_RT); ++i) {
            const int r = 32 * (wr * CV_RT + i) + l31;
            const bool live = r < TW_ROWS;
            const int rc = live ? r : 0;   // padding rows: compute on row 0's address, never store
            uint2 xr[CV_CT][4];
            uint2 *cell[CV_CT][4];
#pragma unroll
            for (int j = 0; j < CV_CT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n0 = wc * 64 + j * 32 + 8 * q + 4 * khalf;
                    cell[j][q] = reinterpret_cast<uint2 *>(smem + dst_off + rc * CV_ROWB + (((n0 >> 3) ^ (rc & 15)) << 4) + ((n0 & 4) << 1));
                    xr[j][q] = (layer & 1) ? *cell[j][q] : make_uint2(0, 0);
                }
#pragma unroll
            for (int j = 0; j < CV_CT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bq = breg[j][q];
                    float v0 = acc[i][j][4 * q + 0] + bq.x + __uint_as_float(xr[j][q].x << 16);
                    float v1 = acc[i][j][4 * q + 1] + bq.y + __uint_as_float(xr[j][q].x & 0xFFFF0000u);
                    float v2 = acc[i][j][4 * q + 2] + bq.z + __uint_as_float(xr[j][q].y << 16);
                    float v3 = acc[i][j][4 * q + 3] + bq.w + __uint_as_float(xr[j][q].y & 0xFFFF0000u);
                    uint2 pk;
                    pk.x = pack_bf16x2(fmaxf(v0, 0.f), fmaxf(v1, 0.f));
                    pk.y = pack_bf16x2(fmaxf(v2, 0.f), fmaxf(v3, 0.f));
                    if (live) *cell[j][q] = pk;
                }
        }
        if (CZ_TABL & 8) { float t = 0.f; for (int i = 0; i < CV_RT; ++i) for (int j = 0; j < CV_CT; ++j) for (int e = 0; e < 16; ++e) t += acc[i][j][e]; if (t == 1234.5f) out[tid] = 1; }
        __syncthreads();
        TW_STAMP(4096 + layer * 2 + 1)
    }
    if (CZ_TTRACE) return;
    {   // the tower output sits in U (nlayers is even): full-row coalesced stores
        uint4 *go = reinterpret_cast<uint4 *>(out + (size_t)pos0 * 90 * 128);
        const int fin = (nlayers & 1) ? TW_BUF_BYTES : 0;
        for (int idx = tid; idx < nrows * 16; idx += TW_THREADS) {
            const int r = idx >> 4, c = idx & 15;
            go[idx] = *reinterpret_cast<const uint4 *>(smem + fin + r * CV_ROWB + ((c ^ (r & 15)) << 4));
        }
    }
}

}  // namespace czconv
