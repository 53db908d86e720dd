// cz_conv.hip — N1: the residual tower's 3x3 convolution as a fused MFMA implicit GEMM (gfx950).
//
// Replaces, per layer, tf.layers.conv2d(128, 3, 'SAME') + batch_norm(no affine) [+ residual add]
// + ReLU of the reference (policy_value_network.py:45-47, 151-162).  BN is folded into the packed
// weights/bias on the host (net.py), so one launch = one conv layer end to end.
//
// GEMM view (per layer):  M = B*90 board cells, N = 128 output channels, K = 9 taps * 128 channels.
//   - a workgroup owns 4 whole positions (360 rows -> 12 row tiles of 32, the last 24 rows padding):
//     their bf16 activations (92 KB) are loaded ONCE into LDS and stay there for all 9 taps; the
//     im2col shift of a tap is an LDS address offset, out-of-board taps read a zero row.
//   - the weight matrix (288 KB/layer, L2 resident) is streamed through two 16 KB LDS slabs
//     (64 input channels of one tap), prefetched to registers one slab ahead (issue-early /
//     write-late) so L2 latency hides under the MFMAs.
//   - 8 waves = 4 (row groups of 3 tiles) x 2 (column groups of 2 tiles); each wave keeps
//     3x2 accumulators of v_mfma_f32_32x32x16_bf16 (96 regs), A/B fragments by ds_read_b128.
//   - LDS rows are 256 B (128 bf16): the 16-byte chunk c of row r lives at chunk c ^ (r & 15), which
//     makes the column-slice fragment reads bank-conflict free (guide T2).
//   - epilogue: + bias, + residual, ReLU, bf16, staged through LDS and written as full 256-byte rows.
// Roofline: MFMA-bound; algorithmic flops per launch = 2 * B*90 * 1152 * 128.
#include "cz_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CV_P = 4;                 // positions per workgroup
constexpr int CV_ROWS = CV_P * 90;      // 360
constexpr int CV_RT = 3;                // row tiles per wave
constexpr int CV_CT = 2;                // col tiles per wave
constexpr int CV_ROWB = 256;            // bytes per LDS activation row
constexpr int CV_ACT_BYTES = (CV_ROWS + 1) * CV_ROWB;  // + zero row
constexpr int CV_SLAB_BYTES = 64 * 128 * 2;             // 16 KB
constexpr int CV_LDS_BYTES = CV_ACT_BYTES + 2 * CV_SLAB_BYTES;
constexpr int CV_THREADS = 512;

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// in/out/res: [B][90][128] bf16 (NHWC with H*W = 90).  wpk: [9 taps][16 kchunks][128 n][8] bf16.
__global__ __launch_bounds__(CV_THREADS) void k_conv3x3_c128(const uint16_t *__restrict__ in,
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
            if (r < nrows) v = g[idx];
            *reinterpret_cast<uint4 *>(act + r * CV_ROWB + ((c ^ (r & 15)) << 4)) = v;
        }
        if (tid < 16) *reinterpret_cast<uint4 *>(act + CV_ROWS * CV_ROWB + (tid << 4)) = make_uint4(0, 0, 0, 0);
    }
    // ---- slab 0 of the weights ----
    const uint4 *wg = reinterpret_cast<const uint4 *>(wpk);
    uint4 pre0 = wg[tid], pre1 = wg[tid + CV_THREADS];
    *reinterpret_cast<uint4 *>(wbuf + (tid << 4)) = pre0;
    *reinterpret_cast<uint4 *>(wbuf + ((tid + CV_THREADS) << 4)) = pre1;
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
    for (int s = 0; s < 18; ++s) {
        const int tap = s >> 1, half = s & 1;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        if (s + 1 < 18) {  // issue-early: next slab to registers
            pre0 = wg[(s + 1) * (CV_SLAB_BYTES / 16) + tid];
            pre1 = wg[(s + 1) * (CV_SLAB_BYTES / 16) + tid + CV_THREADS];
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
        for (int kk = 0; kk < 4; ++kk) {
            const int c = half * 8 + kk * 2 + khalf;  // 16-byte chunk (8 channels) of the activation row
            bf16x8 a[CV_RT], b[CV_CT];
#pragma unroll
            for (int i = 0; i < CV_RT; ++i)
                a[i] = *reinterpret_cast<const bf16x8 *>(act + abase[i] + ((c ^ asw[i]) << 4));
#pragma unroll
            for (int j = 0; j < CV_CT; ++j)
                b[j] = *reinterpret_cast<const bf16x8 *>(wb + (kk * 2 + khalf) * 2048 + bcol + j * 512);
#pragma unroll
            for (int i = 0; i < CV_RT; ++i)
#pragma unroll
                for (int j = 0; j < CV_CT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (s + 1 < 18) {  // write-late: the other slab buffer was last read in iteration s-1
            unsigned char *nb = wbuf + ((s + 1) & 1) * CV_SLAB_BYTES;
            *reinterpret_cast<uint4 *>(nb + (tid << 4)) = pre0;
            *reinterpret_cast<uint4 *>(nb + ((tid + CV_THREADS) << 4)) = pre1;
        }
        __syncthreads();
    }

    // ---- epilogue: bias (+ residual) (+ ReLU) -> bf16, through LDS, full-row stores ----
    // After the last barrier nobody reads `act` any more: reuse it as the [360][128] bf16 output tile
    // (same swizzle, so the row-wise read-back below is conflict free as well).
#pragma unroll
    for (int j = 0; j < CV_CT; ++j) {
        const int n = wc * 64 + j * 32 + l31;
        const float bn = bias[n];
#pragma unroll
        for (int i = 0; i < CV_RT; ++i) {
            const int r0 = 32 * (wr * CV_RT + i) + 4 * khalf;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int r = r0 + (e & 3) + 8 * (e >> 2);
                if (r < CV_ROWS) {
                    const float v = acc[i][j][e] + bn;
                    // stash fp32->bf16 later (after residual); keep fp32 precision for the add: store bf16 of v only
                    // when there is no residual, else add in the read-back pass.  Two bytes per element either way.
                    *reinterpret_cast<uint16_t *>(act + r * CV_ROWB + (((n >> 3) ^ (r & 15)) << 4) + ((n & 7) << 1)) =
                        res ? f32_to_bf16(v) : f32_to_bf16(relu ? fmaxf(v, 0.f) : v);
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

}  // namespace

extern "C" int cz_conv3x3_c128_bf16(cz_ctx *c, const void *in, const void *wpk, const float *bias, const void *residual,
                                    void *out, int B, int relu) {
    CZ_REQUIRE(c && in && wpk && bias && out && B >= 0, "cz_conv3x3_c128_bf16: null argument");
    if (B == 0) return CZ_OK;
    static bool attr_set = false;
    if (!attr_set) {
        CZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_conv3x3_c128), hipFuncAttributeMaxDynamicSharedMemorySize, CV_LDS_BYTES));
        attr_set = true;
    }
    const int grid = (B + CV_P - 1) / CV_P;
    hipLaunchKernelGGL(k_conv3x3_c128, dim3(grid), dim3(CV_THREADS), CV_LDS_BYTES, c->stream, (const uint16_t *)in,
                       (const uint16_t *)wpk, bias, (const uint16_t *)residual, (uint16_t *)out, B, relu);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}
