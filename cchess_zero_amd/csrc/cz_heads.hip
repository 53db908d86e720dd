// cz_heads.hip — the three fully connected layers behind the head convolutions, in two launches.
//
// Reference: policy_value_network.py:56-74 — policy head: flatten [9,10,2] in (h,w,c) order -> FC 180->2086
// (raw logits, no softmax before the search reads them, quirk Q3); value head: flatten [9,10,1] -> FC 90->256 +
// ReLU -> FC 256->1 + tanh.  Input is the post-BN/ReLU output of the two 1x1 head convolutions as the tower
// kernel leaves it: z [B][90][3] f32 (channels 0,1 = policy, 2 = value).
//
// k_policy_fc : GEMM [B x 180] x [180 x 2086] on v_mfma_f32_32x32x16_bf16 with both operands split into
//               bf16 hi + lo parts and three products (hi*hi + hi*lo + lo*hi) accumulated in fp32, i.e. ~16
//               mantissa bits per operand: the FC is not the place where the net's precision is decided and
//               it is output-write bound (B*2086*4 bytes), so the extra MFMA passes are free.
//               A operand = activations (rows = positions), B operand = weights (columns = labels) so that in
//               the accumulator layout consecutive lanes hold consecutive labels of one position: every store
//               instruction writes two 128-byte runs.
//               Weights arrive pre-packed in fragment order [label/32][k/16][lane][8] (net.py packs them once),
//               so a B fragment is one coalesced 16-byte-per-lane load that the 4 waves of a workgroup share
//               through L1/L2.
// k_value_fc  : fp32 VALU; thread t owns hidden unit t (its 90 weights in registers), 32 positions per
//               workgroup staged in LDS, wave + LDS reduction for the 256->1 layer.
#include "cz_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PFC_K = 180, PFC_KB = 12;      // 12 k-blocks of 16 (K padded to 192)
constexpr int PFC_N = CZ_NLABELS;            // 2086
constexpr int PFC_LT = (PFC_N + 31) / 32;    // 66 label tiles of 32
constexpr int PFC_POS_PER_WG = 128;          // 4 waves x 32 positions
constexpr int PFC_SPLIT = 6;                 // label tiles are dealt to 6 workgroups per position chunk (11 each)

__device__ __forceinline__ void split_bf16(float x, __bf16 &hi, __bf16 &lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

__global__ __launch_bounds__(256, 2) void k_policy_fc(const float *__restrict__ z, const uint4 *__restrict__ w_hi,
                                                       const uint4 *__restrict__ w_lo, const float *__restrict__ bias,
                                                       float *__restrict__ logits, int B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x / PFC_SPLIT, part = blockIdx.x % PFC_SPLIT;
    const int p0 = chunk * PFC_POS_PER_WG + wave * 32;
    if (p0 >= B) return;
    // A fragments of this wave's 32 positions, all 12 k-blocks: lane holds position p0 + (lane & 31),
    // k = kb*16 + (lane >> 5)*8 + j; input k of the policy FC is z[pos][k >> 1][k & 1]
    const int pos = p0 + (lane & 31);
    const bool pos_ok = pos < B;
    const float *zp = z + (size_t)(pos_ok ? pos : 0) * 270;
    bf16x8 a_hi[PFC_KB], a_lo[PFC_KB];
#pragma unroll
    for (int kb = 0; kb < PFC_KB; ++kb) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = kb * 16 + (lane >> 5) * 8 + j;
            float x = (pos_ok && k < PFC_K) ? zp[(k >> 1) * 3 + (k & 1)] : 0.f;
            __bf16 h, l;
            split_bf16(x, h, l);
            a_hi[kb][j] = h;
            a_lo[kb][j] = l;
        }
    }
    const int row0 = 4 * (lane >> 5);  // accumulator reg r holds row (r & 3) + 8 * (r >> 2) + row0, column lane & 31
    for (int lt = part; lt < PFC_LT; lt += PFC_SPLIT) {
        const int label = lt * 32 + (lane & 31);
        const float bv = label < PFC_N ? bias[label] : 0.f;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bv;
        const uint4 *wh = w_hi + ((size_t)lt * PFC_KB) * 64 + lane;
        const uint4 *wl = w_lo + ((size_t)lt * PFC_KB) * 64 + lane;
        uint4 bh[PFC_KB], bl[PFC_KB];
#pragma unroll
        for (int kb = 0; kb < PFC_KB; ++kb) {
            bh[kb] = wh[kb * 64];
            bl[kb] = wl[kb * 64];
        }
#pragma unroll
        for (int kb = 0; kb < PFC_KB; ++kb) {
            const bf16x8 fh = __builtin_bit_cast(bf16x8, bh[kb]), fl = __builtin_bit_cast(bf16x8, bl[kb]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[kb], fh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[kb], fl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[kb], fh, acc, 0, 0, 0);
        }
        if (label < PFC_N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + (r & 3) + 8 * (r >> 2) + row0;
                if (p < B) logits[(size_t)p * PFC_N + label] = acc[r];
            }
        }
    }
}

constexpr int VFC_POS = 32;

__global__ __launch_bounds__(256) void k_value_fc(const float *__restrict__ z, const float *__restrict__ w1t /*[90][256]*/,
                                                   const float *__restrict__ b1, const float *__restrict__ w2,
                                                   const float *__restrict__ b2, float *__restrict__ value, int B) {
    __shared__ float vin[VFC_POS][92];
    __shared__ float red[VFC_POS][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int p0 = blockIdx.x * VFC_POS;
    const int np = min(VFC_POS, B - p0);
    for (int i = t; i < VFC_POS * 90; i += 256) {
        const int p = i / 90, c = i - p * 90;
        vin[p][c] = p < np ? z[((size_t)(p0 + p) * 90 + c) * 3 + 2] : 0.f;
    }
    float w[90];
#pragma unroll
    for (int i = 0; i < 90; ++i) w[i] = w1t[i * 256 + t];
    const float bt = b1[t], w2t = w2[t];
    __syncthreads();
    for (int p = 0; p < VFC_POS; ++p) {
        float h = bt;
#pragma unroll
        for (int i = 0; i < 90; ++i) h = fmaf(w[i], vin[p][i], h);
        float s = fmaxf(h, 0.f) * w2t;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) red[p][wave] = s;
    }
    __syncthreads();
    if (t < np) value[p0 + t] = tanhf(((red[t][0] + red[t][1]) + (red[t][2] + red[t][3])) + b2[0]);
}

}  // namespace

extern "C" int cz_fc_heads_f32(cz_ctx *c, const float *z, const void *pfc_w_hi, const void *pfc_w_lo, const float *pfc_b,
                               const float *v1_wt, const float *v1_b, const float *v2_w, const float *v2_b,
                               float *logits, float *value, int B) {
    CZ_REQUIRE(c && B >= 0, "cz_fc_heads_f32: null context / negative batch");
    if (B == 0) return CZ_OK;
    CZ_REQUIRE(z && (logits || value), "cz_fc_heads_f32: null argument");
    CZ_REQUIRE(!logits || (pfc_w_hi && pfc_w_lo && pfc_b), "cz_fc_heads_f32: logits need the packed policy FC weights");
    CZ_REQUIRE(!value || (v1_wt && v1_b && v2_w && v2_b), "cz_fc_heads_f32: value needs the value FC weights");
    if (logits) {
        const int chunks = (B + PFC_POS_PER_WG - 1) / PFC_POS_PER_WG;
        hipLaunchKernelGGL(k_policy_fc, dim3(chunks * PFC_SPLIT), dim3(256), 0, c->stream, z, (const uint4 *)pfc_w_hi,
                           (const uint4 *)pfc_w_lo, pfc_b, logits, B);
        CZ_HIP(hipGetLastError());
    }
    if (value) {
        hipLaunchKernelGGL(k_value_fc, dim3((B + VFC_POS - 1) / VFC_POS), dim3(256), 0, c->stream, z, v1_wt, v1_b, v2_w, v2_b,
                           value, B);
        CZ_HIP(hipGetLastError());
    }
    return CZ_OK;
}
