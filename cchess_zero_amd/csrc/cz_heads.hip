// cz_heads.hip — the three fully connected layers behind the head convolutions, in two launches.
//
// Reference: policy_value_network.py:62-63,72-74 — policy head: flatten [9,10,2] in (h,w,c) order -> FC 180->2086
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
//               so a B fragment is one coalesced 16-byte-per-lane load (L2-resident, 1.6 MB in all).
// k_value_fc  : fp32 VALU; thread t owns hidden unit t (its 90 weights in registers), 32 positions per
//               workgroup staged in LDS and read back as broadcast float4, wave + LDS reduction for the
//               256->1 layer.
#include "cz_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PFC_KB = 12;                   // K = 180 inputs padded to 12 k-blocks of 16
constexpr int PFC_N = CZ_NLABELS;            // 2086
constexpr int PFC_LT = (PFC_N + 31) / 32;    // 66 label tiles of 32
constexpr int PFC_WAVES = 4;
constexpr int PFC_POS = 32 * PFC_WAVES;      // 128 positions per workgroup, 32 per wave
constexpr int PFC_SPLIT = 4;                 // label tiles are dealt to 4 workgroups per position chunk (16-17 each)
constexpr int PFC_FRAG_U4 = PFC_KB * 64;     // one operand image (hi or lo) of one 32-row tile: 768 uint4 = 12 KB
constexpr int PFC_A_BYTES = PFC_WAVES * 2 * PFC_FRAG_U4 * 16;   // 98,304: hi + lo activation fragments of 4 waves
constexpr int PFC_W_BYTES = 2 * 2 * PFC_FRAG_U4 * 16;           // 49,152: two buffers of (hi, lo) weight fragments
constexpr int PFC_LDS_BYTES = PFC_A_BYTES + PFC_W_BYTES;        // 147,456

__device__ __forceinline__ void split_bf16(float x, __bf16 &hi, __bf16 &lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

// 128 positions per workgroup (32 per wave) x 16-17 label tiles.  Everything the MFMAs read comes from LDS:
//   activations: z is copied coalesced (it is contiguous per position) through a raw LDS buffer and converted to
//     bf16 hi + lo in MFMA A-fragment order (24 KB per wave) -- gathering fragments straight from global memory
//     costs 64 cache lines per load instruction (the first version of this kernel was TA-bound), and scattering
//     element by element costs ~30 VALU operations per element (the second version spent 18 us there);
//   weights: the (hi, lo) fragments of one label tile are 24 KB; the four waves share them through a double
//     buffer, the next tile's 6 x 16 bytes per thread being requested before the current tile's MFMAs.  Per-wave
//     streaming from L2 instead moved 405 MB per call; this moves 102 MB.
__global__ __launch_bounds__(256, 1) void k_policy_fc(const float *__restrict__ z, const uint4 *__restrict__ w_hi,
                                                       const uint4 *__restrict__ w_lo, const float *__restrict__ bias,
                                                       float *__restrict__ logits, int B, const int *__restrict__ bcount) {
    if (bcount) { const int live = *bcount; B = live < B ? live : B; }   // compact batches: first *bcount rows only
    if ((int)(blockIdx.x / PFC_SPLIT) * PFC_POS >= B) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint16_t (*a_hi)[64][8] = reinterpret_cast<uint16_t (*)[64][8]>(smem + (size_t)(wave * 2 + 0) * PFC_FRAG_U4 * 16);
    uint16_t (*a_lo)[64][8] = reinterpret_cast<uint16_t (*)[64][8]>(smem + (size_t)(wave * 2 + 1) * PFC_FRAG_U4 * 16);
    uint4 *wbuf = reinterpret_cast<uint4 *>(smem + PFC_A_BYTES);   // [2 buffers][hi | lo][12][64]
    const int chunk = blockIdx.x / PFC_SPLIT, part = blockIdx.x % PFC_SPLIT;
    const int p0 = chunk * PFC_POS + wave * 32;
    const int np = max(0, min(32, B - p0));

    // weight tile `lt` -> registers (6 x 16 bytes per thread: uint4 index tid + 256 q of [hi 768 | lo 768])
    uint4 wr0, wr1, wr2, wr3, wr4, wr5;
#define PFC_FETCH_W(LT)                                                                              \
    do {                                                                                             \
        const uint4 *fh_ = w_hi + (size_t)(LT) * PFC_FRAG_U4 + tid, *fl_ = w_lo + (size_t)(LT) * PFC_FRAG_U4 + tid; \
        wr0 = fh_[0]; wr1 = fh_[256]; wr2 = fh_[512]; wr3 = fl_[0]; wr4 = fl_[256]; wr5 = fl_[512];   \
    } while (0)
#define PFC_PARK_W(BUF)                                                                              \
    do {                                                                                             \
        uint4 *d_ = wbuf + (BUF) * 2 * PFC_FRAG_U4 + tid;                                            \
        d_[0] = wr0; d_[256] = wr1; d_[512] = wr2; d_[768] = wr3; d_[1024] = wr4; d_[1280] = wr5;    \
    } while (0)

    // Stage the activations, one 32-position block (= one wave's rows) per round: all 256 threads copy the block
    // (8640 contiguous floats) into a raw buffer borrowed from the weight double buffer, then thread (g = wave,
    // lane = (khalf, p)) builds the bf16 hi / lo A fragments of k-blocks 3g .. 3g+2 from it.  The copy of the
    // next block is in flight (registers) while the current one is converted.
    {
        float *raw = reinterpret_cast<float *>(wbuf);           // 34,560 of the 49,152 bytes
        constexpr int RAW_N = 32 * 270, RAW_Q = (RAW_N + 255) / 256;   // 8640 floats, 34 per thread
        const int pblk = chunk * PFC_POS;
        float v[RAW_Q];
        auto load_block = [&](int w) {
            const int nb = max(0, min(32, B - (pblk + 32 * w))) * 270;
            const float *src = z + (size_t)(pblk + 32 * w) * 270;
#pragma unroll
            for (int q = 0; q < RAW_Q; ++q) {
                const int i = tid + 256 * q;
                v[q] = i < nb ? src[i] : 0.f;
            }
        };
        load_block(0);
#pragma unroll 1
        for (int w = 0; w < PFC_WAVES; ++w) {
#pragma unroll
            for (int q = 0; q < RAW_Q; ++q) {
                const int i = tid + 256 * q;
                if (i < RAW_N) raw[i] = v[q];
            }
            __syncthreads();
            if (w + 1 < PFC_WAVES) load_block(w + 1);
            const int p = lane & 31, kh = lane >> 5;
            uint4 *dst_hi = reinterpret_cast<uint4 *>(smem + (size_t)(w * 2 + 0) * PFC_FRAG_U4 * 16);
            uint4 *dst_lo = reinterpret_cast<uint4 *>(smem + (size_t)(w * 2 + 1) * PFC_FRAG_U4 * 16);
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int kb = wave * 3 + kk;
                uint32_t ph[4], pl[4];
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2) {   // k = kb*16 + kh*8 + 2*j2 (+1): one cell, its two policy channels
                    const int cell = kb * 8 + kh * 4 + j2;
                    float x0 = 0.f, x1 = 0.f;
                    if (cell < 90) { x0 = raw[p * 270 + cell * 3]; x1 = raw[p * 270 + cell * 3 + 1]; }
                    __bf16 h0, l0, h1, l1;
                    split_bf16(x0, h0, l0);
                    split_bf16(x1, h1, l1);
                    ph[j2] = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
                    pl[j2] = (uint32_t)__builtin_bit_cast(uint16_t, l0) | ((uint32_t)__builtin_bit_cast(uint16_t, l1) << 16);
                }
                dst_hi[kb * 64 + lane] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
                dst_lo[kb * 64 + lane] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
            }
            __syncthreads();
        }
    }
    PFC_FETCH_W(part);
    PFC_PARK_W(0);
    __syncthreads();

    const int row0 = 4 * (lane >> 5);  // accumulator reg r holds row (r & 3) + 8 * (r >> 2) + row0, column lane & 31
    int buf = 0;
#pragma unroll 1
    for (int lt = part; lt < PFC_LT; lt += PFC_SPLIT) {
        const int label = lt * 32 + (lane & 31);
        const float bv = label < PFC_N ? bias[label] : 0.f;   // before the fetch: vmcnt retires in order
        // next tile's weights: in flight during this tile's MFMAs (past the end: a harmless re-fetch of this tile,
        // which keeps the code straight-line and the six registers out of scratch)
        PFC_FETCH_W(lt + PFC_SPLIT < PFC_LT ? lt + PFC_SPLIT : lt);
        __builtin_amdgcn_sched_barrier(0);   // left alone, the scheduler sinks these loads behind the MFMAs
        // three independent accumulation chains (one per product): back-to-back MFMAs on ONE accumulator wait for
        // each other's full latency, which made this loop twice as long as its issue time
        f32x16 acc, acc_lh = {}, acc_hl = {};
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = bv;
        const uint4 *wt = wbuf + buf * 2 * PFC_FRAG_U4 + lane;
#pragma unroll
        for (int kb = 0; kb < PFC_KB; ++kb) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(&a_hi[kb][lane][0]);
            const bf16x8 al = *reinterpret_cast<const bf16x8 *>(&a_lo[kb][lane][0]);
            const bf16x8 fh = __builtin_bit_cast(bf16x8, wt[kb * 64]), fl = __builtin_bit_cast(bf16x8, wt[PFC_FRAG_U4 + kb * 64]);
            acc_lh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, fh, acc_lh, 0, 0, 0);
            acc_hl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, fl, acc_hl, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, fh, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc_lh[r] + acc_hl[r];
#ifdef CZ_PFC_NOSTORE   // timing experiment (tools/fc_ubench.hip): keep the MFMAs alive, never store
        if (acc[0] == 123456.f && acc[7] == 654321.f) {
#else
        if (label < PFC_N && np > 0) {
#endif
            float *lp = logits + (size_t)(p0 + row0) * PFC_N + label;
            if (np == 32) {   // full wave: 16 unconditional stores, two 128-byte runs each
#pragma unroll
                for (int r = 0; r < 16; ++r) lp[(size_t)((r & 3) + 8 * (r >> 2)) * PFC_N] = acc[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = (r & 3) + 8 * (r >> 2) + row0;
                    if (p < np) lp[(size_t)((r & 3) + 8 * (r >> 2)) * PFC_N] = acc[r];
                }
            }
        }
        PFC_PARK_W(buf ^ 1);   // the other buffer was last read one tile ago, before the previous barrier
        __syncthreads();
        buf ^= 1;
    }
}

#undef PFC_FETCH_W
#undef PFC_PARK_W

constexpr int VFC_POS = 8;    // positions per workgroup: 1024 workgroups at B = 8192, several resident per CU

__global__ __launch_bounds__(256) void k_value_fc(const float *__restrict__ z, const float *__restrict__ w1t /*[90][256]*/,
                                                   const float *__restrict__ b1, const float *__restrict__ w2,
                                                   const float *__restrict__ b2, float *__restrict__ value, int B,
                                                   const int *__restrict__ bcount) {
    if (bcount) { const int live = *bcount; B = live < B ? live : B; }
    if ((int)blockIdx.x * VFC_POS >= B) return;
    __shared__ __attribute__((aligned(16))) float vin[VFC_POS][92];   // 90 inputs + 2 zeros: 23 float4 per position
    __shared__ float part[VFC_POS][256 + 8];   // per-position products of the 256 hidden units (padded rows)
    const int t = threadIdx.x;
    const int p0 = blockIdx.x * VFC_POS;
    const int np = min(VFC_POS, B - p0);
    for (int i = t; i < VFC_POS * 92; i += 256) {
        const int p = i / 92, c = i - p * 92;
        vin[p][c] = (p < np && c < 90) ? z[((size_t)(p0 + p) * 90 + c) * 3 + 2] : 0.f;
    }
    float w[92];
#pragma unroll
    for (int i = 0; i < 90; ++i) w[i] = w1t[i * 256 + t];
    w[90] = w[91] = 0.f;
    const float bt = b1[t], w2t = w2[t];
    __syncthreads();
    for (int p = 0; p < VFC_POS; ++p) {
        const float4 *vp = reinterpret_cast<const float4 *>(vin[p]);   // wave-uniform address: LDS broadcast
        float h = bt;
#pragma unroll
        for (int i = 0; i < 23; ++i) {
            const float4 v = vp[i];
            h = fmaf(w[4 * i + 0], v.x, h);
            h = fmaf(w[4 * i + 1], v.y, h);
            h = fmaf(w[4 * i + 2], v.z, h);
            h = fmaf(w[4 * i + 3], v.w, h);
        }
        part[p][t] = fmaxf(h, 0.f) * w2t;   // FC 256 -> 1 is summed below, all positions at once
    }
    __syncthreads();
    // thread (p = t / 8, g = t % 8) sums 32 of the 256 products of position p, then the 8 group sums are folded
    // with three shuffles: one short dependent chain per workgroup instead of one per position
    if (t < VFC_POS * 8) {
        const int p = t >> 3, g = t & 7;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += part[p][g + 8 * i];
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        if (g == 0 && p < np) value[p0 + p] = tanhf(s + b2[0]);
    }
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
        static bool attr_done[64] = {};   // per device: opt in to > 64 KB of dynamic LDS once
        if (c->device >= 0 && c->device < 64 && !attr_done[c->device]) {
            CZ_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_policy_fc), hipFuncAttributeMaxDynamicSharedMemorySize, PFC_LDS_BYTES));
            attr_done[c->device] = true;
        }
        const int chunks = (B + PFC_POS - 1) / PFC_POS;
        hipLaunchKernelGGL(k_policy_fc, dim3(chunks * PFC_SPLIT), dim3(256), PFC_LDS_BYTES, c->stream, z, (const uint4 *)pfc_w_hi,
                           (const uint4 *)pfc_w_lo, pfc_b, logits, B, c->batch_count);
        CZ_HIP(hipGetLastError());
    }
    if (value) {
        hipLaunchKernelGGL(k_value_fc, dim3((B + VFC_POS - 1) / VFC_POS), dim3(256), 0, c->stream, z, v1_wt, v1_b, v2_w, v2_b,
                           value, B, c->batch_count);
        CZ_HIP(hipGetLastError());
    }
    return CZ_OK;
}
