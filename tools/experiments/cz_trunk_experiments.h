// tools/experiments/cz_trunk_experiments.h — the trunk-kernel variants that were built, measured and NOT adopted
// (rounds 1-3).  They are kept buildable as the record of those experiments:
//   k_tower_c128    "4w": 2 positions / 4 waves, activation buffers U and V
//   k_towerp_c128   "pw": 4 positions, one per wave, accumulators in AGPRs, no activation barrier
//   k_towersk_c128  "sk": the two half-workgroups of k_tower8_c128 four slabs apart on an 8-slot ring
//   k_towerd_c128   "d" : weight fragments straight from global memory, no LDS ring
// ("2x", k_tower8_c128 with 2 positions / 4 waves and two workgroups per CU, was a template instance of the product
// kernel; it was removed with round 4's clean-up, see git history before that commit.)
// Build: tools/experiments/build.sh; check + A/B timing against the product kernel: tools/experiments/variants_check.hip.
#pragma once
#include "../../cchess_zero_amd/csrc/cz_conv_kernel.h"

namespace czconv {

#include "cz_experiments_slab_asm.inc"

#define TW_LOADSET(CA, OB0, OB1, X, AB, KEY, VB)                                                     \
    asm volatile(                                                                                    \
        "v_xor_b32 %[t0], " #CA ", %[k0]\n\t"                                                        \
        "v_xor_b32 %[t1], " #CA ", %[k1]\n\t"                                                        \
        "v_xor_b32 %[t2], " #CA ", %[k2]\n\t"                                                        \
        "v_lshl_add_u32 %[t0], %[t0], 4, %[b0]\n\t"                                                  \
        "v_lshl_add_u32 %[t1], %[t1], 4, %[b1]\n\t"                                                  \
        "v_lshl_add_u32 %[t2], %[t2], 4, %[b2]\n\t"                                                  \
        "ds_read_b128 %[xa0], %[t0]\n\t"                                                             \
        "ds_read_b128 %[xa1], %[t1]\n\t"                                                             \
        "ds_read_b128 %[xa2], %[t2]\n\t"                                                             \
        "ds_read_b128 %[xb0], %[vb] offset:" #OB0 "\n\t"                                             \
        "ds_read_b128 %[xb1], %[vb] offset:" #OB1 "\n\t"                                             \
        : [xa0] "=&v"(X.a[0]), [xa1] "=&v"(X.a[1]), [xa2] "=&v"(X.a[2]), [xb0] "=&v"(X.b[0]),         \
          [xb1] "=&v"(X.b[1]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)                          \
        : [k0] "v"(KEY[0]), [k1] "v"(KEY[1]), [k2] "v"(KEY[2]), [b0] "v"(AB[0]), [b1] "v"(AB[1]),      \
          [b2] "v"(AB[2]), [vb] "v"(VB)                                                               \
        : "memory")

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
constexpr int TW_LDS_BYTES = TW_W_OFF + TW_NBUF * TW_SLAB_BYTES;   // 157,952
constexpr int TW_SLAB_U4 = TW_SLAB_BYTES / 16;     // 1024 uint4 per slab, 4 per thread

__global__ __launch_bounds__(TW_THREADS, 1) void k_tower_c128(const uint16_t *__restrict__ in,
                                                              const uint16_t *__restrict__ wpk,   // [L][9][16][128][8]
                                                              const float *__restrict__ bias,     // [L][128]
                                                              uint16_t *__restrict__ out,         // trunk [B][90][128] or NULL
                                                              const float *__restrict__ head_w,   // [3][128] or NULL
                                                              const float *__restrict__ head_b,   // [3]
                                                              float *__restrict__ head_out,       // [B][90][3] or NULL
                                                              const uint16_t *__restrict__ planes,  // [B][90][16] bf16 or NULL
                                                              const uint16_t *__restrict__ w0,      // [9][2][128][8] bf16
                                                              const float *__restrict__ b0,         // [128]
                                                              int B, int nlayers) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *wbuf = smem + TW_W_OFF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int pos0 = blockIdx.x * TW_P;
    const int npos = (B - pos0) < TW_P ? (B - pos0) : TW_P;
    const int nrows = npos * 90;
    const int nslabs = nlayers * 18;
    // LDS addressing: rows of 256 B; the 16-byte chunk c of a row lives at chunk c ^ (absolute row & 15)
    auto lds_addr = [](int row_byte_off, int c) { return row_byte_off + ((c ^ ((row_byte_off >> 8) & 15)) << 4); };

    // a wave-instruction of LDS-DMA moves 1 KB to [wave-uniform LDS base + lane*16]; 4 per thread and slab.
    // Past the end of the stream the last slab is re-fetched into a free buffer (never read): the slab body
    // stays branch-free and the vmcnt accounting constant.
    static_assert(TW_NBUF == 4, "ring index is computed with & 3");
    const unsigned dma_voff = (unsigned)tid << 4;   // this lane's 16 bytes inside a 4 KB quarter-slab
    auto dma_slab = [&](int slab) {
        const int gs = slab < nslabs ? slab : nslabs - 1;
        // uniform 64-bit base + 32-bit lane offset: selects the SGPR-base form of global_load_lds
        const unsigned char *src = reinterpret_cast<const unsigned char *>(wpk) + (size_t)gs * TW_SLAB_BYTES;
        unsigned char *dst = wbuf + ((unsigned)slab & 3u) * TW_SLAB_BYTES + (wave << 10);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + q * 4096 + dma_voff),
                                             (__attribute__((address_space(3))) void *)(dst + q * 4096), 16, 0, 0);
    };
    for (int q = 0; q < 3; ++q) dma_slab(q);
    if (planes == nullptr) {   // stage x into U (swizzled rows)
        const uint4 *g = reinterpret_cast<const uint4 *>(in + (size_t)pos0 * 90 * 128);
        for (int idx = tid; idx < TW_ROWS * 16; idx += TW_THREADS) {
            const int r = idx >> 4, c = idx & 15;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < nrows) v = g[idx];
            *reinterpret_cast<uint4 *>(smem + lds_addr(r * CV_ROWB, c)) = v;
        }
    } else {                   // stage the 16-channel input planes (32 B per cell) into the V region
        const uint4 *g = reinterpret_cast<const uint4 *>(planes + (size_t)pos0 * 90 * 16);
        for (int idx = tid; idx < TW_ROWS * 2; idx += TW_THREADS) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < nrows * 2) v = g[idx];
            *reinterpret_cast<uint4 *>(smem + TW_BUF_BYTES + (idx << 4)) = v;
        }
    }
    if (tid < 16) *reinterpret_cast<uint4 *>(smem + TW_ZERO_OFF + (tid << 4)) = make_uint4(0, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // per-lane geometry of the 3 row tiles this wave owns: byte offset of the cell's row inside a buffer and
    // a 9-bit mask of the taps that stay on the 9x10 board
    int rowb[CV_RT], tapmask[CV_RT];
#pragma unroll
    for (int i = 0; i < CV_RT; ++i) {
        const int r = 32 * (wr * CV_RT + i) + l31;
        const int pix = r % 90, h = pix / 10, w = pix - h * 10;
        rowb[i] = r * CV_ROWB;
        int m = 0;
        for (int t = 0; t < 9; ++t) {
            const int y = h + t / 3 - 1, x = w + t % 3 - 1;
            if (r < TW_ROWS && y >= 0 && y < 9 && x >= 0 && x < 10) m |= 1 << t;
        }
        tapmask[i] = m;
    }
    // activation-row addressing of one tap: out-of-board taps (and padding rows) read the zero row.
    // key = swizzle key of the source row with the lane's k-half folded in.
    auto tap_addr = [&](int tap, int src_off, int (&ab)[CV_RT], int (&key)[CV_RT]) {
        const int delta = src_off + ((tap / 3 - 1) * 10 + (tap - (tap / 3) * 3 - 1)) * CV_ROWB;
#pragma unroll
        for (int i = 0; i < CV_RT; ++i) {
            const int a = ((tapmask[i] >> tap) & 1) ? rowb[i] + delta : TW_ZERO_OFF;
            ab[i] = a;
            // a masked lane reads the all-zero row, but in the 16-byte slot its REAL (off-board) neighbour row would have
            // used: the 16 lanes of a ds_read_b128 group then still hit 16 distinct slots.  With the zero row's own swizzle
            // key every group containing a border cell paid a 2-way bank conflict (24 % of the LDS cycles).
            key[i] = (((rowb[i] + delta) >> 8) & 15) ^ khalf;
        }
    };
    const int vb0 = TW_W_OFF + khalf * 2048 + ((wc * 64 + l31) << 4);   // this lane's B column in slab buffer 0
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff0 = (unsigned)tid << 4, voff1 = voff0 + 4096u, voff2 = voff0 + 8192u, voff3 = voff0 + 12288u;
    int keep;

    // Layer epilogue, entirely in LDS: + bias [+ x] -> ReLU -> bf16.  For the second conv of a block the block
    // input x still sits in U at exactly the cells this lane is about to overwrite.
    // The bias is already in the accumulators (they are initialised with it, not with zero).
    auto layer_epilogue = [&](f32x16 (&acc)[CV_RT][CV_CT], int dst_off, bool residual) {
#pragma unroll
        for (int i = 0; i < CV_RT; ++i) {
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
                    cell[j][q] = reinterpret_cast<uint2 *>(smem + lds_addr(dst_off + rc * CV_ROWB, n0 >> 3) + ((n0 & 4) << 1));
                    xr[j][q] = residual ? *cell[j][q] : make_uint2(0, 0);
                }
#pragma unroll
            for (int j = 0; j < CV_CT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v0 = acc[i][j][4 * q + 0] + __uint_as_float(xr[j][q].x << 16);
                    float v1 = acc[i][j][4 * q + 1] + __uint_as_float(xr[j][q].x & 0xFFFF0000u);
                    float v2 = acc[i][j][4 * q + 2] + __uint_as_float(xr[j][q].y << 16);
                    float v3 = acc[i][j][4 * q + 3] + __uint_as_float(xr[j][q].y & 0xFFFF0000u);
                    uint2 pk;
                    pk.x = pack_bf16x2(fmaxf(v0, 0.f), fmaxf(v1, 0.f));
                    pk.y = pack_bf16x2(fmaxf(v2, 0.f), fmaxf(v3, 0.f));
                    if (live) *cell[j][q] = pk;
                }
        }
    };

    // accumulators start at the layer's (BN-folded) bias: acc[i][j][4q+e] belongs to channel wc*64 + j*32 + 8q + 4*khalf + e
    auto init_acc = [&](f32x16 (&acc)[CV_RT][CV_CT], const float *bl) {
#pragma unroll
        for (int j = 0; j < CV_CT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4 *>(bl + wc * 64 + j * 32 + 8 * q + 4 * khalf);
#pragma unroll
                for (int i = 0; i < CV_RT; ++i) {
                    acc[i][j][4 * q + 0] = bq.x; acc[i][j][4 * q + 1] = bq.y;
                    acc[i][j][4 * q + 2] = bq.z; acc[i][j][4 * q + 3] = bq.w;
                }
            }
    };

    if (planes != nullptr) {
        // ---- first layer: conv3x3(14 -> 128, input padded to 16 channels) + BN + ReLU (policy_value_network.py:45-47)
        // one k-step (16 channels) per tap; planes are in the V region (32 B per cell), the result goes to U.
        bf16x8 wf[9][CV_CT];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < CV_CT; ++j)
                wf[t][j] = *reinterpret_cast<const bf16x8 *>(w0 + ((size_t)((t * 2 + khalf) * 128 + wc * 64 + j * 32 + l31) << 3));
        f32x16 acc[CV_RT][CV_CT];
        init_acc(acc, b0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int shift = (t / 3 - 1) * 10 + (t % 3 - 1);
            bf16x8 af[CV_RT];
#pragma unroll
            for (int i = 0; i < CV_RT; ++i) {
                const int a = ((tapmask[i] >> t) & 1) ? TW_BUF_BYTES + ((rowb[i] >> 3) + shift * 32) : TW_ZERO_OFF;
                af[i] = *reinterpret_cast<const bf16x8 *>(smem + a + khalf * 16);
            }
#pragma unroll
            for (int i = 0; i < CV_RT; ++i)
#pragma unroll
                for (int j = 0; j < CV_CT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][j], af[i], acc[i][j], 0, 0, 0);
        }
        layer_epilogue(acc, 0, false);
        __syncthreads();
    }

    int g = 0;  // running slab index over all layers
#pragma unroll 1
    for (int layer = 0; layer < nlayers; ++layer) {
        const int src_off = (layer & 1) ? TW_BUF_BYTES : 0;   // even layers read U write V, odd read V write U
        const int dst_off = (layer & 1) ? 0 : TW_BUF_BYTES;
        f32x16 acc[CV_RT][CV_CT];
        init_acc(acc, bias + layer * 128);
        int ab[CV_RT], key[CV_RT], nab[CV_RT], nkey[CV_RT], t0, t1, t2;
        TwFrag f0, f1, f2, f3;
        tap_addr(0, src_off, ab, key);
        {   // the only exposed fragment loads of the layer.  Issued from asm as well: LDS reads the compiler
            // knows about would make it drain lgkmcnt(0) in front of the first k-steps of every iteration.
            const int vb = vb0 + (((unsigned)g & 3u) << 14);
            TW_LOADSET(0, 0, 512, f0, ab, key, vb);
            TW_LOADSET(2, 4096, 4608, f1, ab, key, vb);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // One asm statement per slab (generated by tools/gen_tower_asm.py): 24 MFMAs with the fragment reads two
        // k-steps ahead, the mid-slab `s_waitcnt vmcnt(4); s_barrier` (slab g+1 — DMA issued two barriers ago —
        // has landed and is published; every wave has left slab g-1) and the 4 LDS-DMA instructions that refill
        // slab g-1's buffer with slab g+3, all hand-placed in MFMA issue shadows.  All LDS byte offsets are
        // relative to LDS address 0 (the kernel has no static LDS, the dynamic region starts there).
#define TW_SLAB(ASMSTR, NAB, NKEY)                                                                               \
        asm volatile(ASMSTR                                                                                      \
            : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]),          \
              [c20] "+v"(acc[2][0]), [c21] "+v"(acc[2][1]),                                                        \
              [f0a0] "+v"(f0.a[0]), [f0a1] "+v"(f0.a[1]), [f0a2] "+v"(f0.a[2]), [f0b0] "+v"(f0.b[0]), [f0b1] "+v"(f0.b[1]), \
              [f1a0] "+v"(f1.a[0]), [f1a1] "+v"(f1.a[1]), [f1a2] "+v"(f1.a[2]), [f1b0] "+v"(f1.b[0]), [f1b1] "+v"(f1.b[1]), \
              [f2a0] "=&v"(f2.a[0]), [f2a1] "=&v"(f2.a[1]), [f2a2] "=&v"(f2.a[2]), [f2b0] "=&v"(f2.b[0]), [f2b1] "=&v"(f2.b[1]), \
              [f3a0] "=&v"(f3.a[0]), [f3a1] "=&v"(f3.a[1]), [f3a2] "=&v"(f3.a[2]), [f3b0] "=&v"(f3.b[0]), [f3b1] "=&v"(f3.b[1]), \
              [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [keep] "=&s"(keep)                                     \
            : [ab0] "v"(ab[0]), [ab1] "v"(ab[1]), [ab2] "v"(ab[2]), [key0] "v"(key[0]), [key1] "v"(key[1]),          \
              [key2] "v"(key[2]), [nab0] "v"(NAB[0]), [nab1] "v"(NAB[1]), [nab2] "v"(NAB[2]), [nkey0] "v"(NKEY[0]),   \
              [nkey1] "v"(NKEY[1]), [nkey2] "v"(NKEY[2]), [vb] "v"(vb), [vbn] "v"(vbn), [voff0] "v"(voff0),          \
              [voff1] "v"(voff1), [voff2] "v"(voff2), [voff3] "v"(voff3), [sbase] "s"(sbase), [ldst] "s"(ldst)        \
            : "memory", "scc")
#define TW_SLAB_ARGS()                                                                                          \
        const int vb = vb0 + (((unsigned)g & 3u) << 14), vbn = vb0 + ((((unsigned)g + 1u) & 3u) << 14);         \
        const int gn = g + 3 < nslabs ? g + 3 : nslabs - 1; /* past the end: re-fetch the last slab (never read) */ \
        const unsigned char *sbase = reinterpret_cast<const unsigned char *>(wpk) + (size_t)gn * TW_SLAB_BYTES;  \
        const int ldst = TW_W_OFF + ((((unsigned)g + 3u) & 3u) << 14) + (wave_u << 10);

#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            {   // ---- slab (tap, channels 0..63): the next slab reads the same cells ----
                TW_SLAB_ARGS()
                TW_SLAB(TW_SLAB_ASM_H0, ab, key);
                ++g;
            }
            {   // ---- slab (tap, channels 64..127): the next slab belongs to the next tap ----
                tap_addr(tap + 1, src_off, nab, nkey);   // tap 9 after the last tap: in-bounds garbage, unused
                TW_SLAB_ARGS()
                TW_SLAB(TW_SLAB_ASM_H1, nab, nkey);
#pragma unroll
                for (int i = 0; i < CV_RT; ++i) { ab[i] = nab[i]; key[i] = nkey[i]; }
                ++g;
            }
        }
        // the MFMAs were issued from inline asm: give the last ones time to retire before the accumulators are read
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        layer_epilogue(acc, dst_off, (layer & 1) != 0);
        __syncthreads();
    }
    // the ring's tail DMAs (re-fetches of the last slab into free buffers) must land before the LDS is given back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int fin = (nlayers & 1) ? TW_BUF_BYTES : 0;   // the tower output sits in U (nlayers is even)
    if (out) {   // full-row coalesced stores of the trunk
        uint4 *go = reinterpret_cast<uint4 *>(out + (size_t)pos0 * 90 * 128);
        for (int idx = tid; idx < nrows * 16; idx += TW_THREADS) {
            const int r = idx >> 4, c = idx & 15;
            go[idx] = *reinterpret_cast<const uint4 *>(smem + lds_addr(fin + r * CV_ROWB, c));
        }
    }
    if (head_out) {
        // Fused head 1x1 convolutions (policy_value_network.py:57-59,68-70: conv1x1(128->2) and conv1x1(128->1),
        // BN folded, ReLU) straight from the LDS-resident trunk: 3 dot products of length 128 per board cell,
        // so only 12 B per cell leave the CU instead of the 256 B trunk row.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail DMAs of the weight ring
        __syncthreads();
        float *hw = reinterpret_cast<float *>(wbuf);
        for (int i = tid; i < 3 * 128; i += TW_THREADS) hw[i] = head_w[i];
        __syncthreads();
        for (int idx = tid; idx < nrows * 3; idx += TW_THREADS) {
            const int r = idx / 3, c3 = idx - r * 3;
            const int rowoff = fin + r * CV_ROWB, key = (rowoff >> 8) & 15;
            const float *w = hw + c3 * 128;
            float acc = 0.f;
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int c = it;                   // logical chunk = channels 8c .. 8c+7, always summed in this order:
                const int p = c ^ key;              // a position's result must not depend on the row / thread it lands on
                const uint4 v = *reinterpret_cast<const uint4 *>(smem + rowoff + (p << 4));
                const float *wc8 = w + c * 8;
                acc += __uint_as_float(v.x << 16) * wc8[0] + __uint_as_float(v.x & 0xFFFF0000u) * wc8[1]
                     + __uint_as_float(v.y << 16) * wc8[2] + __uint_as_float(v.y & 0xFFFF0000u) * wc8[3]
                     + __uint_as_float(v.z << 16) * wc8[4] + __uint_as_float(v.z & 0xFFFF0000u) * wc8[5]
                     + __uint_as_float(v.w << 16) * wc8[6] + __uint_as_float(v.w & 0xFFFF0000u) * wc8[7];
            }
            head_out[((size_t)pos0 * 90 + r) * 3 + c3] = fmaxf(acc + head_b[c3], 0.f);
        }
    }
}

// =================================================================================================
// k_towersk_c128 ("sk", opt-in: CCHESS_TOWER_VARIANT=sk): k_tower8_c128 with its two half-workgroups a fixed number of slabs
// apart.  tools/tower_trace.hip: 13.4 % of a layer of k_tower8_c128 is the layer boundary (epilogue, accumulator
// initialisation: ~540 VALU instructions per wave) with the MFMA pipe idle, because the weight ring's per-slab barrier keeps
// all eight waves in lock-step and both waves of a SIMD reach the boundary together.  Here
//   * waves 0-3 (half A) own positions 0-1 (rows 0..179), waves 4-7 (half B) positions 2-3 (rows 180..359): six 32-row
//     tiles per half, the last 12 lanes of each half dead — a tap never leaves its position, so the halves share no row;
//   * each SIMD hosts one wave of each half; B runs the same program SKEW = 4 barriers behind A (it starts with four bare
//     barriers, A ends with four), so A's layer boundary falls under B's slabs 32..35 and B's under A's slabs 1..4;
//   * the ring is 8 slots of 8 KB (32 input channels of a tap, one barrier per slab): a slab stays until B has read it,
//     SKEW + 4 slots; only A's waves feed it (two 1 KB pieces per wave and slab), B finds its slabs published by the
//     barriers it shares with A;
//   * the layer-boundary barriers are the same s_barrier (every wave must execute the same number): three per layer and
//     half — reads of U done | first two tile rows stored | all stored and visible.
// Arithmetic per output element is that of k_tower8_c128 (same taps, same k order, same rounding): bit-identical outputs.
// =================================================================================================
struct SkGeo {
    static constexpr int P = 4, ROWS = 360, THREADS = 512, HALF_ROWS = 180, SKEW = 4, NBUF = 8;
    static constexpr int SLAB_BYTES = 32 * 128 * 2, SLAB_SHIFT = 13, SLABS_PER_LAYER = 36;
    static constexpr int ZERO_OFF = ROWS * CV_ROWB;
    static constexpr int W_OFF = ZERO_OFF + CV_ROWB;
    static constexpr int LDS_BYTES = W_OFF + NBUF * SLAB_BYTES;   // 157,952
    static constexpr int PLANES_OFF = W_OFF + 3 * SLAB_BYTES;     // 11.5 KB over ring slots 3-4 (first written by the DMA of slabs 3, 4: issued in slabs 0, 1)
    static constexpr int HEADW_OFF = LDS_BYTES;
    static constexpr int LDS_TOTAL = LDS_BYTES + 3 * 128 * 4;
};
constexpr int SK_P = 4, SK_THREADS = SkGeo::THREADS, SK_LDS_BYTES = SkGeo::LDS_TOTAL;

template <bool F16>
__global__ __launch_bounds__(512, 2) void k_towersk_c128(const uint16_t *__restrict__ in,
                                                         const uint16_t *__restrict__ wpk,
                                                         const float *__restrict__ bias,
                                                         uint16_t *__restrict__ out,
                                                         const float *__restrict__ head_w,
                                                         const float *__restrict__ head_b,
                                                         float *__restrict__ head_out,
                                                         const uint16_t *__restrict__ planes,
                                                         const uint16_t *__restrict__ w0,
                                                         const float *__restrict__ b0,
                                                         int B, int nlayers,
                                                         const int *__restrict__ bcount) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Geo = SkGeo;
    constexpr int P = Geo::P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int pos0 = blockIdx.x * P;
    if (bcount) {
        const int live = *bcount;
        B = live < B ? live : B;
    }
    if (pos0 >= B) return;
    const int npos = (B - pos0) < P ? (B - pos0) : P;
    const int nrows = npos * 90;
    const int nslabs = nlayers * Geo::SLABS_PER_LAYER;
    auto lds_addr = [](int row_byte_off, int c) { return row_byte_off + ((c ^ ((row_byte_off >> 8) & 15)) << 4); };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int isa = __builtin_amdgcn_readfirstlane(wave_u < 4 ? 1 : 0);   // half A feeds the ring and runs ahead (an SGPR: the slab asm branches on it)
    const unsigned voff0 = ((unsigned)tid & 255u) << 4, voff1 = voff0 + 4096u;   // A's lane offsets inside an 8 KB slab
    // row r of the [360][128] activation matrix owned by lane l31 of this wave's tile i; dead past the half's 180 rows
    auto row_of = [&](int i, bool &live) {
        const int lr = 32 * ((wr & 1) * CV_RT + i) + l31;
        live = lr < Geo::HALF_ROWS;
        return (wr >> 1) * Geo::HALF_ROWS + lr;
    };

    auto dma_slab = [&](int slab) {   // prologue only (half A); the loop issues its DMAs from the slab asm
        const unsigned char *src = reinterpret_cast<const unsigned char *>(wpk) + (size_t)slab * Geo::SLAB_BYTES;
        unsigned char *dst = smem + Geo::W_OFF + ((unsigned)slab & 7u) * Geo::SLAB_BYTES + ((wave_u & 3) << 10);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + voff0),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + voff1),
                                         (__attribute__((address_space(3))) void *)(dst + 4096), 16, 0, 0);
    };
    if (isa)
        for (int q = 0; q < 3; ++q) dma_slab(q < nslabs ? q : nslabs - 1);
    if (planes == nullptr) {
        const uint4 *g = reinterpret_cast<const uint4 *>(in + (size_t)pos0 * 90 * 128);
        for (int idx = tid; idx < Geo::ROWS * 16; idx += Geo::THREADS) {
            const int r = idx >> 4, c = idx & 15;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < nrows) v = g[idx];
            *reinterpret_cast<uint4 *>(smem + lds_addr(r * CV_ROWB, c)) = v;
        }
    } else {
        const uint4 *g = reinterpret_cast<const uint4 *>(planes + (size_t)pos0 * 90 * 16);
        for (int idx = tid; idx < Geo::ROWS * 2; idx += Geo::THREADS) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < nrows * 2) v = g[idx];
            *reinterpret_cast<uint4 *>(smem + Geo::PLANES_OFF + (idx << 4)) = v;
        }
    }
    if (tid < 16) *reinterpret_cast<uint4 *>(smem + Geo::ZERO_OFF + (tid << 4)) = make_uint4(0, 0, 0, 0);
    if (head_out && tid < 3 * 128 / 4)
        reinterpret_cast<float4 *>(smem + Geo::HEADW_OFF)[tid] = reinterpret_cast<const float4 *>(head_w)[tid];
    bf16x8 wf[9][CV_CT];
    if (planes != nullptr) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < CV_CT; ++j)
                wf[t][j] = *reinterpret_cast<const bf16x8 *>(w0 + ((size_t)((t * 2 + (lane >> 5)) * 128 + (wave & 1) * 64 + j * 32 + (lane & 31)) << 3));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int rowb[CV_RT], tapmask[CV_RT];
#pragma unroll
    for (int i = 0; i < CV_RT; ++i) {
        bool live;
        const int r = row_of(i, live);
        const int pix = r % 90, h = pix / 10, w = pix - h * 10;
        rowb[i] = r * CV_ROWB;
        int m = 0;
        for (int t = 0; t < 9; ++t) {
            const int y = h + t / 3 - 1, x = w + t % 3 - 1;
            if (live && y >= 0 && y < 9 && x >= 0 && x < 10) m |= 1 << t;
        }
        tapmask[i] = m;
    }
    auto tap_addr = [&](int tap, int (&ab)[CV_RT], int (&key)[CV_RT]) {
        const int delta = ((tap / 3 - 1) * 10 + (tap - (tap / 3) * 3 - 1)) * CV_ROWB;
#pragma unroll
        for (int i = 0; i < CV_RT; ++i) {
            const int a = ((tapmask[i] >> tap) & 1) ? rowb[i] + delta : Geo::ZERO_OFF;
            ab[i] = a;
            key[i] = (((rowb[i] + delta) >> 8) & 15) ^ khalf;   // see k_tower8_c128
        }
    };
    const int vb0 = Geo::W_OFF + khalf * 2048 + ((wc * 64 + l31) << 4);
    int keep;

    int rb[CV_RT];
    auto refresh_rb = [&]() {
#pragma unroll
        for (int i = 0; i < CV_RT; ++i) {
            bool live;
            const int r = row_of(i, live);
            rb[i] = (live ? r : 0) * CV_ROWB;
            asm volatile("" : "+v"(rb[i]));
        }
    };
    auto cell_ptr = [&](int i, int j, int q, bool &live) -> uint2 * {
        row_of(i, live);
        const int n0 = wc * 64 + j * 32 + 8 * q + 4 * khalf;
        return reinterpret_cast<uint2 *>(smem + lds_addr(rb[i], n0 >> 3) + ((n0 & 4) << 1));
    };
    uint2 xreg[CV_RT][CV_CT][4];
    auto init_acc = [&](f32x16 (&acc)[CV_RT][CV_CT], const float *bl, bool add_x) {
#pragma unroll
        for (int j = 0; j < CV_CT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4 *>(bl + wc * 64 + j * 32 + 8 * q + 4 * khalf);
#pragma unroll
                for (int i = 0; i < CV_RT; ++i) {
                    float a0 = bq.x, a1 = bq.y, a2 = bq.z, a3 = bq.w;
                    if (add_x) {
                        const uint2 x = xreg[i][j][q];
                        const f32x2 xl = unpack_pair<F16>(x.x), xh = unpack_pair<F16>(x.y);
                        a0 += xl[0]; a1 += xl[1]; a2 += xh[0]; a3 += xh[1];
                    }
                    acc[i][j][4 * q + 0] = a0; acc[i][j][4 * q + 1] = a1; acc[i][j][4 * q + 2] = a2; acc[i][j][4 * q + 3] = a3;
                }
            }
    };
    auto store_rows = [&](f32x16 (&acc)[CV_RT][CV_CT], int i) {   // ReLU -> 16 bit -> U in place, tile row i (see k_tower8_c128)
#pragma unroll
        for (int j = 0; j < CV_CT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bool live;
                uint2 *cell = cell_ptr(i, j, q, live);
                const s16x2 z = {0, 0};
                const s16x2 rl = __builtin_elementwise_max(
                    __builtin_bit_cast(s16x2, pack_pair<F16>(f32x2{acc[i][j][4 * q + 0], acc[i][j][4 * q + 1]})), z);
                const s16x2 rh = __builtin_elementwise_max(
                    __builtin_bit_cast(s16x2, pack_pair<F16>(f32x2{acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]})), z);
                if (live) *cell = make_uint2(__builtin_bit_cast(uint32_t, rl), __builtin_bit_cast(uint32_t, rh));
            }
    };
    // the one barrier of this kernel: every wave executes the same number of them, wherever it is in its program
    auto tick = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    if (planes != nullptr) {   // first layer: conv3x3(14 -> 128) + BN + ReLU, one k-step per tap; both halves together
        f32x16 acc[CV_RT][CV_CT];
        init_acc(acc, b0, false);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int shift = (t / 3 - 1) * 10 + (t % 3 - 1);
            bf16x8 af[CV_RT];
#pragma unroll
            for (int i = 0; i < CV_RT; ++i) {
                const int a = ((tapmask[i] >> t) & 1) ? Geo::PLANES_OFF + ((rowb[i] >> 3) + shift * 32) : Geo::ZERO_OFF;
                af[i] = *reinterpret_cast<const bf16x8 *>(smem + a + khalf * 16);
            }
#pragma unroll
            for (int i = 0; i < CV_RT; ++i)
#pragma unroll
                for (int j = 0; j < CV_CT; ++j)
                    acc[i][j] = mfma_32x32x16<F16>(wf[t][j], af[i], acc[i][j]);
        }
        refresh_rb();
#pragma unroll
        for (int i = 0; i < CV_RT; ++i) store_rows(acc, i);
        __syncthreads();
    }

#define SK_SLAB(ASMSTR, NAB, NKEY)                                                                               \
        asm volatile(ASMSTR                                                                                      \
            : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]),          \
              [c20] "+v"(acc[2][0]), [c21] "+v"(acc[2][1]),                                                        \
              [f0a0] "+v"(f0.a[0]), [f0a1] "+v"(f0.a[1]), [f0a2] "+v"(f0.a[2]), [f0b0] "+v"(f0.b[0]), [f0b1] "+v"(f0.b[1]), \
              [f1a0] "=&v"(f1.a[0]), [f1a1] "=&v"(f1.a[1]), [f1a2] "=&v"(f1.a[2]), [f1b0] "=&v"(f1.b[0]), [f1b1] "=&v"(f1.b[1]), \
              [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [keep] "=&s"(keep)                                     \
            : [ab0] "v"(ab[0]), [ab1] "v"(ab[1]), [ab2] "v"(ab[2]), [key0] "v"(key[0]), [key1] "v"(key[1]),          \
              [key2] "v"(key[2]), [nab0] "v"(NAB[0]), [nab1] "v"(NAB[1]), [nab2] "v"(NAB[2]), [nkey0] "v"(NKEY[0]),   \
              [nkey1] "v"(NKEY[1]), [nkey2] "v"(NKEY[2]), [vb] "v"(vb), [vbn] "v"(vbn), [voff0] "v"(voff0),          \
              [voff1] "v"(voff1), [sbase] "s"(sbase), [ldst] "s"(ldst), [wv] "s"(wave_u)                              \
            : "memory", "scc")
#define SK_RUN(BF, HF, NAB, NKEY)                                                                               \
        {                                                                                                       \
            const int vb = vb0 + (((unsigned)g & 7u) << Geo::SLAB_SHIFT), vbn = vb0 + ((((unsigned)g + 1u) & 7u) << Geo::SLAB_SHIFT); \
            const int gn = g + 3 < nslabs ? g + 3 : nslabs - 1;                                                 \
            const unsigned char *sbase = reinterpret_cast<const unsigned char *>(wpk) + (size_t)gn * Geo::SLAB_BYTES; \
            const int ldst = Geo::W_OFF + ((((unsigned)g + 3u) & 7u) << Geo::SLAB_SHIFT) + ((wave_u & 3) << 10); \
            if constexpr (F16) { SK_SLAB(HF, NAB, NKEY); } else { SK_SLAB(BF, NAB, NKEY); }                     \
            ++g;                                                                                                \
        }

    int g = 0;
    if (!isa)
        for (int s = 0; s < Geo::SKEW; ++s) tick();   // half B starts SKEW barriers late
#pragma unroll 1
    for (int layer = 0; layer < nlayers; ++layer) {
        f32x16 acc[CV_RT][CV_CT];
        CZ_T8_STAMP(0);
        refresh_rb();
        if (!(layer & 1)) {   // first conv of a block: remember x, start from the bias
#pragma unroll
            for (int i = 0; i < CV_RT; ++i)
#pragma unroll
                for (int j = 0; j < CV_CT; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { bool live; xreg[i][j][q] = *cell_ptr(i, j, q, live); }
            init_acc(acc, bias + layer * 128, false);
        } else {
            init_acc(acc, bias + layer * 128, true);
        }
        int ab[CV_RT], key[CV_RT], nab[CV_RT], nkey[CV_RT], t0, t1, t2;
        TwFrag f0, f1;
        tap_addr(0, ab, key);
        {
            const int vb = vb0 + (((unsigned)g & 7u) << Geo::SLAB_SHIFT);
            TW_LOADSET(0, 0, 512, f0, ab, key, vb);   // waited for by the first k-step itself
        }
        CZ_T8_STAMP(1);
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            SK_RUN(TWS_SLAB_ASM_Q0, TWSF_SLAB_ASM_Q0, ab, key)
            SK_RUN(TWS_SLAB_ASM_Q1, TWSF_SLAB_ASM_Q1, ab, key)
            SK_RUN(TWS_SLAB_ASM_Q2, TWSF_SLAB_ASM_Q2, ab, key)
            tap_addr(tap + 1, nab, nkey);
            SK_RUN(TWS_SLAB_ASM_Q3, TWSF_SLAB_ASM_Q3, nab, nkey)
#pragma unroll
            for (int i = 0; i < CV_RT; ++i) { ab[i] = nab[i]; key[i] = nkey[i]; }
        }
        // drain the garbage prefetch of a non-existent next slab, let the MFMAs retire
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        CZ_T8_STAMP(2);
        tick();               // every wave of this half is done reading U (the other half never touches these rows)
#if defined(CZ_T8_TRACE)
        CZ_T8_STAMP(13);
#endif
        refresh_rb();
        store_rows(acc, 0);
        store_rows(acc, 1);
        tick();
        store_rows(acc, 2);
        tick();               // (waits for lgkmcnt(0) first) this half's new U is complete and visible
        CZ_T8_STAMP(3);
    }
    if (isa)
        for (int s = 0; s < Geo::SKEW; ++s) tick();   // half A waits for B's last SKEW barriers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (out) {
        uint4 *go = reinterpret_cast<uint4 *>(out + (size_t)pos0 * 90 * 128);
        for (int idx = tid; idx < nrows * 16; idx += Geo::THREADS) {
            const int r = idx >> 4, c = idx & 15;
            go[idx] = *reinterpret_cast<const uint4 *>(smem + lds_addr(r * CV_ROWB, c));
        }
    }
    if (head_out) {   // as in k_tower8_c128: one thread per board cell, all three head channels, fixed summation order
        const float *hw = reinterpret_cast<const float *>(smem + Geo::HEADW_OFF);
        for (int r = tid; r < nrows; r += Geo::THREADS) {
            const int rowoff = r * CV_ROWB, key = (rowoff >> 8) & 15;
            float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
#pragma unroll 4
            for (int c = 0; c < 16; ++c) {
                const int p = c ^ key;
                const uint4 v = *reinterpret_cast<const uint4 *>(smem + rowoff + (p << 4));
                const f32x2 e0 = unpack_pair<F16>(v.x), e1 = unpack_pair<F16>(v.y), e2 = unpack_pair<F16>(v.z), e3 = unpack_pair<F16>(v.w);
                const float *w0 = hw + c * 8, *w1 = hw + 128 + c * 8, *w2 = hw + 256 + c * 8;
                acc0 += e0[0] * w0[0] + e0[1] * w0[1] + e1[0] * w0[2] + e1[1] * w0[3]
                      + e2[0] * w0[4] + e2[1] * w0[5] + e3[0] * w0[6] + e3[1] * w0[7];
                acc1 += e0[0] * w1[0] + e0[1] * w1[1] + e1[0] * w1[2] + e1[1] * w1[3]
                      + e2[0] * w1[4] + e2[1] * w1[5] + e3[0] * w1[6] + e3[1] * w1[7];
                acc2 += e0[0] * w2[0] + e0[1] * w2[1] + e1[0] * w2[2] + e1[1] * w2[3]
                      + e2[0] * w2[4] + e2[1] * w2[5] + e3[0] * w2[6] + e3[1] * w2[7];
            }
            float *o = head_out + ((size_t)pos0 * 90 + r) * 3;
            o[0] = fmaxf(acc0 + head_b[0], 0.f);
            o[1] = fmaxf(acc1 + head_b[1], 0.f);
            o[2] = fmaxf(acc2 + head_b[2], 0.f);
        }
    }
}
#undef SK_SLAB
#undef SK_RUN

// =================================================================================================
// k_towerd_c128 ("d", opt-in: CCHESS_TOWER_VARIANT=d): k_tower8_c128 without the weight ring.  Every wave loads the two weight
// fragments of a k-step straight from global memory into registers (16 bytes per lane and fragment, 512 contiguous bytes per
// half-wave in the packed layout [layer][tap][8-channel chunk][128 out][8]), two k-steps ahead, three register sets in
// rotation; the four waves that share a channel half read the same lines within a few k-steps of each other, so all but the
// first find them in the CU's vector L1.  No LDS-DMA, no per-slab barrier (the only barriers left are the two per layer), the
// LDS pipe serves 3 instead of 5 fragment reads per 6 MFMAs, 64 KB of LDS stay free.  Same tiles, same k order, same rounding
// as k_tower8_c128: bit-identical outputs.  One asm macro per k-step (tools/gen_tower_asm.py, kstepD); the rotation period of
// the register sets is 24 k-steps = three taps.
// =================================================================================================
struct DGeo {
    static constexpr int P = 4, ROWS = 360, THREADS = 512;
    static constexpr int KSTEP_BYTES = 16 * 128 * 2, KSTEPS_PER_LAYER = 72;
    static constexpr int ZERO_OFF = ROWS * CV_ROWB;
    static constexpr int PLANES_OFF = ZERO_OFF + CV_ROWB;          // the input planes (32 B per cell), first layer only
    static constexpr int HEADW_OFF = PLANES_OFF + ROWS * 32;
    static constexpr int LDS_TOTAL = HEADW_OFF + 3 * 128 * 4;      // 105,472
};
constexpr int TD_P = 4, TD_THREADS = DGeo::THREADS, TD_LDS_BYTES = DGeo::LDS_TOTAL;
struct TdA { bf16x8 a[CV_RT]; };
struct TdW { bf16x8 b[CV_CT]; };

template <bool F16>
__global__ __launch_bounds__(512, 2) void k_towerd_c128(const uint16_t *__restrict__ in,
                                                        const uint16_t *__restrict__ wpk,
                                                        const float *__restrict__ bias,
                                                        uint16_t *__restrict__ out,
                                                        const float *__restrict__ head_w,
                                                        const float *__restrict__ head_b,
                                                        float *__restrict__ head_out,
                                                        const uint16_t *__restrict__ planes,
                                                        const uint16_t *__restrict__ w0,
                                                        const float *__restrict__ b0,
                                                        int B, int nlayers,
                                                        const int *__restrict__ bcount) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using Geo = DGeo;
    constexpr int P = Geo::P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int pos0 = blockIdx.x * P;
    if (bcount) {
        const int live = *bcount;
        B = live < B ? live : B;
    }
    if (pos0 >= B) return;
    const int npos = (B - pos0) < P ? (B - pos0) : P;
    const int nrows = npos * 90;
    const int total_k = nlayers * Geo::KSTEPS_PER_LAYER;
    auto lds_addr = [](int row_byte_off, int c) { return row_byte_off + ((c ^ ((row_byte_off >> 8) & 15)) << 4); };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // this lane's 16 bytes inside a k-step's 4 KB of packed weights: chunk khalf, output channel wc*64 + l31 (+32: offset 512)
    const unsigned voff = (unsigned)khalf * 2048u + (((unsigned)wc * 64u + (unsigned)l31) << 4);
    auto kstep_ptr = [&](int k) {
        return reinterpret_cast<const unsigned char *>(wpk) + (size_t)(k < total_k ? k : total_k - 1) * Geo::KSTEP_BYTES;
    };
    TdA A0, A1;
    TdW W0, W1, W2;
    {   // the weight pipeline is filled once: k-steps 0 and 1 (it never drains between layers: the layers are contiguous)
        const unsigned char *p0 = kstep_ptr(0), *p1 = kstep_ptr(1);
        asm volatile("global_load_dwordx4 %[w0b0], %[voff], %[p0]\n\t"
                     "global_load_dwordx4 %[w0b1], %[voff], %[p0] offset:512\n\t"
                     "global_load_dwordx4 %[w1b0], %[voff], %[p1]\n\t"
                     "global_load_dwordx4 %[w1b1], %[voff], %[p1] offset:512\n\t"
                     : [w0b0] "=&v"(W0.b[0]), [w0b1] "=&v"(W0.b[1]), [w1b0] "=&v"(W1.b[0]), [w1b1] "=&v"(W1.b[1])
                     : [voff] "v"(voff), [p0] "s"(p0), [p1] "s"(p1)
                     : "memory");
        W2.b[0] = W2.b[1] = bf16x8{}; A0.a[0] = A0.a[1] = A0.a[2] = bf16x8{}; A1 = A0;   // defined values for the read-write asm operands below
    }
    if (planes == nullptr) {
        const uint4 *g = reinterpret_cast<const uint4 *>(in + (size_t)pos0 * 90 * 128);
        for (int idx = tid; idx < Geo::ROWS * 16; idx += Geo::THREADS) {
            const int r = idx >> 4, c = idx & 15;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < nrows) v = g[idx];
            *reinterpret_cast<uint4 *>(smem + lds_addr(r * CV_ROWB, c)) = v;
        }
    } else {
        const uint4 *g = reinterpret_cast<const uint4 *>(planes + (size_t)pos0 * 90 * 16);
        for (int idx = tid; idx < Geo::ROWS * 2; idx += Geo::THREADS) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < nrows * 2) v = g[idx];
            *reinterpret_cast<uint4 *>(smem + Geo::PLANES_OFF + (idx << 4)) = v;
        }
    }
    if (tid < 16) *reinterpret_cast<uint4 *>(smem + Geo::ZERO_OFF + (tid << 4)) = make_uint4(0, 0, 0, 0);
    if (head_out && tid < 3 * 128 / 4)
        reinterpret_cast<float4 *>(smem + Geo::HEADW_OFF)[tid] = reinterpret_cast<const float4 *>(head_w)[tid];
    bf16x8 wf[9][CV_CT];
    if (planes != nullptr) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < CV_CT; ++j)
                wf[t][j] = *reinterpret_cast<const bf16x8 *>(w0 + ((size_t)((t * 2 + (lane >> 5)) * 128 + (wave & 1) * 64 + j * 32 + (lane & 31)) << 3));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int rowb[CV_RT], tapmask[CV_RT];
#pragma unroll
    for (int i = 0; i < CV_RT; ++i) {
        const int r = 32 * (wr * CV_RT + i) + l31;
        const int pix = r % 90, h = pix / 10, w = pix - h * 10;
        rowb[i] = r * CV_ROWB;
        int m = 0;
        for (int t = 0; t < 9; ++t) {
            const int y = h + t / 3 - 1, x = w + t % 3 - 1;
            if (r < Geo::ROWS && y >= 0 && y < 9 && x >= 0 && x < 10) m |= 1 << t;
        }
        tapmask[i] = m;
    }
    auto tap_addr = [&](int tap, int (&ab)[CV_RT], int (&key)[CV_RT]) {   // tap 9 (past the last): every lane reads the zero row
        const int delta = ((tap / 3 - 1) * 10 + (tap - (tap / 3) * 3 - 1)) * CV_ROWB;
#pragma unroll
        for (int i = 0; i < CV_RT; ++i) {
            const int a = ((tapmask[i] >> tap) & 1) ? rowb[i] + delta : Geo::ZERO_OFF;
            ab[i] = a;
            key[i] = (((rowb[i] + delta) >> 8) & 15) ^ khalf;   // see k_tower8_c128
        }
    };
    int rb[CV_RT];
    auto refresh_rb = [&]() {
#pragma unroll
        for (int i = 0; i < CV_RT; ++i) {
            const int r = 32 * (wr * CV_RT + i) + l31;
            rb[i] = (r < Geo::ROWS ? r : 0) * CV_ROWB;
            asm volatile("" : "+v"(rb[i]));
        }
    };
    auto cell_ptr = [&](int i, int j, int q, bool &live) -> uint2 * {
        live = 32 * (wr * CV_RT + i) + l31 < Geo::ROWS;
        const int n0 = wc * 64 + j * 32 + 8 * q + 4 * khalf;
        return reinterpret_cast<uint2 *>(smem + lds_addr(rb[i], n0 >> 3) + ((n0 & 4) << 1));
    };
    uint2 xreg[CV_RT][CV_CT][4];
    auto init_acc = [&](f32x16 (&acc)[CV_RT][CV_CT], const float *bl, bool add_x) {
#pragma unroll
        for (int j = 0; j < CV_CT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4 *>(bl + wc * 64 + j * 32 + 8 * q + 4 * khalf);
#pragma unroll
                for (int i = 0; i < CV_RT; ++i) {
                    float a0 = bq.x, a1 = bq.y, a2 = bq.z, a3 = bq.w;
                    if (add_x) {
                        const uint2 x = xreg[i][j][q];
                        const f32x2 xl = unpack_pair<F16>(x.x), xh = unpack_pair<F16>(x.y);
                        a0 += xl[0]; a1 += xl[1]; a2 += xh[0]; a3 += xh[1];
                    }
                    acc[i][j][4 * q + 0] = a0; acc[i][j][4 * q + 1] = a1; acc[i][j][4 * q + 2] = a2; acc[i][j][4 * q + 3] = a3;
                }
            }
    };
    auto store_layer = [&](f32x16 (&acc)[CV_RT][CV_CT]) {   // ReLU -> 16 bit -> U in place (see k_tower8_c128)
#pragma unroll
        for (int i = 0; i < CV_RT; ++i)
#pragma unroll
            for (int j = 0; j < CV_CT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bool live;
                    uint2 *cell = cell_ptr(i, j, q, live);
                    const s16x2 z = {0, 0};
                    const s16x2 rl = __builtin_elementwise_max(
                        __builtin_bit_cast(s16x2, pack_pair<F16>(f32x2{acc[i][j][4 * q + 0], acc[i][j][4 * q + 1]})), z);
                    const s16x2 rh = __builtin_elementwise_max(
                        __builtin_bit_cast(s16x2, pack_pair<F16>(f32x2{acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]})), z);
                    if (live) *cell = make_uint2(__builtin_bit_cast(uint32_t, rl), __builtin_bit_cast(uint32_t, rh));
                }
    };
    // workgroup barrier that leaves the weight loads in flight (__syncthreads() would wait for vmcnt(0))
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    if (planes != nullptr) {   // first layer: conv3x3(14 -> 128) + BN + ReLU, one k-step per tap
        f32x16 acc[CV_RT][CV_CT];
        init_acc(acc, b0, false);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int shift = (t / 3 - 1) * 10 + (t % 3 - 1);
            bf16x8 af[CV_RT];
#pragma unroll
            for (int i = 0; i < CV_RT; ++i) {
                const int a = ((tapmask[i] >> t) & 1) ? Geo::PLANES_OFF + ((rowb[i] >> 3) + shift * 32) : Geo::ZERO_OFF;
                af[i] = *reinterpret_cast<const bf16x8 *>(smem + a + khalf * 16);
            }
#pragma unroll
            for (int i = 0; i < CV_RT; ++i)
#pragma unroll
                for (int j = 0; j < CV_CT; ++j)
                    acc[i][j] = mfma_32x32x16<F16>(wf[t][j], af[i], acc[i][j]);
        }
        refresh_rb();
        store_layer(acc);
        lds_barrier();
    }

#define TD_STEP(ASMSTR, KOFF)                                                                                    \
        {                                                                                                        \
            const unsigned char *wn = kstep_ptr(gk + (KOFF) + 2);                                                \
            asm volatile(ASMSTR                                                                                  \
                : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]),      \
                  [c20] "+v"(acc[2][0]), [c21] "+v"(acc[2][1]),                                                    \
                  [A0a0] "+v"(A0.a[0]), [A0a1] "+v"(A0.a[1]), [A0a2] "+v"(A0.a[2]),                                 \
                  [A1a0] "+v"(A1.a[0]), [A1a1] "+v"(A1.a[1]), [A1a2] "+v"(A1.a[2]),                                 \
                  [W0b0] "+v"(W0.b[0]), [W0b1] "+v"(W0.b[1]), [W1b0] "+v"(W1.b[0]), [W1b1] "+v"(W1.b[1]),            \
                  [W2b0] "+v"(W2.b[0]), [W2b1] "+v"(W2.b[1]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)         \
                : [ab0] "v"(ab[0]), [ab1] "v"(ab[1]), [ab2] "v"(ab[2]), [key0] "v"(key[0]), [key1] "v"(key[1]),      \
                  [key2] "v"(key[2]), [nab0] "v"(nab[0]), [nab1] "v"(nab[1]), [nab2] "v"(nab[2]), [nkey0] "v"(nkey[0]), \
                  [nkey1] "v"(nkey[1]), [nkey2] "v"(nkey[2]), [voff] "v"(voff), [wn] "s"(wn)                          \
                : "memory");                                                                                     \
        }
#define TD_K(K) if constexpr (F16) { TD_STEP(TWDF_KSTEP_##K, K) } else { TD_STEP(TWD_KSTEP_##K, K) }
    int gk = 0;   // global k-step index of the current group of three taps
#pragma unroll 1
    for (int layer = 0; layer < nlayers; ++layer) {
        f32x16 acc[CV_RT][CV_CT];
        CZ_T8_STAMP(0);
        refresh_rb();
        if (!(layer & 1)) {   // first conv of a block: remember x, start from the bias
#pragma unroll
            for (int i = 0; i < CV_RT; ++i)
#pragma unroll
                for (int j = 0; j < CV_CT; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { bool live; xreg[i][j][q] = *cell_ptr(i, j, q, live); }
            init_acc(acc, bias + layer * 128, false);
        } else {
            init_acc(acc, bias + layer * 128, true);
        }
        int ab[CV_RT], key[CV_RT], nab[CV_RT], nkey[CV_RT], t0, t1, t2;
        tap_addr(0, ab, key);
        asm volatile(   // the activation fragments of the layer's first k-step
            "v_xor_b32 %[t0], 0, %[k0]\n\t"
            "v_xor_b32 %[t1], 0, %[k1]\n\t"
            "v_xor_b32 %[t2], 0, %[k2]\n\t"
            "v_lshl_add_u32 %[t0], %[t0], 4, %[b0]\n\t"
            "v_lshl_add_u32 %[t1], %[t1], 4, %[b1]\n\t"
            "v_lshl_add_u32 %[t2], %[t2], 4, %[b2]\n\t"
            "ds_read_b128 %[xa0], %[t0]\n\t"
            "ds_read_b128 %[xa1], %[t1]\n\t"
            "ds_read_b128 %[xa2], %[t2]\n\t"
            : [xa0] "=&v"(A0.a[0]), [xa1] "=&v"(A0.a[1]), [xa2] "=&v"(A0.a[2]), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
            : [k0] "v"(key[0]), [k1] "v"(key[1]), [k2] "v"(key[2]), [b0] "v"(ab[0]), [b1] "v"(ab[1]), [b2] "v"(ab[2])
            : "memory");
        CZ_T8_STAMP(1);
#pragma unroll 1
        for (int t3 = 0; t3 < 3; ++t3) {   // three taps = 24 k-steps = one period of the register rotation
            tap_addr(3 * t3 + 1, nab, nkey);
            TD_K(0) TD_K(1) TD_K(2) TD_K(3) TD_K(4) TD_K(5) TD_K(6) TD_K(7)
#pragma unroll
            for (int i = 0; i < CV_RT; ++i) { ab[i] = nab[i]; key[i] = nkey[i]; }
            tap_addr(3 * t3 + 2, nab, nkey);
            TD_K(8) TD_K(9) TD_K(10) TD_K(11) TD_K(12) TD_K(13) TD_K(14) TD_K(15)
#pragma unroll
            for (int i = 0; i < CV_RT; ++i) { ab[i] = nab[i]; key[i] = nkey[i]; }
            tap_addr(3 * t3 + 3, nab, nkey);
            TD_K(16) TD_K(17) TD_K(18) TD_K(19) TD_K(20) TD_K(21) TD_K(22) TD_K(23)
#pragma unroll
            for (int i = 0; i < CV_RT; ++i) { ab[i] = nab[i]; key[i] = nkey[i]; }
            gk += 24;
        }
        // the last k-step requested activation fragments for a tap that does not exist: drain them, let the MFMAs retire
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        CZ_T8_STAMP(2);
        lds_barrier();        // every wave is done reading U before anyone overwrites it in place
#if defined(CZ_T8_TRACE)
        CZ_T8_STAMP(13);
#endif
        refresh_rb();
        store_layer(acc);
        lds_barrier();
        CZ_T8_STAMP(3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the weight fragments requested past the last k-step
    if (out) {
        uint4 *go = reinterpret_cast<uint4 *>(out + (size_t)pos0 * 90 * 128);
        for (int idx = tid; idx < nrows * 16; idx += Geo::THREADS) {
            const int r = idx >> 4, c = idx & 15;
            go[idx] = *reinterpret_cast<const uint4 *>(smem + lds_addr(r * CV_ROWB, c));
        }
    }
    if (head_out) {   // as in k_tower8_c128
        const float *hw = reinterpret_cast<const float *>(smem + Geo::HEADW_OFF);
        for (int r = tid; r < nrows; r += Geo::THREADS) {
            const int rowoff = r * CV_ROWB, key = (rowoff >> 8) & 15;
            float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
#pragma unroll 4
            for (int c = 0; c < 16; ++c) {
                const int p = c ^ key;
                const uint4 v = *reinterpret_cast<const uint4 *>(smem + rowoff + (p << 4));
                const f32x2 e0 = unpack_pair<F16>(v.x), e1 = unpack_pair<F16>(v.y), e2 = unpack_pair<F16>(v.z), e3 = unpack_pair<F16>(v.w);
                const float *w0 = hw + c * 8, *w1 = hw + 128 + c * 8, *w2 = hw + 256 + c * 8;
                acc0 += e0[0] * w0[0] + e0[1] * w0[1] + e1[0] * w0[2] + e1[1] * w0[3]
                      + e2[0] * w0[4] + e2[1] * w0[5] + e3[0] * w0[6] + e3[1] * w0[7];
                acc1 += e0[0] * w1[0] + e0[1] * w1[1] + e1[0] * w1[2] + e1[1] * w1[3]
                      + e2[0] * w1[4] + e2[1] * w1[5] + e3[0] * w1[6] + e3[1] * w1[7];
                acc2 += e0[0] * w2[0] + e0[1] * w2[1] + e1[0] * w2[2] + e1[1] * w2[3]
                      + e2[0] * w2[4] + e2[1] * w2[5] + e3[0] * w2[6] + e3[1] * w2[7];
            }
            float *o = head_out + ((size_t)pos0 * 90 + r) * 3;
            o[0] = fmaxf(acc0 + head_b[0], 0.f);
            o[1] = fmaxf(acc1 + head_b[1], 0.f);
            o[2] = fmaxf(acc2 + head_b[2], 0.f);
        }
    }
}
#undef TD_STEP
#undef TD_K

// =================================================================================================
// k_towerp_c128: the one-launch net trunk with ONE POSITION PER WAVE (four waves, four positions per workgroup).
//
// k_tower8_c128 is co-limited by LDS bandwidth: with 3 cell tiles x 2 channel tiles per wave every 6 MFMAs need
// 5 ds_read_b128, 40 KB per k-step and CU against 256 B/clk, which is as long as the MFMAs take.  Here a wave
// owns ALL 128 output channels of one position (3 cell tiles of 32, 90 live cells): 12 MFMAs per 7 fragment
// reads, 28 KB per k-step and CU.  The 12 accumulators (192 registers) live in AGPRs, the block input x (96
// packed registers) and two fragment sets in VGPRs; one wave per SIMD, latency hidden by software pipelining
// (the next k-step's fragments are requested one per MFMA gap of the current one, see tools/gen_tower_asm.py).
// A wave only ever reads and writes the LDS rows of its own position, so a layer is written back in place with
// no workgroup barrier at all; the only barrier left is the one per slab that publishes the weight ring.
// =================================================================================================
constexpr int TP_P = 4;
constexpr int TP_THREADS = 256;
constexpr int TP_CT = 4;                            // channel tiles per wave: all of them

#define TP_LOADSET(CA, OB, XA, XB, AB, KEY, VB)                                                      \
    asm volatile(                                                                                    \
        "v_xor_b32 %[t0], " #CA ", %[k0]\n\t"                                                        \
        "v_xor_b32 %[t1], " #CA ", %[k1]\n\t"                                                        \
        "v_xor_b32 %[t2], " #CA ", %[k2]\n\t"                                                        \
        "v_lshl_add_u32 %[t0], %[t0], 4, %[b0]\n\t"                                                  \
        "v_lshl_add_u32 %[t1], %[t1], 4, %[b1]\n\t"                                                  \
        "v_lshl_add_u32 %[t2], %[t2], 4, %[b2]\n\t"                                                  \
        "ds_read_b128 %[xa0], %[t0]\n\t"                                                             \
        "ds_read_b128 %[xa1], %[t1]\n\t"                                                             \
        "ds_read_b128 %[xa2], %[t2]\n\t"                                                             \
        "ds_read_b128 %[xb0], %[vb] offset:" #OB "\n\t"                                              \
        "ds_read_b128 %[xb1], %[vb] offset:" #OB "+512\n\t"                                          \
        "ds_read_b128 %[xb2], %[vb] offset:" #OB "+1024\n\t"                                         \
        "ds_read_b128 %[xb3], %[vb] offset:" #OB "+1536\n\t"                                         \
        : [xa0] "=&v"(XA[0]), [xa1] "=&v"(XA[1]), [xa2] "=&v"(XA[2]), [xb0] "=&v"(XB[0]),               \
          [xb1] "=&v"(XB[1]), [xb2] "=&v"(XB[2]), [xb3] "=&v"(XB[3]), [t0] "=&v"(t0), [t1] "=&v"(t1),   \
          [t2] "=&v"(t2)                                                                              \
        : [k0] "v"(KEY[0]), [k1] "v"(KEY[1]), [k2] "v"(KEY[2]), [b0] "v"(AB[0]), [b1] "v"(AB[1]),      \
          [b2] "v"(AB[2]), [vb] "v"(VB)                                                               \
        : "memory")

__global__ __launch_bounds__(TP_THREADS, 1) void k_towerp_c128(const uint16_t *__restrict__ in,
                                                               const uint16_t *__restrict__ wpk,
                                                               const float *__restrict__ bias,
                                                               uint16_t *__restrict__ out,
                                                               const float *__restrict__ head_w,
                                                               const float *__restrict__ head_b,
                                                               float *__restrict__ head_out,
                                                               const uint16_t *__restrict__ planes,
                                                               const uint16_t *__restrict__ w0,
                                                               const float *__restrict__ b0,
                                                               int B, int nlayers) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, khalf = lane >> 5;
    const int pos0 = blockIdx.x * TP_P;
    const int npos = (B - pos0) < TP_P ? (B - pos0) : TP_P;
    const int nrows = npos * 90;
    const int nslabs = nlayers * 18;
    auto lds_addr = [](int row_byte_off, int c) { return row_byte_off + ((c ^ ((row_byte_off >> 8) & 15)) << 4); };
    const unsigned voff0 = (unsigned)tid << 4;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool wave_live = wave_u < npos;   // a dead wave (batch tail) reads the zero row and stores nothing

    auto dma_slab = [&](int slab) {   // prologue only; the loop issues its DMAs from the slab asm
        const unsigned char *src = reinterpret_cast<const unsigned char *>(wpk) + (size_t)slab * TW_SLAB_BYTES;
        unsigned char *dst = smem + T8_W_OFF + ((unsigned)slab & 3u) * TW_SLAB_BYTES + (wave_u << 10);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + q * 4096 + voff0),
                                             (__attribute__((address_space(3))) void *)(dst + q * 4096), 16, 0, 0);
    };
    for (int q = 0; q < 3; ++q) dma_slab(q < nslabs ? q : nslabs - 1);
    if (planes == nullptr) {
        const uint4 *g = reinterpret_cast<const uint4 *>(in + (size_t)pos0 * 90 * 128);
        for (int idx = tid; idx < T8_ROWS * 16; idx += TP_THREADS) {
            const int r = idx >> 4, c = idx & 15;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < nrows) v = g[idx];
            *reinterpret_cast<uint4 *>(smem + lds_addr(r * CV_ROWB, c)) = v;
        }
    } else {
        const uint4 *g = reinterpret_cast<const uint4 *>(planes + (size_t)pos0 * 90 * 16);
        for (int idx = tid; idx < T8_ROWS * 2; idx += TP_THREADS) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (idx < nrows * 2) v = g[idx];
            *reinterpret_cast<uint4 *>(smem + T8_PLANES_OFF + (idx << 4)) = v;
        }
    }
    if (tid < 16) *reinterpret_cast<uint4 *>(smem + T8_ZERO_OFF + (tid << 4)) = make_uint4(0, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // this lane's three cells (tile i: cell 32*i + l31 of position `wave`; cells >= 90 are padding)
    int rowb[CV_RT], tapmask[CV_RT];
#pragma unroll
    for (int i = 0; i < CV_RT; ++i) {
        const int pix = 32 * i + l31, h = pix / 10, w = pix - h * 10;
        rowb[i] = (wave * 90 + (pix < 90 ? pix : 0)) * CV_ROWB;
        int m = 0;
        for (int t = 0; t < 9; ++t) {
            const int y = h + t / 3 - 1, x = w + t % 3 - 1;
            if (wave_live && pix < 90 && y >= 0 && y < 9 && x >= 0 && x < 10) m |= 1 << t;
        }
        tapmask[i] = m;
    }
    auto tap_addr = [&](int tap, int (&ab)[CV_RT], int (&key)[CV_RT]) {
        const int delta = ((tap / 3 - 1) * 10 + (tap - (tap / 3) * 3 - 1)) * CV_ROWB;
#pragma unroll
        for (int i = 0; i < CV_RT; ++i) {
            const int a = ((tapmask[i] >> tap) & 1) ? rowb[i] + delta : T8_ZERO_OFF;
            ab[i] = a;
            // a masked lane reads the all-zero row, but in the 16-byte slot its REAL (off-board) neighbour row would have
            // used: the 16 lanes of a ds_read_b128 group then still hit 16 distinct slots.  With the zero row's own swizzle
            // key every group containing a border cell paid a 2-way bank conflict (24 % of the LDS cycles).
            key[i] = (((rowb[i] + delta) >> 8) & 15) ^ khalf;
        }
    };
    const int vb0 = T8_W_OFF + khalf * 2048 + (l31 << 4);
    int keep;

    // 48 swizzled addresses per lane: recomputed from an opaque copy of the row offset wherever they are needed
    // (hoisted out of the layer loop they would only be spilled)
    int rb[CV_RT];
    auto refresh_rb = [&]() {
#pragma unroll
        for (int i = 0; i < CV_RT; ++i) { rb[i] = rowb[i]; asm volatile("" : "+v"(rb[i])); }
    };
    auto cell_ptr = [&](int i, int j, int q, bool &live) -> uint2 * {
        live = wave_live && (32 * i + l31) < 90;
        const int n0 = j * 32 + 8 * q + 4 * khalf;
        return reinterpret_cast<uint2 *>(smem + lds_addr(rb[i], n0 >> 3) + ((n0 & 4) << 1));
    };
    uint2 xreg[CV_RT][TP_CT][4];   // block input x at this lane's accumulator positions (packed bf16)
    // Layer epilogue, one accumulator tile at a time out of the AGPRs: + bias (+ x) in packed fp32, -> bf16
    // (v_cvt_pk_bf16_f32, RNE), ReLU as a packed signed-16-bit max with 0 (a bf16 is negative exactly when its bit
    // pattern is a negative int16; rounding never changes the sign, so relu(round(v)) == round(relu(v))), then
    // 8 bytes per lane into this wave's own rows of U, in place.
    auto finish_layer = [&](f32x16 (&acc)[CV_RT][TP_CT], const float *bl, bool add_x) {
#pragma unroll
        for (int j = 0; j < TP_CT; ++j) {
            f32x2 bq[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = *reinterpret_cast<const float4 *>(bl + j * 32 + 8 * q + 4 * khalf);
                bq[q][0] = f32x2{b4.x, b4.y};
                bq[q][1] = f32x2{b4.z, b4.w};
            }
#pragma unroll
            for (int i = 0; i < CV_RT; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // explicit, ordered AGPR reads: left to the scheduler all 192 are hoisted to the top of the
                    // epilogue and the block input x gets spilled to make room
                    float a[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a[e]) : "a"(acc[i][j][4 * q + e]));
                    f32x2 lo = f32x2{a[0], a[1]} + bq[q][0];
                    f32x2 hi = f32x2{a[2], a[3]} + bq[q][1];
                    if (add_x) {
                        const uint2 x = xreg[i][j][q];
                        lo += f32x2{__uint_as_float(x.x << 16), __uint_as_float(x.x & 0xFFFF0000u)};
                        hi += f32x2{__uint_as_float(x.y << 16), __uint_as_float(x.y & 0xFFFF0000u)};
                    }
                    const s16x2 z = {0, 0};
                    const s16x2 rl = __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(lo, bf16x2)), z);
                    const s16x2 rh = __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(hi, bf16x2)), z);
                    bool live;
                    uint2 *cell = cell_ptr(i, j, q, live);
                    if (live) *cell = make_uint2(__builtin_bit_cast(uint32_t, rl), __builtin_bit_cast(uint32_t, rh));
                }
            }
        }
    };

    if (planes != nullptr) {   // first layer: conv3x3(14 -> 128) + BN + ReLU, one k-step per tap
        f32x16 acc[CV_RT][TP_CT];
#pragma unroll
        for (int i = 0; i < CV_RT; ++i)
#pragma unroll
            for (int j = 0; j < TP_CT; ++j) acc[i][j] = f32x16{};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int shift = (t / 3 - 1) * 10 + (t % 3 - 1);
            bf16x8 wf[TP_CT], af[CV_RT];
#pragma unroll
            for (int j = 0; j < TP_CT; ++j)
                wf[j] = *reinterpret_cast<const bf16x8 *>(w0 + ((size_t)((t * 2 + khalf) * 128 + j * 32 + l31) << 3));
#pragma unroll
            for (int i = 0; i < CV_RT; ++i) {
                const int a = ((tapmask[i] >> t) & 1) ? T8_PLANES_OFF + ((rowb[i] >> 3) + shift * 32) : T8_ZERO_OFF;
                af[i] = *reinterpret_cast<const bf16x8 *>(smem + a + khalf * 16);
            }
#pragma unroll
            for (int i = 0; i < CV_RT; ++i)
#pragma unroll
                for (int j = 0; j < TP_CT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
        refresh_rb();
        finish_layer(acc, b0, false);
        __syncthreads();       // ring buffer 3 (the planes) is about to be overwritten by the DMA of slab 3
    }

#define TP_SLAB(ASMSTR, NAB, NKEY, AC)                                                                           \
        asm volatile(ASMSTR                                                                                      \
            : [p00] AC(acc[0][0]), [p01] AC(acc[0][1]), [p02] AC(acc[0][2]), [p03] AC(acc[0][3]),                  \
              [p10] AC(acc[1][0]), [p11] AC(acc[1][1]), [p12] AC(acc[1][2]), [p13] AC(acc[1][3]),                  \
              [p20] AC(acc[2][0]), [p21] AC(acc[2][1]), [p22] AC(acc[2][2]), [p23] AC(acc[2][3]),                  \
              [f0a0] "+v"(fa0[0]), [f0a1] "+v"(fa0[1]), [f0a2] "+v"(fa0[2]), [f0b0] "+v"(fb0[0]),                  \
              [f0b1] "+v"(fb0[1]), [f0b2] "+v"(fb0[2]), [f0b3] "+v"(fb0[3]),                                       \
              [f1a0] "=&v"(fa1[0]), [f1a1] "=&v"(fa1[1]), [f1a2] "=&v"(fa1[2]), [f1b0] "=&v"(fb1[0]),              \
              [f1b1] "=&v"(fb1[1]), [f1b2] "=&v"(fb1[2]), [f1b3] "=&v"(fb1[3]),                                    \
              [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [keep] "=&s"(keep)                                   \
            : [ab0] "v"(ab[0]), [ab1] "v"(ab[1]), [ab2] "v"(ab[2]), [key0] "v"(key[0]), [key1] "v"(key[1]),        \
              [key2] "v"(key[2]), [nab0] "v"(NAB[0]), [nab1] "v"(NAB[1]), [nab2] "v"(NAB[2]), [nkey0] "v"(NKEY[0]), \
              [nkey1] "v"(NKEY[1]), [nkey2] "v"(NKEY[2]), [vb] "v"(vb), [vbn] "v"(vbn), [voff0] "v"(voff0),        \
              [sbase] "s"(sbase), [sbase1] "s"(sbase + 4096), [sbase2] "s"(sbase + 8192),                          \
              [sbase3] "s"(sbase + 12288), [ldst] "s"(ldst)                                                        \
            : "memory", "scc")
#define TP_ACC_RW "+a"
#define TP_ACC_W "=&a"   /* first slab of a layer: the accumulators start from the MFMA's inline 0 */
#define TP_SLAB_ARGS()                                                                                          \
        const int vb = vb0 + (((unsigned)g & 3u) << 14), vbn = vb0 + ((((unsigned)g + 1u) & 3u) << 14);         \
        const int gn = g + 3 < nslabs ? g + 3 : nslabs - 1;                                                     \
        const unsigned char *sbase = reinterpret_cast<const unsigned char *>(wpk) + (size_t)gn * TW_SLAB_BYTES;  \
        const int ldst = T8_W_OFF + ((((unsigned)g + 3u) & 3u) << 14) + (wave_u << 10);

    int g = 0;
#pragma unroll 1
    for (int layer = 0; layer < nlayers; ++layer) {
        f32x16 acc[CV_RT][TP_CT];
        if (!(layer & 1)) {   // first conv of a block: remember x for the residual add of the second
            refresh_rb();
#pragma unroll
            for (int i = 0; i < CV_RT; ++i)
#pragma unroll
                for (int j = 0; j < TP_CT; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { bool live; xreg[i][j][q] = *cell_ptr(i, j, q, live); }
        }
        int ab[CV_RT], key[CV_RT], nab[CV_RT], nkey[CV_RT], t0, t1, t2;
        bf16x8 fa0[CV_RT], fb0[TP_CT], fa1[CV_RT], fb1[TP_CT];   // two fragment sets (cells a, weights b)
        tap_addr(0, ab, key);
        {
            const int vb = vb0 + (((unsigned)g & 3u) << 14);
            TP_LOADSET(0, 0, fa0, fb0, ab, key, vb);   // waited for by the first k-step itself
        }
        {
            TP_SLAB_ARGS()
            TP_SLAB(TWP_SLAB_ASM_FIRST, ab, key, TP_ACC_W);
            ++g;
        }
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            {
                tap_addr(tap + 1, nab, nkey);
                TP_SLAB_ARGS()
                TP_SLAB(TWP_SLAB_ASM_H1, nab, nkey, TP_ACC_RW);
#pragma unroll
                for (int i = 0; i < CV_RT; ++i) { ab[i] = nab[i]; key[i] = nkey[i]; }
                ++g;
            }
            if (tap < 8) {
                TP_SLAB_ARGS()
                TP_SLAB(TWP_SLAB_ASM_H0, ab, key, TP_ACC_RW);
                ++g;
            }
        }
        // drain the garbage prefetch of the non-existent next slab and let the last MFMAs retire; the rows this
        // wave overwrites are read by no other wave, so no barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
        refresh_rb();
        finish_layer(acc, bias + layer * 128, (layer & 1) != 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (out) {
        uint4 *go = reinterpret_cast<uint4 *>(out + (size_t)pos0 * 90 * 128);
        for (int idx = tid; idx < nrows * 16; idx += TP_THREADS) {
            const int r = idx >> 4, c = idx & 15;
            go[idx] = *reinterpret_cast<const uint4 *>(smem + lds_addr(r * CV_ROWB, c));
        }
    }
    if (head_out) {
        float *hw = reinterpret_cast<float *>(smem + T8_W_OFF);
        for (int i = tid; i < 3 * 128; i += TP_THREADS) hw[i] = head_w[i];
        __syncthreads();
        for (int idx = tid; idx < nrows * 3; idx += TP_THREADS) {
            const int r = idx / 3, c3 = idx - r * 3;
            const int rowoff = r * CV_ROWB, key = (rowoff >> 8) & 15;
            const float *w = hw + c3 * 128;
            float acc = 0.f;
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int c = it;                   // fixed summation order: the result of a position does not depend on its row
                const int p = c ^ key;
                const uint4 v = *reinterpret_cast<const uint4 *>(smem + rowoff + (p << 4));
                const float *wc8 = w + c * 8;
                acc += __uint_as_float(v.x << 16) * wc8[0] + __uint_as_float(v.x & 0xFFFF0000u) * wc8[1]
                     + __uint_as_float(v.y << 16) * wc8[2] + __uint_as_float(v.y & 0xFFFF0000u) * wc8[3]
                     + __uint_as_float(v.z << 16) * wc8[4] + __uint_as_float(v.z & 0xFFFF0000u) * wc8[5]
                     + __uint_as_float(v.w << 16) * wc8[6] + __uint_as_float(v.w & 0xFFFF0000u) * wc8[7];
            }
            head_out[((size_t)pos0 * 90 + r) * 3 + c3] = fmaxf(acc + head_b[c3], 0.f);
        }
    }
}
#undef TP_SLAB
#undef TP_SLAB_ARGS
#undef TP_LOADSET
#undef TP_ACC_RW
#undef TP_ACC_W

#undef TW_SLAB
#undef TW_SLAB_ARGS
#undef TW_LOADSET
}  // namespace czconv
