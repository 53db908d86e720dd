// tools/experiments/k_movegen_lane_piece.h — the ordered-list kernel of rounds 2-4a (lane = (position, piece), four positions per
// wave: czd_group_movegen of cz_device.h, which the search kernels still use for their one position per wave).  Replaced in the
// library by k_movegen_list (one lane = one position, czm_list of cz_maskgen.h): 1.95 -> 4.74 G positions/s for the list alone.
// Kept as the record of the design; it compiled inside cz_rules.hip's anonymous namespace (needs cz_device.h, load helpers).
#pragma once
// K1: FOUR positions per wave64 (czd_group_movegen: lane = (position, piece of the side to move)); boards come in with
// 2-byte loads (the ABI only promises byte alignment), the ordered lists leave as one 16-byte store per lane.
__global__ __launch_bounds__(64) void k_movegen(CzTables tab, const uint8_t *__restrict__ boards,
                                                const uint8_t *__restrict__ side, int G,
                                                uint16_t *__restrict__ moves, uint16_t *__restrict__ count,
                                                uint32_t *__restrict__ mask) {
    __shared__ __attribute__((aligned(16))) uint8_t b[4 * CZ_NSQ + 8];   // four boards, packed (stride 90)
    __shared__ uint16_t stage[64 * CZD_STAGE_STRIDE];
    __shared__ __attribute__((aligned(16))) uint16_t out[4 * CZD_MAXMOVES];
    __shared__ CzdGroupLds GL;
    __shared__ uint32_t m[4 * (CZ_MASK_WORDS + 2)];
    __shared__ uint8_t sd[4];
    const int lane = threadIdx.x, q = lane >> 4, s = lane & 15;
    const int ngroups = (G + 3) >> 2;
    const bool aligned4 = (reinterpret_cast<uintptr_t>(boards) & 3u) == 0;   // 4 boards = 360 bytes = 90 dwords
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int g0 = grp * 4;
        const int np = min(4, G - g0);
        if (aligned4 && np == 4) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(boards + (size_t)g0 * CZ_NSQ);
            reinterpret_cast<uint32_t *>(b)[lane] = src[lane];
            if (lane < 26) reinterpret_cast<uint32_t *>(b)[lane + 64] = src[lane + 64];
        } else {
            for (int j = lane; j < 4 * CZ_NSQ; j += 64) {
                const int p = j / CZ_NSQ;
                b[j] = p < np ? boards[(size_t)g0 * CZ_NSQ + j] : (uint8_t)0;
            }
        }
        if (lane < 4) sd[lane] = (lane < np && side[g0 + lane]) ? 1 : 0;
        __syncthreads();
        const int n = czd_group_movegen<4, CZ_NSQ>(b, [&](int p) { return (int)sd[p]; }, tab.lut, GL, stage, out, lane);
        const int nn = n < 0 ? 0 : n;
        if (s == 0 && q < np) count[g0 + q] = n < 0 ? (uint16_t)0xFFFF : (uint16_t)n;
        if (moves) {
            for (int i = s; i < CZD_MAXMOVES; i += 16)
                if (i >= nn) out[q * CZD_MAXMOVES + i] = (uint16_t)0xFFFF;
            __syncthreads();
            if (q < np) reinterpret_cast<uint4 *>(moves + (size_t)g0 * CZD_MAXMOVES)[lane] = reinterpret_cast<const uint4 *>(out)[lane];
        }
        if (mask) {
            for (int i = lane; i < 4 * (CZ_MASK_WORDS + 2); i += 64) m[i] = 0;
            __syncthreads();
            for (int i = s; i < nn; i += 16) {
                const int l = out[q * CZD_MAXMOVES + i];
                atomicOr(&m[q * (CZ_MASK_WORDS + 2) + (l >> 5)], 1u << (l & 31));
            }
            __syncthreads();
            for (int i = lane; i < np * CZ_MASK_WORDS; i += 64) {
                const int p = i / CZ_MASK_WORDS, w = i - p * CZ_MASK_WORDS;
                mask[(size_t)g0 * CZ_MASK_WORDS + i] = m[p * (CZ_MASK_WORDS + 2) + w];
            }
        }
        __syncthreads();
    }
}

