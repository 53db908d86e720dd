// cz_rules.hip — stand-alone rules kernels K1 (move generation: four positions per wave64), K2 (make move + hash +
// flags), K3 (input planes: one wave per position).  Boards are staged in LDS, results leave with coalesced stores.  These are HBM/issue-bound byte
// kernels: no MFMA here by design.
#include "cz_internal.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ void load_board(const uint8_t *__restrict__ g, uint8_t *lds, int lane) {
    // 90 bytes: lanes 0..44 move 2 bytes each (boards are only byte-aligned in the ABI)
    if (lane < 45) {
        lds[2 * lane] = g[2 * lane];
        lds[2 * lane + 1] = g[2 * lane + 1];
    } else if (lane < 48) {
        lds[2 * lane] = 0;
        lds[2 * lane + 1] = 0;
    }
    __syncthreads();
}

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

// K1, thread per position (round 3 EXPERIMENT, not the default: measured 0.89 G positions/s with the mask / 1.28 list only
// against k_movegen's 1.68 / 1.96 on the same box).  It executes 3.4x fewer instructions per position, as designed, but needs
// 25 KB of LDS per wave (ordered lists, boards): 6 waves per CU instead of 32, and a lane's long dependent chains (the
// generator's LDS round trips, ~28 cycles per instruction observed) are no longer hidden by anybody.  Kept selectable
// (CCHESS_MOVEGEN=lane) and under the same golden tests as the evidence behind DESIGN.md 4.5.
// k_movegen above spends 249 VALU + 156 SALU instructions per position, most of them on
// per-position wave-wide work (14 ballots for the occupancy sets and the piece list, the staging copy, three LDS passes
// for labels / mask / padding) — it is issue-bound at 7 % of its HBM roofline.  Here a LANE owns a position: 64 positions
// per wave, the reference's scan order (main.py:754-755) falls out of the lane's own loop over its <= 16 pieces in ascending
// square order, and the per-position overhead becomes per-lane work done for 64 positions at once:
//   P0  the lane reads its 90 board bytes (2-byte loads), keeps a copy in LDS (piece codes by square) and builds the four
//       90-bit sets (rank-major / file-major occupancy and black pieces) with constant shifts in a fully unrolled loop;
//   P1  per piece the same branch-free generator as everywhere else (czd_gen_piece_bf) appends (src, dst) pairs straight to
//       the lane's row of the LDS list — no staging, no prefix sums; then the flying general (main.py:1097-1107);
//   P2  two positions per pass (a half-wave each): (src, dst) -> label through the LUT, legality-mask bits by LDS atomics
//       into 8 staging rows that leave as one contiguous 2 112-byte block; the lists leave with 16-byte stores.
// LDS: 146 u16 per list row (128 + the generator's dump slot at +17), 92 bytes per board: 27 KB per wave, 5 waves per CU.
#define TPK_LSTRIDE 146
#define TPK_BSTRIDE 92
template <bool WANT_MASK>
__global__ __launch_bounds__(64) void k_movegen_tp(CzTables tab, const uint8_t *__restrict__ boards,
                                                   const uint8_t *__restrict__ side, int G,
                                                   uint16_t *__restrict__ moves, uint16_t *__restrict__ count,
                                                   uint32_t *__restrict__ mask) {
    __shared__ __attribute__((aligned(16))) uint16_t list[64 * TPK_LSTRIDE];
    __shared__ __attribute__((aligned(16))) uint8_t B[64 * TPK_BSTRIDE];
    __shared__ uint32_t leap[64];
    __shared__ uint32_t mrow[8 * CZ_MASK_WORDS];
    __shared__ int cnt[64];
    const int lane = threadIdx.x;
    leap[lane] = (&c_czd_leap[0][0])[lane];
    const int ngroups = (G + 63) >> 6;
    const bool al2 = (reinterpret_cast<uintptr_t>(boards) & 1u) == 0;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int g0 = grp * 64, p = g0 + lane;
        const bool live = p < G;
        const int sd = (live && side[p]) ? 1 : 0;
        const uint8_t *bp = boards + (size_t)(live ? p : g0) * CZ_NSQ;
        uint8_t *Bl = B + lane * TPK_BSTRIDE;
        uint16_t *row = list + lane * TPK_LSTRIDE;
        // the list row starts as padding (0xFFFF): what the generator does not overwrite is the ABI's tail
#pragma unroll
        for (int k = 0; k < TPK_LSTRIDE / 2; ++k) reinterpret_cast<uint32_t *>(row)[k] = 0xFFFFFFFFu;
        if (WANT_MASK)
            for (int i = lane; i < 8 * CZ_MASK_WORDS; i += 64) mrow[i] = 0u;
        // ---- P0: board -> LDS + the four sets
        uint32_t occ[3] = {0u, 0u, 0u}, blk[3] = {0u, 0u, 0u}, occT[3] = {0u, 0u, 0u}, blkT[3] = {0u, 0u, 0u};
        int Ksq = -1, ksq = -1;
#pragma unroll
        for (int k = 0; k < CZ_NSQ / 2; ++k) {
            unsigned v = 0;
            if (live) v = al2 ? (unsigned)reinterpret_cast<const uint16_t *>(bp)[k] : ((unsigned)bp[2 * k] | ((unsigned)bp[2 * k + 1] << 8));
            *reinterpret_cast<uint16_t *>(Bl + 2 * k) = (uint16_t)v;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                constexpr int dummy = 0; (void)dummy;
                const int sq = 2 * k + h;
                const unsigned c = (v >> (8 * h)) & 0xFFu;
                const unsigned nz = c ? 1u : 0u, bk = c >> 3;          // codes 8..14 are black
                const int tq = (sq % 9) * 10 + sq / 9;                   // file-major index of the square
                occ[sq >> 5] |= nz << (sq & 31);   blk[sq >> 5] |= bk << (sq & 31);
                occT[tq >> 5] |= nz << (tq & 31);  blkT[tq >> 5] |= bk << (tq & 31);
                Ksq = c == 1u ? sq : Ksq;
                ksq = c == 8u ? sq : ksq;
            }
        }
        CzdBoardSets S;
        S.occ.lo = occ[0] | ((unsigned long long)occ[1] << 32);   S.occ.hi = occ[2];
        S.occT.lo = occT[0] | ((unsigned long long)occT[1] << 32); S.occT.hi = occT[2];
        CzdSet bl, blT;
        bl.lo = blk[0] | ((unsigned long long)blk[1] << 32);   bl.hi = blk[2];
        blT.lo = blkT[0] | ((unsigned long long)blkT[1] << 32); blT.hi = blkT[2];
        S.enemy = sd ? czd_andn(S.occ, bl) : bl;
        S.enemyT = sd ? czd_andn(S.occT, blT) : blT;
        uint32_t m0 = sd ? blk[0] : occ[0] & ~blk[0], m1 = sd ? blk[1] : occ[1] & ~blk[1], m2 = sd ? blk[2] : occ[2] & ~blk[2];
        __syncthreads();   // leap[] / mrow[] initialised (first group) ; B is lane-private
        // ---- P1: the lane's pieces in ascending square order
        int n = 0;
        bool err = false;
#pragma unroll 1
        for (int it = 0; it < 16; ++it) {
            const bool has = live && (m0 | m1 | m2) != 0u;
            if (__ballot(has) == 0ull) break;
            if (has) {
                int sq;
                if (m0) { sq = __ffs(m0) - 1; m0 &= m0 - 1u; }
                else if (m1) { sq = 32 + __ffs(m1) - 1; m1 &= m1 - 1u; }
                else { sq = 64 + __ffs(m2) - 1; m2 &= m2 - 1u; }
                const int c = Bl[sq];
                if (n <= CZD_MAXMOVES) n += czd_gen_piece_bf(c, sq, sd, S, leap, row + n);
                else err = true;
            }
        }
        if (n > CZD_MAXMOVES) { err = true; n = CZD_MAXMOVES; }
        int base = n;
        // flying general, main.py:1097-1107: kings on one file with nothing between -> the mover's king captures
        if (live && Ksq >= 0 && ksq >= 0 && (Ksq % 9) == (ksq % 9)) {
            const int fx = Ksq % 9, y0 = Ksq / 9, y1 = ksq / 9;
            const unsigned col = czd_bits(S.occT, fx * 10) & 0x3FFu;
            const unsigned between = (y1 > y0 + 1) ? (((1u << y1) - 1u) & ~((1u << (y0 + 1)) - 1u)) : 0u;
            if ((col & between) == 0u) {
                const int src = sd ? ksq : Ksq, dst = sd ? Ksq : ksq;
                if (base >= CZD_MAXMOVES) err = true;
                else { row[base] = (uint16_t)(src | (dst << 8)); base += 1; }
            }
        }
        // the generator's dump slot (row[start of a piece + 17]) may have left junk behind the list: back to padding
        if (!err) {
#pragma unroll
            for (int k = 0; k < 18; ++k) if (base + k < TPK_LSTRIDE) row[base + k] = (uint16_t)0xFFFF;
        }
        // (src, dst) -> label (label2i, main.py:217) by the lane itself: the iterations are independent, so the LUT gathers
        // (16 KB table, L1-resident) are in flight together instead of one dependent round trip per position
        {
            bool bad = false;
            int nmax = (live && !err) ? base : 0;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) nmax = max(nmax, __shfl_xor(nmax, d, 64));
            for (int e = 0; e < nmax; ++e) {
                if (live && !err && e < base) {
                    const int s2 = row[e];
                    const int l = tab.lut[(s2 & 0xFF) * CZD_NSQ + (s2 >> 8)];
                    if (l < 0) bad = true; else row[e] = (uint16_t)l;
                }
            }
            // a move without a label (only possible on a board no game produces) fails the whole position: count 0xFFFF,
            // padding-only list, empty mask — like k_movegen
            if (bad) err = true;
        }
        cnt[lane] = (live && !err) ? base : (live ? -1 : 0);
        __syncthreads();
        // ---- P2: the legality masks; two positions per pass (a half-wave each), LDS only
        const int half = lane >> 5, l5 = lane & 31;
        if (WANT_MASK) {
#pragma unroll 1
            for (int q2 = 0; q2 < 64; q2 += 2) {
                const int q = q2 + half;
                const int nq = cnt[q];
                const uint16_t *rq = list + q * TPK_LSTRIDE;
                for (int e = l5; e < nq; e += 32) {
                    const int l = rq[e];
                    atomicOr(&mrow[(q & 7) * CZ_MASK_WORDS + (l >> 5)], 1u << (l & 31));
                }
                if ((q2 & 7) == 6) {   // the 8 positions q2-6 .. q2+1 are complete: one contiguous block of 8 x 66 words
                    __syncthreads();
                    const int q0 = q2 - 6, rows = min(8, G - (g0 + q0));
                    if (rows > 0) {
                        uint32_t *dstm = mask + (size_t)(g0 + q0) * CZ_MASK_WORDS;
                        for (int i = lane; i < rows * CZ_MASK_WORDS; i += 64) dstm[i] = mrow[i];
                    }
                    __syncthreads();
                    for (int i = lane; i < 8 * CZ_MASK_WORDS; i += 64) mrow[i] = 0u;
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        if (live) { const int c2 = cnt[lane]; count[p] = c2 < 0 ? (uint16_t)0xFFFF : (uint16_t)c2; }
        if (moves) {
            // 64 rows x 256 bytes: lane -> (row, 16-byte chunk); LDS rows are 292 bytes apart (4-byte aligned)
            const int nrows = min(64, G - g0);
            for (int idx = lane; idx < nrows * 16; idx += 64) {
                const int r = idx >> 4, ch = idx & 15;
                const uint32_t *src = reinterpret_cast<const uint32_t *>(list + r * TPK_LSTRIDE + ch * 8);
                uint4 v4 = make_uint4(src[0], src[1], src[2], src[3]);
                if (cnt[r] < 0) v4 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                reinterpret_cast<uint4 *>(moves + (size_t)(g0 + r) * CZD_MAXMOVES)[ch] = v4;
            }
        }
        __syncthreads();
    }
}

// K2: thread per game (the work per game is a handful of byte moves)
__global__ void k_apply_move(CzTables tab, uint8_t *__restrict__ boards, uint8_t *__restrict__ side,
                             const uint16_t *__restrict__ label, int G, uint64_t *__restrict__ hash,
                             uint8_t *__restrict__ captured, int8_t *__restrict__ terminal) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const uint16_t l = label[g];
    uint8_t *b = boards + (size_t)g * CZ_NSQ;
    uint8_t cap = 0;
    if (l < CZ_NLABELS) {
        const int src = tab.srcdst[l] & 0xFF, dst = tab.srcdst[l] >> 8;
        const uint8_t pc = b[src];
        cap = b[dst];
        b[dst] = pc;  // sim_do_action, main.py:671-672
        b[src] = 0;
        if (hash) {
            uint64_t h = hash[g];
            if (pc) h ^= tab.zob[pc * CZ_NSQ + src] ^ tab.zob[pc * CZ_NSQ + dst];
            if (cap) h ^= tab.zob[cap * CZ_NSQ + dst];
            h ^= tab.zob[15 * CZ_NSQ];
            hash[g] = h;
        }
        side[g] = side[g] ? 0 : 1;
    }
    if (captured) captured[g] = cap;  // is_kill_move != 0, main.py:219-227
    if (terminal) {
        bool K = false, k = false;
        for (int i = 0; i < CZ_NSQ; ++i) { K |= (b[i] == 1); k |= (b[i] == 8); }
        terminal[g] = (int8_t)((K ? 0 : 1) | (k ? 0 : 2));  // find('K') == -1 / find('k') == -1, main.py:409-413
    }
}

__global__ void k_hash(CzTables tab, const uint8_t *__restrict__ boards, const uint8_t *__restrict__ side, int G,
                       uint64_t *__restrict__ hash) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const uint8_t *b = boards + (size_t)g * CZ_NSQ;
    uint64_t h = side[g] ? tab.zob[15 * CZ_NSQ] : 0ull;
    for (int q = 0; q < CZ_NSQ; ++q) { const int c = b[q]; if (c) h ^= tab.zob[c * CZ_NSQ + q]; }
    hash[g] = h;
}

template <typename T>
__global__ __launch_bounds__(64) void k_encode_planes(const uint8_t *__restrict__ boards, const uint8_t *__restrict__ side,
                                                      int G, T *__restrict__ planes, int C, int quirk, T one) {
    __shared__ __attribute__((aligned(16))) uint8_t b[CZD_BOARD_LDS];
    const int lane = threadIdx.x;
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        load_board(boards + (size_t)g * CZ_NSQ, b, lane);
        czd_wave_encode_planes<T>(b, side[g] ? 1 : 0, quirk, planes + (size_t)g * 90 * C, C, one, lane);
        __syncthreads();
    }
}

inline int grid_for(int G) { return G < 65536 ? G : 65536; }

}  // namespace

int czk_movegen(cz_ctx *c, const uint8_t *boards, const uint8_t *side, int G, uint16_t *moves, uint16_t *count, uint32_t *mask) {
    if (G == 0) return CZ_OK;
    if (moves && (reinterpret_cast<uintptr_t>(moves) & 15u)) { cz_set_error("cz_movegen: moves must be 16-byte aligned"); return CZ_EINVAL; }
    // default: four positions per wave (k_movegen).  CCHESS_MOVEGEN=lane selects the one-position-per-lane experiment
    // (k_movegen_tp: correct — the same golden tests — but 0.5-0.65x the speed, see the comment at the kernel).
    const char *fe = getenv("CCHESS_MOVEGEN");   // read per call: the tests switch it
    const bool tp = fe && fe[0] == 'l';
    if (tp) {
        const int grid = grid_for((G + 63) / 64);
        if (mask) hipLaunchKernelGGL(k_movegen_tp<true>, dim3(grid), dim3(64), 0, c->stream, c->tab, boards, side, G, moves, count, mask);
        else hipLaunchKernelGGL(k_movegen_tp<false>, dim3(grid), dim3(64), 0, c->stream, c->tab, boards, side, G, moves, count, mask);
    } else {
        hipLaunchKernelGGL(k_movegen, dim3(grid_for((G + 3) / 4)), dim3(64), 0, c->stream, c->tab, boards, side, G, moves, count, mask);
    }
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_apply_move(cz_ctx *c, uint8_t *boards, uint8_t *side, const uint16_t *label, int G, uint64_t *hash, uint8_t *captured, int8_t *terminal) {
    if (G == 0) return CZ_OK;
    hipLaunchKernelGGL(k_apply_move, dim3((G + 255) / 256), dim3(256), 0, c->stream, c->tab, boards, side, label, G, hash, captured, terminal);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_hash(cz_ctx *c, const uint8_t *boards, const uint8_t *side, int G, uint64_t *hash) {
    if (G == 0) return CZ_OK;
    hipLaunchKernelGGL(k_hash, dim3((G + 255) / 256), dim3(256), 0, c->stream, c->tab, boards, side, G, hash);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_encode_planes(cz_ctx *c, const uint8_t *boards, const uint8_t *side, int G, void *planes, int dtype, int C, int quirk) {
    if (G == 0) return CZ_OK;
    if (dtype == CZ_F32)
        hipLaunchKernelGGL(k_encode_planes<float>, dim3(grid_for(G)), dim3(64), 0, c->stream, boards, side, G, (float *)planes, C, quirk, 1.0f);
    else
        hipLaunchKernelGGL(k_encode_planes<uint16_t>, dim3(grid_for(G)), dim3(64), 0, c->stream, boards, side, G, (uint16_t *)planes, C, quirk, (uint16_t)(dtype == CZ_F16 ? 0x3C00 : 0x3F80));
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}
