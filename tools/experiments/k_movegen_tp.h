// tools/experiments/k_movegen_tp.h — the thread-per-position ORDERED-LIST move generator of round 3 (CCHESS_MOVEGEN=lane), removed from
// the library in round 4: measured 0.5-0.65x the speed of k_movegen.  Kept as the record of the experiment;
// it compiled inside cchess_zero_amd/csrc/cz_rules.hip (anonymous namespace, after k_movegen).  Its mask-only successor, which
// drops the ordered list and with it the 25 KB of LDS per wave that cost this kernel its occupancy, is k_movegen_mask (cz_rules.hip,
// cz_maskgen.h).

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

