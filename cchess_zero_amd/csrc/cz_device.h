// cz_device.h — wave64 device primitives for Xiangqi rules on gfx950.
//
// One wavefront (64 lanes) owns one position whose 90-byte board lives in LDS.
// Functions here are called with all 64 lanes active and uniform control flow; they use
// __syncthreads() as the LDS fence, so kernels using them launch 64-thread workgroups
// (one wave = one workgroup = one game; thousands of games fill the 256 CUs).
//
// Semantics restate chengstone/cchess-zero main.py (cited per function); the parity
// checker is oracle/cchess_oracle.c — nothing in this file depends on it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CZD_NSQ 90
#define CZD_BOARD_LDS 96   // board bytes padded to a multiple of 16
#define CZD_STAGE_STRIDE 18  // per-lane staging slots (max 17 moves for a rook/cannon)
#define CZD_MAXMOVES 128
#define CZD_NLABELS 2086

struct CzTables {
    const int16_t *lut;      // [90*90] label or -1           (label2i, main.py:217)
    const int16_t *unflip;   // [2086]                         (unflipped_index, main.py:214)
    const uint16_t *srcdst;  // [2086] src | dst << 8
    const uint64_t *zob;     // [15*90] + side key at [15*90]
};

__device__ __forceinline__ int czd_wave_excl_scan(int v, int lane, int *total) {
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    *total = __shfl(x, 63, 64);
    return x - v;
}

// ---- 90-bit square sets ---------------------------------------------------------------------------
// The rules are evaluated on wave-uniform bit sets built with __ballot (lane = square), not by walking the
// board in LDS: a rook ray is a find-first-set on 9 or 10 bits instead of up to nine dependent LDS loads in a
// divergent loop.  Two orders are kept: rank-major (bit y*9+x, the board's own order) for rank rays and
// single-square tests, file-major (bit x*10+y) for file rays.
struct CzdSet { unsigned long long lo, hi; };   // bits 0..63, 64..89
__device__ __forceinline__ bool czd_tst(CzdSet m, int i) { return ((i < 64 ? m.lo >> i : m.hi >> (i - 64)) & 1ull) != 0; }
__device__ __forceinline__ unsigned czd_bits(CzdSet m, int s) {   // the bits from position s (0 <= s <= 81) upward
    unsigned long long v;
    if (s == 0) v = m.lo;
    else if (s < 64) v = (m.lo >> s) | (m.hi << (64 - s));
    else v = m.hi >> (s - 64);
    return (unsigned)v;
}
__device__ __forceinline__ CzdSet czd_andn(CzdSet a, CzdSet b) { CzdSet r; r.lo = a.lo & ~b.lo; r.hi = a.hi & ~b.hi; return r; }

struct CzdBoardSets { CzdSet occ, enemy, occT, enemyT; };   // enemy = pieces of the side NOT to move

// b: LDS board [96] (visible to all lanes).  All lanes must call this.
__device__ __forceinline__ CzdBoardSets czd_board_sets(const uint8_t *b, int side, int lane) {
    const int c0 = b[lane], c1 = (lane + 64 < CZD_NSQ) ? b[lane + 64] : 0;
    const int t0 = lane, t1 = lane + 64;   // file-major index t = x*10 + y  ->  square y*9 + x
    const int d0 = b[(t0 % 10) * 9 + t0 / 10], d1 = (t1 < CZD_NSQ) ? b[(t1 % 10) * 9 + t1 / 10] : 0;
    CzdBoardSets s;
    CzdSet blk, blkT;
    s.occ.lo = __ballot(c0 != 0); s.occ.hi = __ballot(c1 != 0);
    blk.lo = __ballot(c0 > 7);    blk.hi = __ballot(c1 > 7);
    s.occT.lo = __ballot(d0 != 0); s.occT.hi = __ballot(d1 != 0);
    blkT.lo = __ballot(d0 > 7);    blkT.hi = __ballot(d1 > 7);
    s.enemy = side ? czd_andn(s.occ, blk) : blk;       // black to move: the enemy is red (occupied, not black)
    s.enemyT = side ? czd_andn(s.occT, blkT) : blkT;
    return s;
}

// Per-lane generator for the piece on `sq` (if it belongs to `side`), in the exact emission order of
// GameBoard.get_legal_moves (main.py:757-1095).  Writes src | dst << 8 to st[0..n) (the label lookup happens
// after compaction, coalesced).  c = piece code on sq.
__device__ __forceinline__ int czd_gen_piece(int c, int sq, int side, const CzdBoardSets &S, uint16_t *st) {
    const int black = c > 7;
    if (c == 0 || black != side) return 0;
    const int t = black ? c - 7 : c;
    const int y = sq / 9, x = sq - y * 9;
    int n = 0;
    auto emit = [&](int ty, int tx) { st[n++] = (uint16_t)(sq | ((ty * 9 + tx) << 8)); };
    auto occ = [&](int ty, int tx) { return czd_tst(S.occ, ty * 9 + tx); };
    auto notown = [&](int ty, int tx) {   // validate_move, main.py:727: empty or enemy
        const int q = ty * 9 + tx;
        return !czd_tst(S.occ, q) || czd_tst(S.enemy, q);
    };
    auto inb = [&](int ty, int tx) { return ty >= 0 && tx >= 0 && ty < 10 && tx < 9; };  // check_bounds :717
    switch (t) {
    case 3:    // R/r  main.py:757-833: -x, +x, -y, +y: slide over empties, capture the first enemy
    case 7: {  // C/c  main.py:947-1062: slide over empties; behind exactly one screen capture the first enemy
        const bool cannon = t == 7;
        const unsigned row = czd_bits(S.occ, y * 9) & 0x1FFu, rowE = czd_bits(S.enemy, y * 9) & 0x1FFu;
        const unsigned col = czd_bits(S.occT, x * 10) & 0x3FFu, colE = czd_bits(S.enemyT, x * 10) & 0x3FFu;
        // one direction on a line of `len` squares: own index p, occupancy o, enemies e, step -1 / +1
        auto ray = [&](unsigned o, unsigned e, int p, int len, int dir, bool along_x) {
            int hit;   // index of the first occupied square in direction dir, or -1 / len
            if (dir < 0) { const unsigned m = o & ((1u << p) - 1u); hit = m ? 31 - __clz(m) : -1; }
            else { const unsigned m = o >> (p + 1); hit = m ? p + 1 + (__ffs(m) - 1) : len; }
            for (int q = p + dir; q != hit; q += dir) { if (along_x) emit(y, q); else emit(q, x); }
            if (hit < 0 || hit >= len) return;
            if (!cannon) { if ((e >> hit) & 1u) { if (along_x) emit(y, hit); else emit(hit, x); } return; }
            int hit2;  // the cannon's target: the next occupied square behind the screen
            if (dir < 0) { const unsigned m = o & ((1u << hit) - 1u); hit2 = m ? 31 - __clz(m) : -1; }
            else { const unsigned m = o >> (hit + 1); hit2 = m ? hit + 1 + (__ffs(m) - 1) : len; }
            if (hit2 >= 0 && hit2 < len && ((e >> hit2) & 1u)) { if (along_x) emit(y, hit2); else emit(hit2, x); }
        };
        ray(row, rowE, x, 9, -1, true);
        ray(row, rowE, x, 9, +1, true);
        ray(col, colE, y, 10, -1, false);
        ray(col, colE, y, 10, +1, false);
    } break;
    case 5: {  // N/n  main.py:835-856: (2i,j) with leg (i,0), then (i,2j) with leg (0,j)
        for (int i = -1; i <= 1; i += 2)
            for (int j = -1; j <= 1; j += 2) {
                int ty = y + 2 * i, tx = x + j;
                if (inb(ty, tx) && notown(ty, tx) && !occ(y + i, x)) emit(ty, tx);
                ty = y + i; tx = x + 2 * j;
                if (inb(ty, tx) && notown(ty, tx) && !occ(y, x + j)) emit(ty, tx);
            }
    } break;
    case 4: {  // B/b  main.py:857-888: two-step diagonals, eye empty, own half
        for (int i = -2; i <= 2; i += 4) {
            const int h = i / 2;
            int ty = y + i, tx = x + i;
            if (inb(ty, tx) && notown(ty, tx) && (side ? ty >= 5 : ty <= 4) && !occ(y + h, x + h)) emit(ty, tx);
            tx = x - i;
            if (inb(ty, tx) && notown(ty, tx) && (side ? ty >= 5 : ty <= 4) && !occ(y + h, x - h)) emit(ty, tx);
        }
    } break;
    case 2: {  // A/a  main.py:889-918: palace diagonals
        for (int i = -1; i <= 1; i += 2) {
            int ty = y + i, tx = x + i;
            if (inb(ty, tx) && notown(ty, tx) && (side ? ty >= 7 : ty <= 2) && tx >= 3 && tx <= 5) emit(ty, tx);
            tx = x - i;
            if (inb(ty, tx) && notown(ty, tx) && (side ? ty >= 7 : ty <= 2) && tx >= 3 && tx <= 5) emit(ty, tx);
        }
    } break;
    case 1: {  // K/k  main.py:919-946: (0,-1) (0,+1) (-1,0) (+1,0) inside the palace
        for (int i = 0; i < 2; ++i)
            for (int s = -1; s <= 1; s += 2) {
                int ty = y + i * s, tx = x + (1 - i) * s;
                if (inb(ty, tx) && notown(ty, tx) && (side ? ty >= 7 : ty <= 2) && tx >= 3 && tx <= 5) emit(ty, tx);
            }
    } break;
    case 6: {  // P/p  main.py:1063-1095: black advances to y-1, red to y+1; sideways past the river
        if (side) {
            if (inb(y - 1, x) && notown(y - 1, x)) emit(y - 1, x);
            if (y < 5) {
                if (inb(y, x + 1) && notown(y, x + 1)) emit(y, x + 1);
                if (inb(y, x - 1) && notown(y, x - 1)) emit(y, x - 1);
            }
        } else {
            if (inb(y + 1, x) && notown(y + 1, x)) emit(y + 1, x);
            if (y > 4) {
                if (inb(y, x + 1) && notown(y, x + 1)) emit(y, x + 1);
                if (inb(y, x - 1) && notown(y, x - 1)) emit(y, x - 1);
            }
        }
    } break;
    default: break;
    }
    return n;
}

// ---- the same generator without divergent control flow ------------------------------------------------------------
// czd_gen_piece above is the readable statement of the rules (and what round 1 ran).  Its if-per-candidate structure
// costs more scalar instructions (exec-mask save / restore / branch) than vector ones: the SQ counters of k_select
// showed 894 SALU + 210 branch instructions per wave against 1 264 VALU, ~90 % of the scalar ones from the generator,
// and the scalar unit is shared by the 32 waves of a CU.  Here every candidate is evaluated by every lane: a move is
// written to st[valid ? n : 17] (slot 17 is a dump: a piece has at most 17 moves) and n += valid.  Only the
// slider / leaper split is a real branch.  Emission order is the reference's, candidate by candidate.
//   leapers: 8 candidates per (piece kind, side) from a table: (dy, dx) of the destination, (by, bx) of the square that
//            must be empty (horse leg / elephant eye), flags; destination inside the kind's rectangle (board, own half,
//            palace), not an own piece;  pawn rows differ by side (forward direction, sideways only past the river).
//   sliders: per direction the first occupied square by find-first-set on the 9 / 10 line bits, the run of empty
//            squares in front of it, then the capture (rook: that square if enemy; cannon: the next occupied one).
// entry: bits 0-2 dy+2, 3-5 dx+2, 6-7 by+1, 8-9 bx+1, 10 has-block, 11 needs the source past the river, 12 valid
#define CZD_LE(dy, dx, by, bx, blk, src) (uint32_t)(((dy) + 2) | (((dx) + 2) << 3) | (((by) + 1) << 6) | (((bx) + 1) << 8) | ((blk) << 10) | ((src) << 11) | (1u << 12))
__constant__ uint32_t c_czd_leap[8][8] = {
    // row 0: black pawn (p): forward = y-1; sideways below rank 5 (main.py:1063-1078)
    {CZD_LE(-1, 0, 0, 0, 0, 0), CZD_LE(0, 1, 0, 0, 0, 1), CZD_LE(0, -1, 0, 0, 0, 1), 0, 0, 0, 0, 0},
    // row 1: K/k (main.py:919-946): (0,-1) (0,+1) (-1,0) (+1,0)
    {CZD_LE(0, -1, 0, 0, 0, 0), CZD_LE(0, 1, 0, 0, 0, 0), CZD_LE(-1, 0, 0, 0, 0, 0), CZD_LE(1, 0, 0, 0, 0, 0), 0, 0, 0, 0},
    // row 2: A/a (main.py:889-918): (i,i), (i,-i) for i = -1, +1
    {CZD_LE(-1, -1, 0, 0, 0, 0), CZD_LE(-1, 1, 0, 0, 0, 0), CZD_LE(1, 1, 0, 0, 0, 0), CZD_LE(1, -1, 0, 0, 0, 0), 0, 0, 0, 0},
    // row 3: rook (slider, unused)
    {0, 0, 0, 0, 0, 0, 0, 0},
    // row 4: B/b (main.py:857-888): (i,i) eye (i/2,i/2), (i,-i) eye (i/2,-i/2) for i = -2, +2
    {CZD_LE(-2, -2, -1, -1, 1, 0), CZD_LE(-2, 2, -1, 1, 1, 0), CZD_LE(2, 2, 1, 1, 1, 0), CZD_LE(2, -2, 1, -1, 1, 0), 0, 0, 0, 0},
    // row 5: N/n (main.py:835-856): for i, j in (-1,+1)^2: (2i, j) leg (i, 0), then (i, 2j) leg (0, j)
    {CZD_LE(-2, -1, -1, 0, 1, 0), CZD_LE(-1, -2, 0, -1, 1, 0), CZD_LE(-2, 1, -1, 0, 1, 0), CZD_LE(-1, 2, 0, 1, 1, 0),
     CZD_LE(2, -1, 1, 0, 1, 0), CZD_LE(1, -2, 0, -1, 1, 0), CZD_LE(2, 1, 1, 0, 1, 0), CZD_LE(1, 2, 0, 1, 1, 0)},
    // row 6: red pawn (P): forward = y+1; sideways above rank 4 (main.py:1079-1095)
    {CZD_LE(1, 0, 0, 0, 0, 0), CZD_LE(0, 1, 0, 0, 0, 1), CZD_LE(0, -1, 0, 0, 0, 1), 0, 0, 0, 0, 0},
    // row 7: cannon (slider, unused)
    {0, 0, 0, 0, 0, 0, 0, 0},
};

__device__ __forceinline__ int czd_gen_piece_bf(int c, int sq, int side, const CzdBoardSets &S, const uint32_t *leap, uint16_t *st) {
    const int t = c > 7 ? c - 7 : c;     // the caller only passes pieces of the side to move
    const int y = sq / 9, x = sq - y * 9;
    int n = 0;
    if (t == 3 || t == 7) {
        const bool cannon = t == 7;
        const unsigned row = czd_bits(S.occ, y * 9) & 0x1FFu, rowE = czd_bits(S.enemy, y * 9) & 0x1FFu;
        const unsigned col = czd_bits(S.occT, x * 10) & 0x3FFu, colE = czd_bits(S.enemyT, x * 10) & 0x3FFu;
#pragma unroll
        for (int d = 0; d < 4; ++d) {   // -x, +x, -y, +y (main.py:757-833 / 947-1062)
            const bool along_x = d < 2, neg = (d & 1) == 0;
            const unsigned o = along_x ? row : col, e = along_x ? rowE : colE;
            const int p = along_x ? x : y, len = along_x ? 9 : 10;
            int hit, tgt;
            if (neg) {
                const unsigned m = o & ((1u << p) - 1u);
                hit = m ? 31 - __clz(m) : -1;
                const unsigned m2 = hit >= 0 ? o & ((1u << hit) - 1u) : 0u;
                tgt = cannon ? (m2 ? 31 - __clz(m2) : -1) : hit;
            } else {
                const unsigned m = o >> (p + 1);
                hit = m ? p + __ffs(m) : len;
                const unsigned m2 = hit < len ? o >> (hit + 1) : 0u;
                tgt = cannon ? (m2 ? hit + __ffs(m2) : len) : hit;
            }
            const int run = neg ? p - hit - 1 : hit - p - 1;
            const bool cap = tgt >= 0 && tgt < len && ((e >> (tgt & 31)) & 1u);
            const int step = neg ? -1 : 1;
            // dst square of line index q: along x  y*9 + q ; along y  q*9 + x
            const int mul = along_x ? 1 : 9, add = along_x ? y * 9 : x;
            // at most len - 1 squares; the loop leaves (wave-uniform branch) once no lane of the wave has squares left
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                if (i >= len - 1 || __ballot(i < run) == 0ull) break;
                const int q = p + step * (i + 1);
                st[i < run ? n + i : 17] = (uint16_t)(sq | ((q * mul + add) << 8));
            }
            n += run;
            st[cap ? n : 17] = (uint16_t)(sq | ((tgt * mul + add) << 8));
            n += cap ? 1 : 0;
        }
        return n;
    }
    // leapers: the kind's rectangle
    const bool palace = t <= 2, half = t == 4;
    const int ylo = palace ? (side ? 7 : 0) : (half ? (side ? 5 : 0) : 0);
    const int yhi = palace ? (side ? 9 : 2) : (half ? (side ? 9 : 4) : 9);
    const int xlo = palace ? 3 : 0, xhi = palace ? 5 : 8;
    const bool past_river = side ? y < 5 : y > 4;
    const uint32_t *row = leap + ((t == 6 && side) ? 0 : t) * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t e = row[k];
        const int ty = y + (int)(e & 7u) - 2, tx = x + (int)((e >> 3) & 7u) - 2;
        const int q = ty * 9 + tx;
        const int bq = (y + (int)((e >> 6) & 3u) - 1) * 9 + x + (int)((e >> 8) & 3u) - 1;
        bool ok = ((e >> 12) & 1u) && ty >= ylo && ty <= yhi && tx >= xlo && tx <= xhi;
        const int qs = ok ? q : sq, bqs = ok ? bq : sq;   // keep the bit tests in range (a valid destination has its leg / eye on the board)
        ok = ok && (!czd_tst(S.occ, qs) || czd_tst(S.enemy, qs));                 // validate_move, main.py:727
        ok = ok && !(((e >> 10) & 1u) && czd_tst(S.occ, bqs));                    // leg / eye must be empty
        ok = ok && (!((e >> 11) & 1u) || past_river);
        st[ok ? n : 17] = (uint16_t)(sq | (q << 8));
        n += ok ? 1 : 0;
    }
    return n;
}

// ---- ordered pseudo-legal move lists (GameBoard.get_legal_moves, main.py:743-1109), lane = piece ----------------------
// A position has at most 16 pieces of the side to move, so a wave64 generates for up to FOUR positions at once: lanes
// 16 q .. 16 q + 15 own position q, lane 16 q + s the s-th piece of the mover in scan order (ascending square = y outer,
// x inner, main.py:754-755).  Round 1 put one SQUARE on every lane of a whole wave per position: two passes of the
// divergent per-piece generator with two thirds of the lanes idle — the kernels were issue-bound at 0.6 G positions/s.
//   stage A  per position (whole wave, lane = square): occupancy bit sets by ballot, the mover's piece list by
//            popcount-rank, king squares; bit sets and piece list go through LDS to the position's 16 lanes
//   stage B  once: lane = (position, piece) runs czd_gen_piece into its staging row
//   stage C  16-lane segmented prefix sum of the per-piece counts (DPP row shifts: a row IS 16 lanes), runs copied to
//            the position's output list, flying general appended (main.py:1097-1107), (src, dst) -> label through the LUT
struct CzdGroupLds {
    unsigned long long sets[4][8];   // occ, enemy, occT, enemyT (lo, hi)
    uint16_t pl[4][16];              // mover's pieces: sq | code << 8
    int16_t kings[4][2];             // square of 'K', 'k' (or -1)
    int16_t npc[4];
    uint32_t leap[64];               // c_czd_leap, staged once per call
};

__device__ __forceinline__ int czd_row_excl_scan16(int v, int *total, int lane) {
    int x = v;   // inclusive scan inside each row of 16 lanes
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   // row_shr:1, out-of-row lanes read 0
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);   // row_shr:8
    *total = __shfl(x, lane | 15, 64);
    return x - v;
}

// b: LDS boards, position p at b + p * STRIDE; side_of(p) its side to move; NP positions (1..4).
// out: LDS [NP][128] u16 (labels on return); stage: LDS [64 * 18] u16; G: LDS scratch.
// Returns, in every lane of group q, the move count of position q (-1 on overflow / unlabeled move; 0 for q >= NP).
template <int NP, int STRIDE, typename SideFn>
__device__ __forceinline__ int czd_group_movegen(const uint8_t *b, SideFn side_of, const int16_t *lut, CzdGroupLds &G,
                                                 uint16_t *stage, uint16_t *out, int lane) {
    G.leap[lane] = (&c_czd_leap[0][0])[lane];
    // ---- stage A (not unrolled: four copies of it only cost registers, i.e. occupancy)
#pragma unroll 1
    for (int p = 0; p < NP; ++p) {
        const uint8_t *bp = b + p * STRIDE;
        const int side = side_of(p);
        const CzdBoardSets S = czd_board_sets(bp, side, lane);
        const int c0 = bp[lane], c1 = (lane + 64 < CZD_NSQ) ? bp[lane + 64] : 0;
        const bool own0 = c0 != 0 && (c0 > 7) == (side != 0), own1 = c1 != 0 && (c1 > 7) == (side != 0);
        const unsigned long long o0 = __ballot(own0), o1 = __ballot(own1);
        const int n0 = __popcll(o0);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (own0) { const int i = __popcll(o0 & below); if (i < 16) G.pl[p][i] = (uint16_t)(lane | (c0 << 8)); }
        if (own1) { const int i = n0 + __popcll(o1 & below); if (i < 16) G.pl[p][i] = (uint16_t)((lane + 64) | (c1 << 8)); }
        const unsigned long long K0 = __ballot(c0 == 1), K1 = __ballot(c1 == 1), k0 = __ballot(c0 == 8), k1 = __ballot(c1 == 8);
        if (lane == 0) {
            G.sets[p][0] = S.occ.lo; G.sets[p][1] = S.occ.hi; G.sets[p][2] = S.enemy.lo; G.sets[p][3] = S.enemy.hi;
            G.sets[p][4] = S.occT.lo; G.sets[p][5] = S.occT.hi; G.sets[p][6] = S.enemyT.lo; G.sets[p][7] = S.enemyT.hi;
            G.kings[p][0] = (int16_t)(K0 ? __ffsll((long long)K0) - 1 : (K1 ? 64 + __ffsll((long long)K1) - 1 : -1));
            G.kings[p][1] = (int16_t)(k0 ? __ffsll((long long)k0) - 1 : (k1 ? 64 + __ffsll((long long)k1) - 1 : -1));
            const int np_ = n0 + __popcll(o1);
            G.npc[p] = (int16_t)(np_ > 16 ? -1 : np_);   // more than 16 pieces of one colour: not a Xiangqi position
        }
    }
    __syncthreads();
    // ---- stage B
    const int q = lane >> 4, s = lane & 15;
    const bool live = q < NP;
    const int qq = live ? q : 0;
    const int side = live ? side_of(qq) : 0;
    const int npc = G.npc[qq];
    bool err = live && npc < 0;
    CzdBoardSets S;
    S.occ.lo = G.sets[qq][0]; S.occ.hi = G.sets[qq][1]; S.enemy.lo = G.sets[qq][2]; S.enemy.hi = G.sets[qq][3];
    S.occT.lo = G.sets[qq][4]; S.occT.hi = G.sets[qq][5]; S.enemyT.lo = G.sets[qq][6]; S.enemyT.hi = G.sets[qq][7];
    uint16_t *st = stage + lane * CZD_STAGE_STRIDE;
    int n = 0;
    if (live && s < npc) {
        const int pc = G.pl[qq][s];
        n = czd_gen_piece_bf(pc >> 8, pc & 0xFF, side, S, G.leap, st);
    }
    // ---- stage C
    int total;
    const int off = czd_row_excl_scan16(n, &total, lane);
    uint16_t *o = out + qq * CZD_MAXMOVES;
    if (off + n > CZD_MAXMOVES) { err = true; n = 0; }
    for (int k = 0; k < n; ++k) o[off + k] = st[k];   // own staging row: no cross-lane hazard
    int base = total > CZD_MAXMOVES ? CZD_MAXMOVES : total;
    // flying general, main.py:1097-1107: kings on one file with nothing between -> the mover's king captures
    const int Ksq = G.kings[qq][0], ksq = G.kings[qq][1];
    if (live && Ksq >= 0 && ksq >= 0 && (Ksq % 9) == (ksq % 9)) {
        // the reference walks from the red king towards higher ranks (main.py:1100-1104)
        const int fx = Ksq % 9, y0 = Ksq / 9, y1 = ksq / 9;
        const unsigned col = czd_bits(S.occT, fx * 10) & 0x3FFu;
        const unsigned between = (y1 > y0 + 1) ? (((1u << y1) - 1u) & ~((1u << (y0 + 1)) - 1u)) : 0u;
        if ((col & between) == 0u) {
            const int src = side ? ksq : Ksq, dst = side ? Ksq : ksq;
            if (base >= CZD_MAXMOVES) err = true;
            else { if (s == 0) o[base] = (uint16_t)(src | (dst << 8)); base += 1; }
        }
    }
    __syncthreads();
    // (src, dst) -> label (label2i, main.py:217); every entry is read and rewritten by the same lane.  With one position
    // per wave all 64 lanes share the list (one or two LUT round trips instead of up to eight)
    {
        const int step = NP == 1 ? 64 : 16, first = NP == 1 ? lane : s;
        const int lim = NP == 1 ? __shfl(base, 0, 64) : base;
        if (live || NP == 1)
            for (int i = first; i < lim; i += step) {
                const int sd = o[i];
                const int l = lut[(sd & 0xFF) * CZD_NSQ + (sd >> 8)];
                if (l < 0) err = true; else o[i] = (uint16_t)l;
            }
    }
    // an error anywhere in the group spoils the group
    const unsigned long long em = __ballot(err);
    const bool gerr = NP == 1 ? em != 0ull : ((em >> (lane & 48)) & 0xFFFFull) != 0ull;
    __syncthreads();
    return live ? (gerr ? -1 : base) : 0;
}

// One position per wave (the search kernels: one wave = one tree): group 0 does the work.
//   b LDS board [96]; stage LDS [64*18] u16; out LDS [128] u16.  Returns the move count (wave-uniform) or -1.
__device__ __forceinline__ int czd_wave_movegen(const uint8_t *b, int side, const int16_t *lut, CzdGroupLds &G,
                                                uint16_t *stage, uint16_t *out, int lane) {
    const int n = czd_group_movegen<1, CZD_BOARD_LDS>(b, [side](int) { return side; }, lut, G, stage, out, lane);
    return __shfl(n, 0, 64);
}

// MCTS_tree.generate_inputs (main.py:531-533): try_flip (:560-574) for black — reverse the rank
// order and swap case — then state_to_positions (:547-557) with its 9-stride read (quirk Q1).
// Writes [9][10][C] elements of T (float or bf16 bits), C >= 14, coalesced across the wave.
template <typename T>
__device__ __forceinline__ void czd_wave_encode_planes(const uint8_t *b, int side, int quirk_q1, T *out,
                                                       int C, T one, int lane) {
    if (C == 16 && sizeof(T) == 2) {
        // the fused net kernel's input format: 16 two-byte channels = 32 bytes per cell = two 16-byte pieces; lane l writes pieces
        // l, l + 64, l + 128 of the position's 180, so every store instruction covers a contiguous kilobyte (one lane per CELL
        // with two stores each left every instruction writing alternate halves of its 128-byte lines: K3 ran at 3.0 TB/s of a
        // measured 6.8 TB/s store ceiling, tools/hbm_ceiling.py)
        uint4 *o4 = reinterpret_cast<uint4 *>(out);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int piece = lane + 64 * r, cell = piece >> 1, half = piece & 1;
            if (piece < 180) {
                int src;
                if (quirk_q1) { const int h = cell / 10, w = cell - h * 10; src = h * 9 + w; }
                else { const int xx = cell / 10, yy = cell - xx * 10; src = yy * 9 + xx; }
                int code = 0;
                if (src < CZD_NSQ) {
                    if (side) {
                        const int y = src / 9, x = src - y * 9;
                        const int p = b[(9 - y) * 9 + x];
                        code = p == 0 ? 0 : (p > 7 ? p - 7 : p + 7);
                    } else code = b[src];
                }
                const int idx = code - 1;   // channel of the 1, or -1
                const unsigned v = (unsigned)(unsigned short)one << ((idx & 1) * 16);
                const int wd = (idx >> 1) - 4 * half;    // 32-bit word 0..3 of this piece, or outside it
                uint4 o = make_uint4(0, 0, 0, 0);
                if (idx >= 0) { o.x = wd == 0 ? v : 0u; o.y = wd == 1 ? v : 0u; o.z = wd == 2 ? v : 0u; o.w = wd == 3 ? v : 0u; }
                o4[piece] = o;
            }
        }
        return;
    }
    const int total = 90 * C;
    for (int e = lane; e < total; e += 64) {
        const int cell = e / C, c = e - cell * C;
        int src;  // square of the canonical (flipped) board feeding plane cell (h,w)
        if (quirk_q1) {
            const int h = cell / 10, w = cell - h * 10;
            src = h * 9 + w;  // board_state[rank*9+file], rank<9, file<10
        } else {
            const int xx = cell / 10, yy = cell - xx * 10;
            src = yy * 9 + xx;
        }
        int code = 0;
        if (src < CZD_NSQ) {
            if (side) {
                const int y = src / 9, x = src - y * 9;
                const int p = b[(9 - y) * 9 + x];
                code = p == 0 ? 0 : (p > 7 ? p - 7 : p + 7);
            } else code = b[src];
        }
        out[e] = (c < 14 && code == c + 1) ? one : (T)0;
    }
}

__device__ __forceinline__ uint16_t czd_f32_to_bf16_bits(float f) {
    unsigned int u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float czd_bf16_bits_to_f32(uint16_t h) { return __uint_as_float(((unsigned int)h) << 16); }
