// cz_maskgen.h — K1m: the 2086-bit legal-move MASK of a position without the ordered move list.
//
// GameBoard.get_legal_moves (main.py:743-1109) returns an ordered list; what the search's expansion and a policy head
// consume is the SET (which labels are legal).  cz_movegen(moves = NULL) therefore does not need the reference's emission
// order, the staging rows, the prefix sums, the (src, dst) -> label LUT round trips or one LDS atomic per move of
// k_movegen (249 VALU + 156 SALU instructions per position: issue-bound at 7 % of the kernel's HBM roofline).  Here ONE LANE
// owns a position and everything is bit arithmetic in its registers:
//   * the 90 board bytes arrive as 23 dwords; SWAR byte tests + v_dot4_u32_u8 (weights 1, 2, 4, ... 128) turn "byte != 0" /
//     "byte == code" into 90-bit square sets: occupancy and one set per piece kind of the side to move (+ the enemy king);
//   * pieces are visited KIND BY KIND (all lanes run the rook code together, then the knight code, ...: no divergence on the
//     piece type); a piece's squares come off its kind's set by find-first-set;
//   * the move vocabulary (main.py:30-65) lists, per source square, 8 same-rank destinations, 9 same-file destinations and
//     the on-board knight jumps CONTIGUOUSLY, so the legal destinations of a rook / cannon / king / pawn are one <= 17-bit
//     field and a knight's one <= 8-bit field at a per-square base: a ray is find-first-set on the 9 / 10 line bits and two
//     masks, never a loop over squares; advisor / bishop moves are the 48 literals at the end (a [square][direction] table);
//   * a field is OR-ed into the lane's private 66-word row with two ds_or_b32; count = popcount of the fields.
// The rules are those of czd_gen_piece (cz_device.h), restated as set operations; tests/test_hip_rules.py holds both kernels
// to the same 4 381 golden positions and the 20 k oracle-checked corpus, and tools/maskgen_host_check.cpp runs this very
// function on the CPU against the golden lists (the function is host-compilable on purpose).
#pragma once
#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__)
#define CZM_FN __device__ __forceinline__
#elif defined(__HIPCC__)
#define CZM_FN __host__ __device__ inline
#else
#define CZM_FN inline
#endif

#define CZM_NLIT_BASE 2038   // first advisor literal (main.py:54-61): labels 2038 .. 2085

// per-square tables derived from the label LUT on the host (cz_api.hip): 630 bytes, copied to LDS by every workgroup
struct CzmTables {
    uint16_t base[90];     // label of the first same-rank destination of source square sq = y * 9 + x
    uint8_t knon[90];      // bit j: knight jump j (vocabulary order: (dx,dy) = (-2,-1) (-1,-2) (-2,1) (1,-2) (2,-1) (-1,2) (2,1) (1,2)) lands on the board
    uint8_t ab[2][90 * 4]; // [0] advisor (step 1), [1] bishop (step 2): literal of (square, direction d: (dy,dx) = (-s,-s) (-s,+s) (+s,+s) (+s,-s)) - 2038, or 0xFF
};

// host: the tables from the 90 x 90 label LUT (label2i, main.py:217; cz_tables.hip)
inline void czm_build_tables(const int16_t *lut, CzmTables *t) {
    static const int jx[8] = {-2, -1, -2, 1, 2, -1, 2, 1}, jy[8] = {-1, -2, 1, -2, -1, 2, 1, 2};
    for (int sq = 0; sq < 90; ++sq) {
        const int y = sq / 9, x = sq - y * 9;
        t->base[sq] = (uint16_t)lut[sq * 90 + y * 9 + (x == 0 ? 1 : 0)];
        uint8_t on = 0;
        for (int j = 0; j < 8; ++j) {
            const int tx = x + jx[j], ty = y + jy[j];
            if (tx >= 0 && tx < 9 && ty >= 0 && ty < 10) on |= (uint8_t)(1u << j);
        }
        t->knon[sq] = on;
        for (int kind = 0; kind < 2; ++kind)
            for (int d = 0; d < 4; ++d) {
                const int st = kind + 1, sy = d < 2 ? -1 : 1, sx = (d == 0 || d == 3) ? -1 : 1;
                const int ty = y + sy * st, tx = x + sx * st;
                int l = -1;
                if (tx >= 0 && tx < 9 && ty >= 0 && ty < 10) l = lut[sq * 90 + ty * 9 + tx];
                t->ab[kind][sq * 4 + d] = (uint8_t)((l >= CZM_NLIT_BASE && l < CZM_NLIT_BASE + 48) ? l - CZM_NLIT_BASE : 0xFF);
            }
    }
}

struct CzmSet { uint32_t a0, a1, a2; };   // squares 0..31, 32..63, 64..89

CZM_FN uint32_t czm_dot4(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    return c + (a & 0xFF) * (b & 0xFF) + ((a >> 8) & 0xFF) * ((b >> 8) & 0xFF) + ((a >> 16) & 0xFF) * ((b >> 16) & 0xFF) + (a >> 24) * (b >> 24);
#endif
}
CZM_FN uint32_t czm_low(int n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << (n & 31)) - 1u); }   // bits 0 .. n-1, 0 <= n <= 32
CZM_FN bool czm_tst(const CzmSet &s, int q) {   // 0 <= q < 90
    const uint32_t wd = q < 32 ? s.a0 : (q < 64 ? s.a1 : s.a2);
    return ((wd >> (q & 31)) & 1u) != 0u;
}
// low 32 bits of (hi:lo) >> sh, 0 <= sh < 32 (one v_alignbit_b32)
CZM_FN uint32_t czm_funnel(uint32_t hi, uint32_t lo, int sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh);
#else
    return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
#endif
}
// bits [p, p + 32) of the 96-bit value (a2:a1:a0), 0 <= p < 90 (bits beyond 95 read as 0)
CZM_FN uint32_t czm_bits(const CzmSet &s, int p) {
    const uint32_t lo = p < 32 ? s.a0 : (p < 64 ? s.a1 : s.a2);
    const uint32_t hi = p < 32 ? s.a1 : (p < 64 ? s.a2 : 0u);
    return czm_funnel(hi, lo, p & 31);
}
// lowest square of the set (or -1) and its removal
CZM_FN int czm_pop(CzmSet &s) {
    if (s.a0) { const int b = __builtin_ctz(s.a0); s.a0 &= s.a0 - 1u; return b; }
    if (s.a1) { const int b = __builtin_ctz(s.a1); s.a1 &= s.a1 - 1u; return 32 + b; }
    if (s.a2) { const int b = __builtin_ctz(s.a2); s.a2 &= s.a2 - 1u; return 64 + b; }
    return -1;
}
CZM_FN int czm_first(const CzmSet &s) {
    if (s.a0) return __builtin_ctz(s.a0);
    if (s.a1) return 32 + __builtin_ctz(s.a1);
    if (s.a2) return 64 + __builtin_ctz(s.a2);
    return -1;
}

// "byte != 0" of the 90 board bytes XOR rep (rep = 0: occupancy; rep = a piece code in every byte: "byte != code")
CZM_FN CzmSet czm_nonzero_set(const uint32_t (&w)[23], uint32_t rep) {
    uint32_t out[3] = {0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 23; k += 2) {
        const uint32_t f0 = (((w[k] ^ rep) + 0x7F7F7F7Fu) >> 7) & 0x01010101u;      // bytes are <= 15: no carry between bytes
        uint32_t byte = czm_dot4(f0, 0x08040201u, 0u);
        if (k + 1 < 23) {
            const uint32_t f1 = (((w[k + 1] ^ rep) + 0x7F7F7F7Fu) >> 7) & 0x01010101u;
            byte = czm_dot4(f1, 0x80402010u, byte);
        }
        out[k >> 3] |= byte << ((k >> 1 & 3) * 8);   // squares 4k .. 4k+7 -> bits (4k & 31) ..
    }
    return CzmSet{out[0], out[1], out[2]};
}
CZM_FN CzmSet czm_equal_set(const uint32_t (&w)[23], uint32_t code) {
    const CzmSet n = czm_nonzero_set(w, code * 0x01010101u);
    return CzmSet{~n.a0, ~n.a1, ~n.a2 & 0x03FFFFFFu};
}

// 10 bits of file x (bit r = square 9 r + x) of a set
CZM_FN uint32_t czm_file(const CzmSet &s, int x) {   // 0 <= x <= 8
    // t = s >> x, then the bits at the constant positions 9 r: 0, 9, 18, 27 | 36, 45, 54, 63 = t1 bits 4, 13, 22, 31 | 72, 81 = t2 bits 8, 17
    const uint32_t t0 = czm_funnel(s.a1, s.a0, x), t1 = czm_funnel(s.a2, s.a1, x), t2 = s.a2 >> x;
    return (t0 & 1u) | ((t0 >> 9 & 1u) << 1) | ((t0 >> 18 & 1u) << 2) | ((t0 >> 27 & 1u) << 3) | ((t1 >> 4 & 1u) << 4) | ((t1 >> 13 & 1u) << 5) |
           ((t1 >> 22 & 1u) << 6) | ((t1 >> 31) << 7) | ((t2 >> 8 & 1u) << 8) | ((t2 >> 17 & 1u) << 9);
}

// destinations of a rook / cannon at index p on a line of `len` squares: o = occupancy bits, e = enemy bits (main.py:757-833,
// 947-1062): the empty run in both directions; rook: + the first occupied square if it is an enemy; cannon: + the SECOND
// occupied square (behind exactly one screen) if it is an enemy
CZM_FN uint32_t czm_line_dests(uint32_t o, uint32_t e, int p, int len, bool cannon) {
    uint32_t d;
    {   // towards index 0
        const uint32_t m = o & czm_low(p);
        const int hit = m ? 31 - __builtin_clz(m) : -1;
        d = czm_low(p) & ~czm_low(hit + 1);
        const uint32_t m2 = hit >= 0 ? m & czm_low(hit) : 0u;
        const int tgt = cannon ? (m2 ? 31 - __builtin_clz(m2) : -1) : hit;
        if (tgt >= 0) d |= e & (1u << tgt);
    }
    {   // towards index len-1
        const uint32_t m = (o >> p) >> 1;                      // squares above p
        const int hit = m ? p + 1 + __builtin_ctz(m) : len;
        d |= czm_low(hit) & ~czm_low(p + 1);
        const uint32_t m2 = hit < len ? (m >> (hit - p - 1)) >> 1 : 0u;   // squares above hit
        const int tgt = cannon ? (m2 ? hit + 1 + __builtin_ctz(m2) : len) : hit;
        if (tgt < len) d |= e & (1u << tgt);
    }
    return d & czm_low(len);
}
// the 17-bit label field of a source square (x, y): same-rank destinations dx (own file squeezed out) in bits 0..7,
// same-file destinations dy (own rank squeezed out) in bits 8..16 (create_uci_labels, main.py:36-44)
CZM_FN uint32_t czm_ortho_field(uint32_t rank_d, uint32_t file_d, int x, int y) {
    const uint32_t f = (rank_d & czm_low(x)) | (((rank_d >> x) >> 1) << x);
    const uint32_t g = (file_d & czm_low(y)) | (((file_d >> y) >> 1) << y);
    return f | (g << 8);
}

// or_field(bit, field): OR `field` (<= 25 bits) into the position's mask at bit offset `bit`
template <typename OrWord>
CZM_FN void czm_or_field(OrWord &or_word, int bit, uint32_t field) {
    const int wi = bit >> 5, off = bit & 31;
    or_word(wi, field << off);
    or_word(wi + 1, off ? field >> (32 - off) : 0u);     // wi + 1 <= 64 for every base + 25-bit field
}

// w: the 90 board bytes (sq = y * 9 + x, code = 1 + index in "KARBNPCkarbnpc"; bytes 90, 91 of w[22] must be zero);
// side 0 = red ('w', codes 1..7, home ranks 0..4) to move, 1 = black.  Returns the number of legal moves, or -1 when the
// position is not a Xiangqi position the vocabulary can express (more than 16 pieces of a colour; an advisor / bishop move
// without a label).
template <typename OrWord>
CZM_FN int czm_position(const uint32_t (&w)[23], int side, const CzmTables &T, OrWord or_word) {
    const uint32_t own0 = side ? 7u : 0u;   // own piece code = kind + own0 (kind: K 1, A 2, R 3, B 4, N 5, P 6, C 7)
    const CzmSet occ = czm_nonzero_set(w, 0u);
    CzmSet K = czm_equal_set(w, 1u + own0), A = czm_equal_set(w, 2u + own0), R = czm_equal_set(w, 3u + own0);
    CzmSet B = czm_equal_set(w, 4u + own0), N = czm_equal_set(w, 5u + own0), P = czm_equal_set(w, 6u + own0);
    const CzmSet C = czm_equal_set(w, 7u + own0);
    const CzmSet EK = czm_equal_set(w, side ? 1u : 8u);           // the enemy king
    const CzmSet own = {K.a0 | A.a0 | R.a0 | B.a0 | N.a0 | P.a0 | C.a0, K.a1 | A.a1 | R.a1 | B.a1 | N.a1 | P.a1 | C.a1,
                        K.a2 | A.a2 | R.a2 | B.a2 | N.a2 | P.a2 | C.a2};
    const CzmSet enemy = {occ.a0 & ~own.a0, occ.a1 & ~own.a1, occ.a2 & ~own.a2};
    bool err = __builtin_popcount(own.a0) + __builtin_popcount(own.a1) + __builtin_popcount(own.a2) > 16;
    int count = 0;
    auto notown = [&](int q) { return !czm_tst(own, q); };      // validate_move, main.py:727: empty or enemy

    // ---- rooks and cannons (at most 2 + 2; a lane with fewer idles)
    CzmSet S = {R.a0 | C.a0, R.a1 | C.a1, R.a2 | C.a2};
#pragma unroll 1
    for (int it = 0; it < 4; ++it) {
        const int sq = czm_pop(S);
        if (sq < 0) continue;
        const int y = sq / 9, x = sq - y * 9;
        const bool cannon = czm_tst(C, sq);
        const uint32_t rd = czm_line_dests(czm_bits(occ, y * 9) & 0x1FFu, czm_bits(enemy, y * 9) & 0x1FFu, x, 9, cannon);
        const uint32_t fd = czm_line_dests(czm_file(occ, x), czm_file(enemy, x), y, 10, cannon);
        const uint32_t f = czm_ortho_field(rd, fd, x, y);
        count += __builtin_popcount(f);
        czm_or_field(or_word, T.base[sq], f);
    }
    // ---- knights (main.py:835-856): jump j lands on (x + dx, y + dy); the leg is the orthogonal neighbour on the long side
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
        const int sq = czm_pop(N);
        if (sq < 0) continue;
        const int y = sq / 9, x = sq - y * 9;
        const uint32_t on = T.knon[sq];
        uint32_t f = 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int dx = (j == 0 || j == 2) ? -2 : (j == 1 || j == 5) ? -1 : (j == 3 || j == 7) ? 1 : 2;
            const int dy = (j == 1 || j == 3) ? -2 : (j == 0 || j == 4) ? -1 : (j == 2 || j == 6) ? 1 : 2;
            const int leg = (dx == 2 || dx == -2) ? sq + dx / 2 : sq + (dy / 2) * 9;
            const bool onb = ((on >> j) & 1u) != 0u;
            const int q = onb ? sq + dy * 9 + dx : sq, lg = onb ? leg : sq;   // keep the bit tests on the board
            const bool ok = onb && notown(q) && !czm_tst(occ, lg);
            f |= (uint32_t)ok << __builtin_popcount(on & czm_low(j));
        }
        (void)x; (void)y;
        count += __builtin_popcount(f);
        czm_or_field(or_word, T.base[sq] + 17, f);
    }
    // ---- king (main.py:919-946) + the flying general (main.py:1097-1107: kings on one file, nothing between)
    {
        const int sq = czm_first(K);
        if (sq >= 0) {
            const int y = sq / 9, x = sq - y * 9;
            const int ylo = side ? 7 : 0, yhi = side ? 9 : 2;
            uint32_t rd = 0u, fd = 0u;
            if (x - 1 >= 3 && x - 1 <= 5 && y >= ylo && y <= yhi && notown(sq - 1)) rd |= 1u << (x - 1);
            if (x + 1 >= 3 && x + 1 <= 5 && y >= ylo && y <= yhi && notown(sq + 1)) rd |= 1u << (x + 1);
            if (y - 1 >= ylo && y - 1 <= yhi && x >= 3 && x <= 5 && notown(sq - 9)) fd |= 1u << (y - 1);
            if (y + 1 >= ylo && y + 1 <= yhi && x >= 3 && x <= 5 && notown(sq + 9)) fd |= 1u << (y + 1);
            const int eq = czm_first(EK);
            if (eq >= 0) {
                const int ey = eq / 9, ex = eq - ey * 9;
                if (ex == x) {
                    const int lo = y < ey ? y : ey, hi = y < ey ? ey : y;
                    const uint32_t between = czm_low(hi) & ~czm_low(lo + 1);
                    if ((czm_file(occ, x) & between) == 0u) fd |= 1u << ey;
                }
            }
            const uint32_t f = czm_ortho_field(rd, fd, x, y);
            count += __builtin_popcount(f);
            czm_or_field(or_word, T.base[sq], f);
        }
    }
    // ---- pawns (main.py:1063-1095): black advances to y-1, red to y+1; sideways once past the river
#pragma unroll 1
    for (int it = 0; it < 5; ++it) {
        const int sq = czm_pop(P);
        if (sq < 0) continue;
        const int y = sq / 9, x = sq - y * 9;
        uint32_t rd = 0u, fd = 0u;
        const int fy = side ? y - 1 : y + 1;
        if (fy >= 0 && fy <= 9 && notown(fy * 9 + x)) fd |= 1u << fy;
        if (side ? y < 5 : y > 4) {
            if (x + 1 <= 8 && notown(sq + 1)) rd |= 1u << (x + 1);
            if (x - 1 >= 0 && notown(sq - 1)) rd |= 1u << (x - 1);
        }
        const uint32_t f = czm_ortho_field(rd, fd, x, y);
        count += __builtin_popcount(f);
        czm_or_field(or_word, T.base[sq], f);
    }
    // ---- advisors (main.py:889-918: one diagonal step inside the palace) and bishops (main.py:857-888: two diagonal steps,
    //      the eye empty, own half of the board); their labels are the 48 literals at the end of the vocabulary
#pragma unroll
    for (int kind = 0; kind < 2; ++kind) {
        CzmSet &Q = kind ? B : A;
        const int st = kind ? 2 : 1;
        const int ylo = kind ? (side ? 5 : 0) : (side ? 7 : 0), yhi = kind ? (side ? 9 : 4) : (side ? 9 : 2);
        const int xlo = kind ? 0 : 3, xhi = kind ? 8 : 5;
#pragma unroll 1
        for (int it = 0; it < 2; ++it) {
            const int sq = czm_pop(Q);
            if (sq < 0) continue;
            const int y = sq / 9, x = sq - y * 9;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int sy = (d < 2 ? -1 : 1), sx = (d == 0 || d == 3) ? -1 : 1;
                const int ty = y + sy * st, tx = x + sx * st;
                const bool in = ty >= ylo && ty <= yhi && tx >= xlo && tx <= xhi;
                const int q = in ? ty * 9 + tx : sq, eye = in ? (y + sy) * 9 + x + sx : sq;
                const bool ok = in && notown(q) && (kind == 0 || !czm_tst(occ, eye));
                if (ok) {
                    const uint32_t l = T.ab[kind][sq * 4 + d];
                    if (l == 0xFFu) err = true;
                    else {
                        const int bit = CZM_NLIT_BASE + (int)l;
                        or_word(bit >> 5, 1u << (bit & 31));
                        count += 1;
                    }
                }
            }
        }
    }
    return err ? -1 : count;
}
