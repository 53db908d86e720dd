// cz_maskgen.h — K1m: the 2086-bit legal-move MASK of a position without the ordered move list.
//
// GameBoard.get_legal_moves (main.py:743-1109) returns an ordered list; what the search's expansion and a policy head
// consume is the SET (which labels are legal).  cz_movegen(moves = NULL) therefore does not need the reference's emission
// order, the staging rows, the prefix sums, the (src, dst) -> label LUT round trips or one LDS atomic per move of
// k_movegen (249 VALU + 156 SALU instructions per position: issue-bound at 7 % of the kernel's HBM roofline).  Here ONE LANE
// owns a position and everything is bit arithmetic in its registers:
//   * the 90 board bytes arrive as 23 dwords; the four BIT PLANES of the piece codes (v_and + v_dot4_u32_u8 with the byte
//     weights 1, 2, 4, ... 128) are 90-bit square sets, and occupancy, one set per piece kind of the side to move and the enemy
//     king are boolean functions of the planes;
//   * pieces are visited KIND BY KIND (all lanes run the rook code together, then the knight code, ...: no divergence on the
//     piece type); a piece's squares come off its kind's set by find-first-set;
//   * the move vocabulary (main.py:30-65) lists, per source square, 8 same-rank destinations, 9 same-file destinations and
//     the on-board knight jumps CONTIGUOUSLY, so the legal destinations of a rook / cannon / king / pawn are one <= 17-bit
//     field and a knight's one <= 8-bit field at a per-square base: a ray is (o - 2r) ^ o on the 9 / 10 line bits (and on their
//     bit reversal), never a loop over squares; a leaper's candidate squares are constant bit positions of a 64-square window
//     around it; advisor / bishop moves are the 48 literals at the end (a [square][direction] table);
//   * the results leave as exactly 15 (bit offset, field) pairs per position through the caller's emit(); count = popcount of
//     the fields.  k_movegen_mask (cz_rules.hip) stores them and builds the 66-word rows half a wave at a time.
// The rules are those of czd_gen_piece (cz_device.h), restated as set operations; tests/test_hip_rules.py holds both kernels
// to the same 4 381 golden positions and the 20 k oracle-checked corpus, and tests/maskgen_host.cpp (tests/test_maskgen_cpu.py) runs this very
// function on the CPU against the golden lists and the C oracle (the function is host-compilable on purpose).
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
    uint8_t pad_[2];       // 992 bytes = 62 x 16: a wave copies the tables to LDS with ONE 16-byte load per lane
};
static_assert(sizeof(CzmTables) == 992, "CzmTables is copied as 62 uint4");

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

struct CzmSet { uint64_t lo; uint32_t hi; };   // squares 0..63, 64..89 (two sources, not three: a three-way word select by square
                                                // index made hipcc put the sets into scratch and index them there)

CZM_FN uint32_t czm_dot4(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    return c + (a & 0xFF) * (b & 0xFF) + ((a >> 8) & 0xFF) * ((b >> 8) & 0xFF) + ((a >> 16) & 0xFF) * ((b >> 16) & 0xFF) + (a >> 24) * (b >> 24);
#endif
}
CZM_FN uint32_t czm_low(int n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << (n & 31)) - 1u); }   // bits 0 .. n-1, 0 <= n <= 32
CZM_FN uint32_t czm_bitrev32(uint32_t v) {
#if defined(__clang__)
    return __builtin_bitreverse32(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1); v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4); v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
#endif
}
CZM_FN int czm_ctz32(uint32_t v) { return __builtin_ctz(v); }   // v != 0
CZM_FN bool czm_tst(const CzmSet &s, int q) {   // 0 <= q < 90
    const uint32_t v = q < 64 ? (uint32_t)(s.lo >> (q & 63)) : s.hi >> (q & 31);
    return (v & 1u) != 0u;
}
// the 9 bits of rank y (squares 9 y .. 9 y + 8)
CZM_FN uint32_t czm_rank(const CzmSet &s, int y) {
    const int p = 9 * y;                              // 0 .. 81
    uint32_t v;
    if (p < 56) v = (uint32_t)(s.lo >> p);            // ranks 0..6: inside lo
    else if (p == 63) v = (uint32_t)(s.lo >> 63) | (s.hi << 1);   // rank 7: square 63 in lo, 64..71 in hi
    else v = s.hi >> ((p - 64) & 31);                 // ranks 8, 9
    return v & 0x1FFu;
}
// lowest / highest square of a set, -1 when it is empty — no loops: a kind has at most two pieces (pawns: five), so its squares
// are the set's lowest and highest bit, and a lane with fewer pieces just computes a field it then zeroes
CZM_FN int czm_lowest(const CzmSet &s) {
    return s.lo ? __builtin_ctzll(s.lo) : (s.hi ? 64 + __builtin_ctz(s.hi) : -1);
}
CZM_FN int czm_highest(const CzmSet &s) {
    return s.hi ? 95 - __builtin_clz(s.hi) : (s.lo ? 63 - __builtin_clzll(s.lo) : -1);
}
CZM_FN CzmSet czm_without(const CzmSet &s, int q) {   // q < 0: unchanged
    return CzmSet{s.lo & ~((q >= 0 && q < 64) ? 1ull << (q & 63) : 0ull), s.hi & ~(q >= 64 ? 1u << (q & 31) : 0u)};
}

// 64 consecutive squares of a set starting at square `start` (-24 <= start <= 89): bit i = square start + i, zero off the set.
// A piece's neighbourhood (knight: 39 squares around it, pawn / king: 19, advisor / bishop: 41) then sits at CONSTANT bit
// positions: one funnel shift per set instead of a bounds-clamped bit test per candidate square.
CZM_FN uint64_t czm_window(const CzmSet &s, int start) {
    const uint64_t up = (uint64_t)s.hi;
    const uint64_t a = s.lo << ((-start) & 63);                                              // start <= 0
    const uint64_t b = (s.lo >> (start & 63)) | ((up << 1) << ((63 - start) & 63));           // 0 < start < 64 (two shifts: never by 64)
    const uint64_t c = up >> ((start - 64) & 63);                                            // start >= 64
    return start <= 0 ? a : (start < 64 ? b : c);
}
CZM_FN uint32_t czm_wbit(uint64_t w, int pos) { return (uint32_t)(w >> pos) & 1u; }          // pos is a compile-time constant

// Bit plane b of the 90 piece codes (codes are <= 15) as a square set: bit b of every byte, eight squares per pair of dwords by
// two v_dot4_u32_u8 with the byte weights (1, 2, 4, 8) / (16, 32, 64, 128) — the bytes are 0 or 1 << b, so the packed byte
// comes out shifted left by b.  Four planes cost 4 x 72 instructions; every "byte == code" set of the position is then a few
// boolean operations on three registers (round 4: nine SWAR compares of all 23 dwords were a third of the kernel's VALU work).
template <int b>
CZM_FN CzmSet czm_plane(const uint32_t (&w)[23]) {
    const uint32_t m = 0x01010101u << b;
    uint32_t out[3] = {0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 23; k += 2) {
        uint32_t byte = czm_dot4(w[k] & m, 0x08040201u, 0u);
        if (k + 1 < 23) byte = czm_dot4(w[k + 1] & m, 0x80402010u, byte);
        out[k >> 3] |= (byte >> b) << ((k >> 1 & 3) * 8);   // squares 4k .. 4k+7 -> bits (4k & 31) ..
    }
    return CzmSet{(uint64_t)out[0] | ((uint64_t)out[1] << 32), out[2] & 0x03FFFFFFu};
}
CZM_FN CzmSet czm_and(const CzmSet &a, const CzmSet &b) { return CzmSet{a.lo & b.lo, a.hi & b.hi}; }
CZM_FN CzmSet czm_xor(const CzmSet &a, const CzmSet &b) { return CzmSet{a.lo ^ b.lo, a.hi ^ b.hi}; }
CZM_FN CzmSet czm_not(const CzmSet &a) { return CzmSet{~a.lo, ~a.hi & 0x03FFFFFFu}; }

// 10 bits of file x (bit r = square 9 r + x) of a set
CZM_FN uint32_t czm_file(const CzmSet &s, int x) {   // 0 <= x <= 8
    // t = s >> x, then the bits at the constant positions 9 r: 0 .. 63 in the low 64 bits, 72 and 81 = bits 8 and 17 above
    const uint64_t t = x ? (s.lo >> x) | ((uint64_t)s.hi << (64 - x)) : s.lo;
    const uint32_t t0 = (uint32_t)t, t1 = (uint32_t)(t >> 32), t2 = s.hi >> x;
    // bits 0, 9, 18, 27 of t0 are 1, 2, 4, 8 in its four bytes and bits 4, 13, 22, 31 of t1 are 16 .. 128 in its bytes: the sum of
    // the eight bytes (v_dot4 with weights 1) IS file bits 0..7; squares 72 and 81 are bits 8 and 17 of t2
    const uint32_t lo8 = czm_dot4(t1 & 0x80402010u, 0x01010101u, czm_dot4(t0 & 0x08040201u, 0x01010101u, 0u));
    return lo8 | (t2 & 0x100u) | ((t2 >> 8) & 0x200u);
}

// destinations of a rook / cannon at index p on a line of `len` squares: o = occupancy bits, e = enemy bits (main.py:757-833,
// 947-1062): the empty run in both directions; rook: + the first occupied square if it is an enemy; cannon: + the SECOND
// occupied square (behind exactly one screen) if it is an enemy
// o - 2r: subtracting twice the slider's bit borrows through the empty squares above it up to the first occupied one, so
// (o - 2r) ^ o is the run above r INCLUDING its first blocker (everything above when there is none); the run below comes from the
// same formula on the bit-reversed line (v_bfrev_b32).  Rook: runs & (empty | enemy).  Cannon: the empty part of the runs, plus —
// from the screen s = run & o — the second run (o - 2s) ^ o, whose occupied end is the target when it is an enemy.
CZM_FN uint32_t czm_rev32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__) || defined(__clang__)
    return __builtin_bitreverse32(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
#endif
}
template <bool cannon>
CZM_FN uint32_t czm_line_dests(uint32_t o, uint32_t e, int p, int len) {
    const uint32_t r = 1u << p, rr = 0x80000000u >> p, orv = czm_rev32(o);
    const uint32_t up = (o - (r << 1)) ^ o, dn = (orv - (rr << 1)) ^ orv;          // dn in the reversed domain
    uint32_t d;
    if (cannon) {
        const uint32_t s = up & o, sr = dn & orv;                                  // the screens (one bit or none)
        const uint32_t up2 = (o - (s << 1)) ^ o, dn2 = (orv - (sr << 1)) ^ orv;     // no screen: o ^ o = 0
        d = ((up | czm_rev32(dn)) & ~o) | ((up2 | czm_rev32(dn2)) & o & e);
    } else {
        d = (up | czm_rev32(dn)) & (~o | e);
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

// czm_position hands its results to emit(bit, field): OR `field` (<= 20 bits) into the position's 2086-bit mask at bit offset
// `bit` — EXACTLY 15 calls per position, in the same order on every lane (4 sliders, 2 knights, the king, 5 pawns, 3 words of
// advisor / bishop literals; a missing piece emits an empty field), so a caller may store the pairs instead of applying them.
#define CZM_EMITS 15
// czm_or_field: what an emit does to a row of 66 words when it is applied at once
template <typename OrWord>
CZM_FN void czm_or_field(OrWord &&or_word, int bit, uint32_t field) {
    const int wi = bit >> 5;
    const uint64_t v = (uint64_t)field << (bit & 31);    // one 64-bit shift; wi + 1 <= 64 for every base + 25-bit field
    or_word(wi, (uint32_t)v);
    or_word(wi + 1, (uint32_t)(v >> 32));
}

struct CzmNo { static constexpr bool value = false; };
struct CzmYes { static constexpr bool value = true; };

// ---- the pieces of a position, shared by the set form (czm_position) and the ordered list (czm_list) ------------------------
// w: the 90 board bytes (sq = y * 9 + x, code = 1 + index in "KARBNPCkarbnpc"; bytes 90, 91 of w[22] must be zero);
// side 0 = red ('w', codes 1..7, home ranks 0..4) to move, 1 = black.
struct CzmSets { CzmSet occ, own, enemy, K, A, R, B, N, C, P, EK; };

CZM_FN CzmSets czm_sets(const uint32_t (&w)[23], int side) {
    // own piece code = kind + (side ? 7 : 0), kind: K 1, A 2, R 3, B 4, N 5, P 6, C 7.  Bit planes p0..p3 of the codes; red's
    // own pieces have bit 3 clear and the kind in the low three bits, black's have bit 3 set and kind - 1 there: a bit-sliced
    // "+ 1 where black moves" (XOR / AND with the lane's side mask, no selects) makes the low bits the kind for both sides.
    const CzmSet p0 = czm_plane<0>(w), p1 = czm_plane<1>(w), p2 = czm_plane<2>(w), p3 = czm_plane<3>(w);
    CzmSets S;
    S.occ = CzmSet{p0.lo | p1.lo | p2.lo | p3.lo, p0.hi | p1.hi | p2.hi | p3.hi};
    const CzmSet sm = {side ? ~0ull : 0ull, side ? 0x03FFFFFFu : 0u};
    const CzmSet n0 = czm_xor(p0, sm), c0 = czm_and(p0, sm), n1 = czm_xor(p1, c0), c1 = czm_and(p1, c0), n2 = czm_xor(p2, c1);
    const CzmSet mine = czm_and(czm_xor(p3, czm_not(sm)), S.occ);    // bit 3 == side, not empty
    const CzmSet i0 = czm_not(n0), i1 = czm_not(n1), i2 = czm_not(n2);
    S.K = czm_and(mine, czm_and(n0, czm_and(i1, i2))); S.A = czm_and(mine, czm_and(i0, czm_and(n1, i2)));
    S.R = czm_and(mine, czm_and(n0, czm_and(n1, i2))); S.B = czm_and(mine, czm_and(i0, czm_and(i1, n2)));
    S.N = czm_and(mine, czm_and(n0, czm_and(i1, n2))); S.C = czm_and(mine, czm_and(n0, czm_and(n1, n2)));
    S.P = czm_and(mine, czm_and(i0, czm_and(n1, n2)));
    // the enemy king: code 8 (bit 3 only) when red moves, code 1 (bit 0 only) when black moves
    S.EK = czm_and(czm_and(czm_xor(p3, sm), czm_xor(p0, czm_not(sm))), czm_and(czm_not(p1), czm_not(p2)));
    S.own = CzmSet{S.K.lo | S.A.lo | S.R.lo | S.B.lo | S.N.lo | S.P.lo | S.C.lo, S.K.hi | S.A.hi | S.R.hi | S.B.hi | S.N.hi | S.P.hi | S.C.hi};
    S.enemy = CzmSet{S.occ.lo & ~S.own.lo, S.occ.hi & ~S.own.hi};
    return S;
}
// The generators below take a kind's squares as the LOWEST and HIGHEST bit of its set (pawns: five iterations): a board with a third
// rook / cannon / knight / advisor / bishop, a sixth pawn or a second king of the side to move is not a Xiangqi position, and the
// in-search generator (czd_wave_movegen: any board) would list moves this one silently drops — so such a board is an ERROR
// (count 0xFFFF, like a board with more than 16 own pieces), never a shorter list.
// All eight tests in one place, right behind czm_sets, with the result pinned by an opaque barrier: left to its own scheduling
// hipcc spread the popcounts over the function and took k_movegen_mask from 163 to 172 VGPRs (3 -> 2 waves per SIMD: -26 %) and
// k_movegen_list<false> from 156 to 178.
CZM_FN int czm_popcnt(const CzmSet &s) { return __builtin_popcountll(s.lo) + __builtin_popcount(s.hi); }
CZM_FN bool czm_not_a_set(const CzmSets &S) {   // all eight tests at once, the result pinned where it is computed
    int over = (int)(czm_popcnt(S.own) > 16) | (int)(czm_popcnt(S.R) > 2) | (int)(czm_popcnt(S.C) > 2) | (int)(czm_popcnt(S.N) > 2) |
               (int)(czm_popcnt(S.A) > 2) | (int)(czm_popcnt(S.B) > 2) | (int)(czm_popcnt(S.P) > 5) | (int)(czm_popcnt(S.K) > 1);
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(over));
#endif
    return over != 0;
}
// rook / cannon on square q (main.py:757-833, 947-1062): the 17-bit field of its destinations
template <bool cannon>
CZM_FN uint32_t czm_slider_field(const CzmSets &S, int q) {
    const int y = q / 9, x = q - y * 9;
    const uint32_t rd = czm_line_dests<cannon>(czm_rank(S.occ, y), czm_rank(S.enemy, y), x, 9);
    const uint32_t fd = czm_line_dests<cannon>(czm_file(S.occ, x), czm_file(S.enemy, x), y, 10);
    return czm_ortho_field(rd, fd, x, y);
}
// knight on q (main.py:835-856): bit j = jump j of the vocabulary order ((dx,dy) = (-2,-1) (-1,-2) (-2,1) (1,-2) (2,-1) (-1,2) (2,1)
// (1,2)) is legal.  The window starts at q - 19: jump j lands on bit 19 + 9 dy + dx, its leg is bit 18 / 20 (dx = -+2) or 10 / 28
// (dy = -+2); squares off the board read as free and empty — `on` has no bit for a jump that leaves the board
CZM_FN uint32_t czm_knight_good(const CzmSets &S, int q, uint32_t on) {
    const uint64_t fr = ~czm_window(S.own, q - 19), em = ~czm_window(S.occ, q - 19);
    const uint32_t tg = czm_wbit(fr, 8) | (czm_wbit(fr, 0) << 1) | (czm_wbit(fr, 26) << 2) | (czm_wbit(fr, 2) << 3) |
                        (czm_wbit(fr, 12) << 4) | (czm_wbit(fr, 36) << 5) | (czm_wbit(fr, 30) << 6) | (czm_wbit(fr, 38) << 7);
    const uint32_t lg = (czm_wbit(em, 18) * 0x05u) | (czm_wbit(em, 10) * 0x0Au) | (czm_wbit(em, 20) * 0x50u) | (czm_wbit(em, 28) * 0xA0u);
    return tg & lg & on;
}
// king on q (main.py:919-946): the field of its steps inside the palace; *fg = the file-field bit of the flying general
// (main.py:1097-1107: kings on one file, nothing between), which the reference appends to its list LAST
CZM_FN uint32_t czm_king_field(const CzmSets &S, int side, int q, int eq, uint32_t *fg) {
    const int y = q / 9, x = q - y * 9;
    const int ylo = side ? 7 : 0, yhi = side ? 9 : 2;
    const bool iny = (y >= ylo) & (y <= yhi), inx = (x >= 3) & (x <= 5);
    uint32_t rd = 0u, fd = 0u;
    const uint64_t fr = ~czm_window(S.own, q - 9);   // bit 0: q - 9, 8: q - 1, 10: q + 1, 18: q + 9
    rd |= ((uint32_t)(iny & (x >= 4) & (x <= 6)) & czm_wbit(fr, 8)) << ((x + 8) % 9);      // x - 1, kept in range
    rd |= ((uint32_t)(iny & (x >= 2) & (x <= 4)) & czm_wbit(fr, 10)) << ((x + 1) % 9);
    fd |= ((uint32_t)(inx & (y - 1 >= ylo) & (y - 1 <= yhi)) & czm_wbit(fr, 0)) << ((y + 9) % 10);   // y - 1
    fd |= ((uint32_t)(inx & (y + 1 >= ylo) & (y + 1 <= yhi)) & czm_wbit(fr, 18)) << ((y + 1) % 10);
    const int e = eq >= 0 ? eq : 0;
    const int ey = e / 9, ex = e - ey * 9;
    const int lo = y < ey ? y : ey, hi = y < ey ? ey : y;
    const uint32_t between = czm_low(hi) & ~czm_low(lo + 1);
    const uint32_t fgd = (uint32_t)((eq >= 0) & (ex == x) & ((czm_file(S.occ, x) & between) == 0u)) << ey;
    *fg = czm_ortho_field(0u, fgd, x, y);
    return czm_ortho_field(rd, fd, x, y);
}
// pawn on q (main.py:1063-1095): black advances to y - 1, red to y + 1; sideways once past the river.  The field directly
// (czm_ortho_field's squeeze of the own file / rank done by hand): the left neighbour is field bit x - 1, the right one bit x;
// the forward square is file bit y for red (y + 1 with rank y squeezed out), y - 1 for black
CZM_FN uint32_t czm_pawn_field(const CzmSets &S, int side, int q) {
    const int y = q / 9, x = q - y * 9;
    const int fy = side ? y - 1 : y + 1;
    const bool fin = (fy >= 0) & (fy <= 9), river = side ? y < 5 : y > 4;
    const uint64_t fr = ~czm_window(S.own, q - 9);   // bit 0: q - 9, 8: q - 1, 10: q + 1, 18: q + 9
    const uint32_t lr = ((uint32_t)(river & (x >= 1)) & czm_wbit(fr, 8)) | (((uint32_t)(river & (x <= 7)) & czm_wbit(fr, 10)) << 1);
    const uint32_t fwd = (uint32_t)fin & (side ? czm_wbit(fr, 0) : czm_wbit(fr, 18));
    const int gp = side ? (y > 0 ? y - 1 : 0) : y;
    return ((lr << x) >> 1) | (fwd << (gp + 8));
}
// All pawns at once (round 5): which pawns may step left / right / forward as three SETS — the pawn set
// shifted against the own-piece set — so that a pawn's three move bits are three bit tests instead of a 64-square window, its
// bounds and river tests and the extraction per pawn (czm_pawn_field above: ~90 instructions per pawn, five pawns per position).
struct CzmPawnSets { CzmSet l, r, f; };
CZM_FN CzmSet czm_shl(const CzmSet &s, int n) {   // 0 < n < 32: square q -> q + n
    return CzmSet{s.lo << n, ((s.hi << n) | (uint32_t)(s.lo >> (64 - n))) & 0x03FFFFFFu};
}
CZM_FN CzmSet czm_shr(const CzmSet &s, int n) {   // 0 < n < 32: square q -> q - n
    return CzmSet{(s.lo >> n) | ((uint64_t)s.hi << (64 - n)), s.hi >> n};
}
CZM_FN CzmPawnSets czm_pawn_sets(const CzmSets &S, int side) {
    // files 0 / 8 as sets: bit 9 r + x
    const CzmSet f0 = {0x8040201008040201ull, 0x00020100u}, f8 = {0x4020100804020100ull, 0x02010080u};   // squares 0, 9, .. 81 / 8, 17, .. 89
    const CzmSet river = side ? CzmSet{(1ull << 45) - 1ull, 0u} : CzmSet{~((1ull << 45) - 1ull), 0x03FFFFFFu};   // black: y < 5, red: y > 4
    const CzmSet pr = czm_and(S.P, river);
    CzmPawnSets o;
    // left: q - 1 free of own pieces, x >= 1;  right: q + 1, x <= 7
    o.l = CzmSet{pr.lo & ~f0.lo & ~(S.own.lo << 1), pr.hi & ~f0.hi & ~((S.own.hi << 1) | (uint32_t)(S.own.lo >> 63))};
    o.r = CzmSet{pr.lo & ~f8.lo & ~((S.own.lo >> 1) | ((uint64_t)S.own.hi << 63)), pr.hi & ~f8.hi & ~(S.own.hi >> 1)};
    // forward: red to q + 9 (exists for y <= 8), black to q - 9 (y >= 1)
    const CzmSet up = czm_shr(S.own, 9), dn = czm_shl(S.own, 9);       // own piece on q + 9 / on q - 9, as a property of q
    const CzmSet fr = side ? CzmSet{S.P.lo & ~dn.lo & ~0x1FFull, S.P.hi & ~dn.hi} : CzmSet{S.P.lo & ~up.lo, S.P.hi & ~up.hi & 0x0001FFFFu};
    o.f = fr;
    return o;
}
// the pawn on q: its field from the three sets (same layout as czm_pawn_field)
CZM_FN uint32_t czm_pawn_field2(const CzmPawnSets &PS, int side, int q) {
    const int y = q / 9, x = q - y * 9;
    const bool low = q < 64;
    const int sh = q & (low ? 63 : 31);
    const uint32_t l = (uint32_t)((low ? PS.l.lo >> sh : (uint64_t)(PS.l.hi >> sh)) & 1ull);
    const uint32_t r = (uint32_t)((low ? PS.r.lo >> sh : (uint64_t)(PS.r.hi >> sh)) & 1ull);
    const uint32_t f = (uint32_t)((low ? PS.f.lo >> sh : (uint64_t)(PS.f.hi >> sh)) & 1ull);
    const int gp = side ? (y > 0 ? y - 1 : 0) : y;
    return (((l | (r << 1)) << x) >> 1) | (f << (gp + 8));
}
// advisor (kind 0; main.py:889-918: one diagonal step inside the palace) / bishop (kind 1; main.py:857-888: two diagonal steps,
// the eye empty, own half of the board) on q: bit d = direction d ((dy,dx) = (-s,-s) (-s,+s) (+s,+s) (+s,-s)) is legal.  The
// window starts at q - 20: the target of direction d is bit 20 + st (9 sy + sx), a bishop's eye bit 20 + 9 sy + sx
template <int kind>
CZM_FN uint32_t czm_diag_good(const CzmSets &S, int side, int q) {
    const int st = kind ? 2 : 1;
    const int ylo = kind ? (side ? 5 : 0) : (side ? 7 : 0), yhi = kind ? (side ? 9 : 4) : (side ? 9 : 2);
    const int xlo = kind ? 0 : 3, xhi = kind ? 8 : 5;
    const int y = q / 9, x = q - y * 9;
    const uint64_t fr = ~czm_window(S.own, q - 20), em = ~czm_window(S.occ, q - 20);
    uint32_t good = 0u;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int sy = (d < 2 ? -1 : 1), sx = (d == 0 || d == 3) ? -1 : 1;
        const int ty = y + sy * st, tx = x + sx * st;
        const bool in = (ty >= ylo) & (ty <= yhi) & (tx >= xlo) & (tx <= xhi);
        const uint32_t free_t = kind ? czm_wbit(fr, 20 + 2 * (9 * sy + sx)) : czm_wbit(fr, 20 + 9 * sy + sx);
        const uint32_t open_eye = kind ? czm_wbit(em, 20 + 9 * sy + sx) : 1u;
        good |= ((uint32_t)in & free_t & open_eye) << d;
    }
    return good;
}

// ---- the SET: czm_position.  Returns the number of legal moves, or -1 when the position is not a Xiangqi position the
// vocabulary can express (more than 16 pieces of the side to move, or more of a kind than a Xiangqi set holds — czm_not_a_set;
// an advisor / bishop move without a label).  Branch-free apart from
// the loops: every lane runs every kind's code; a missing piece (square -1) computes on square 0 and its field is zeroed before
// it is handed out.
template <typename Emit>
CZM_FN int czm_position(const uint32_t (&w)[23], int side, const CzmTables &T, Emit emit) {
    const CzmSets S = czm_sets(w, side);
    bool err = czm_not_a_set(S);
    int count = 0;
    auto put = [&](int bit, uint32_t f, bool ok) {
        f = ok ? f : 0u;
        count += __builtin_popcount(f);
        emit(bit, f);
    };
    {   // rooks, then cannons: at most two of each — the kind's lowest and highest square
        const int r0 = czm_lowest(S.R), r1 = czm_highest(S.R), c0 = czm_lowest(S.C), c1 = czm_highest(S.C);
#pragma unroll 1
        for (int it = 0; it < 2; ++it) { const int sq = it ? r1 : r0; const bool ok = it ? r1 > r0 : r0 >= 0; const int q = ok ? sq : 0; put(T.base[q], czm_slider_field<false>(S, q), ok); }
#pragma unroll 1
        for (int it = 0; it < 2; ++it) { const int sq = it ? c1 : c0; const bool ok = it ? c1 > c0 : c0 >= 0; const int q = ok ? sq : 0; put(T.base[q], czm_slider_field<true>(S, q), ok); }
    }
    {   // knights: the vocabulary lists on-board jumps only
        const int n0 = czm_lowest(S.N), n1 = czm_highest(S.N);
#pragma unroll 1
        for (int it = 0; it < 2; ++it) {
            const int sq = it ? n1 : n0; const bool ok = it ? n1 > n0 : n0 >= 0; const int q = ok ? sq : 0;
            const uint32_t on = T.knon[q], good = czm_knight_good(S, q, on);
            uint32_t f = 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) f |= ((good >> j) & 1u) << __builtin_popcount(on & czm_low(j));
            put(T.base[q] + 17, f, ok);
        }
    }
    {   // king + the flying general
        const int sq = czm_lowest(S.K), eq = czm_lowest(S.EK);
        const bool ok = sq >= 0;
        const int q = ok ? sq : 0;
        uint32_t fg;
        const uint32_t f = czm_king_field(S, side, q, eq, &fg);
        put(T.base[q], f | fg, ok);
    }
    {   // pawns: at most five; their move bits come from three sets computed once (czm_pawn_sets)
        const CzmPawnSets PS = czm_pawn_sets(S, side);
        CzmSet P = S.P;
#pragma unroll 1
        for (int it = 0; it < 5; ++it) {
            const int sq = czm_lowest(P);
            P = czm_without(P, sq);
            const bool ok = sq >= 0;
            const int q = ok ? sq : 0;
            put(T.base[q], czm_pawn_field2(PS, side, q), ok);
        }
    }
    uint64_t lits = 0ull;   // the 48 advisor / bishop literals of the position (labels 2038 + l)
    auto literal = [&](int sq, bool ok, auto kind_tag) {
        constexpr int kind = decltype(kind_tag)::value ? 1 : 0;
        const int q = ok ? sq : 0;
        const uint32_t good = ok ? czm_diag_good<kind>(S, side, q) : 0u;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t l = T.ab[kind][q * 4 + d];
            const bool g = ((good >> d) & 1u) != 0u;
            err |= g & (l == 0xFFu);
            lits |= (g & (l != 0xFFu)) ? 1ull << (l & 63u) : 0ull;
        }
    };
    {
        const int a0 = czm_lowest(S.A), a1 = czm_highest(S.A), b0 = czm_lowest(S.B), b1 = czm_highest(S.B);
#pragma unroll 1
        for (int it = 0; it < 2; ++it) literal(it ? a1 : a0, it ? a1 > a0 : a0 >= 0, CzmNo{});
#pragma unroll 1
        for (int it = 0; it < 2; ++it) literal(it ? b1 : b0, it ? b1 > b0 : b0 >= 0, CzmYes{});
    }
    {   // labels 2038 .. 2085, sixteen per emit
        count += __builtin_popcountll(lits);
        emit(CZM_NLIT_BASE, (uint32_t)lits & 0xFFFFu);
        emit(CZM_NLIT_BASE + 16, (uint32_t)(lits >> 16) & 0xFFFFu);
        emit(CZM_NLIT_BASE + 32, (uint32_t)(lits >> 32) & 0xFFFFu);
    }
    return err ? -1 : count;
}

// ---- the ORDERED LIST: czm_list — GameBoard.get_legal_moves' list (main.py:743-1109) as labels, in the reference's order:
// pieces in scan order (ascending square: y outer, x inner, main.py:754-755); a rook / cannon its four rays -x, +x, -y, +y, each
// from the piece outwards (for a cannon the capture behind the screen ends its ray); a knight (2i, j) then (i, 2j) for i, j in
// (-1, +1)^2; the king x - 1, x + 1, y - 1, y + 1; a pawn forward, x + 1, x - 1; advisors / bishops their four diagonals (-,-)
// (-,+) (+,+) (+,-); the flying general LAST (main.py:1097-1107).  One lane = one position here too, and no divergence on the
// piece kind: the pieces are generated KIND BY KIND into 16 payload registers (the same fields as czm_position's), their counts
// are summed in SQUARE order through 16 words of per-position scratch (a piece's rank = the number of own pieces below its
// square), and then every piece writes its moves at its offset — label = the square's base + the index of the field bit.
//   put(m, label, b): b = 1: `label` is move number n = 128 + m / 2 of the list — m = 2 (n - 128), the byte offset of the n-th 16-bit
//     slot from slot CZM_IGNORE_SLOT = 128; b = 0: not a move.  A caller stores at byte offset b * m from slot 128 without a select or a
//     branch (one multiply-add): slot 128 collects what is not a move (a position has at most 120).  put returns the next place,
//     m + 2 b (its own instruction in the kernel: left to the compiler the running place becomes a count that is scaled per move);
//   scr(i): the i-th of 17 scratch words (i varies per lane);
//   mid(): called once between the last use of the scratch and the first put (a caller may keep both in the same memory);
//   emit(bit, field): the position's SET as well — the same 15 pairs, in the same order, as czm_position hands out
// Returns the number of moves, or -1 like czm_position.
#if defined(__HIP_DEVICE_COMPILE__)
#define CZM_ANY(c) (__ballot(c) != 0ull)
#else
#define CZM_ANY(c) (c)
#endif
#define CZM_IGNORE_SLOT 128
// own pieces on squares < q: the set as three words o0, o1, o2 with p1 = popcount(o0), p2 = p1 + popcount(o1) — the word of q, the
// pieces in the words below it, one mask (16 calls per position: the two 64-bit masks of the straightforward form were 15 VALU each)
CZM_FN int czm_rank_below(uint32_t o0, uint32_t o1, uint32_t o2, int p1, int p2, int q) {
    const int wi = q >> 5;
    const uint32_t word = wi == 0 ? o0 : (wi == 1 ? o1 : o2);
    const int below = wi == 0 ? 0 : (wi == 1 ? p1 : p2);
    return below + __builtin_popcount(word & ((1u << (q & 31)) - 1u));
}
template <typename Put, typename Scr, typename Mid, typename Emit>
CZM_FN int czm_list(const uint32_t (&w)[23], int side, const CzmTables &T, Put put, Scr scr, Mid mid, Emit emit) {
    const CzmSets S = czm_sets(w, side);
    bool err = czm_not_a_set(S);
    // slots in kind order: 0, 1 rooks; 2, 3 cannons; 4, 5 knights; 6 king; 7 .. 11 pawns; 12, 13 advisors; 14, 15 bishops
    int q[16];
    uint32_t pay[16];
    auto slot = [&](int s, int sq, bool ok, uint32_t p) { q[s] = ok ? sq : 0; pay[s] = ok ? p : 0u; };
    {
        const int r0 = czm_lowest(S.R), r1 = czm_highest(S.R), c0 = czm_lowest(S.C), c1 = czm_highest(S.C);
        slot(0, r0, r0 >= 0, czm_slider_field<false>(S, r0 >= 0 ? r0 : 0));
        slot(1, r1, r1 > r0, czm_slider_field<false>(S, r1 > r0 ? r1 : 0));
        slot(2, c0, c0 >= 0, czm_slider_field<true>(S, c0 >= 0 ? c0 : 0));
        slot(3, c1, c1 > c0, czm_slider_field<true>(S, c1 > c0 ? c1 : 0));
        const int n0 = czm_lowest(S.N), n1 = czm_highest(S.N);
        slot(4, n0, n0 >= 0, czm_knight_good(S, n0 >= 0 ? n0 : 0, T.knon[n0 >= 0 ? n0 : 0]));
        slot(5, n1, n1 > n0, czm_knight_good(S, n1 > n0 ? n1 : 0, T.knon[n1 > n0 ? n1 : 0]));
    }
    uint32_t fg = 0u;
    const int kq = czm_lowest(S.K);
    {
        uint32_t f = czm_king_field(S, side, kq >= 0 ? kq : 0, czm_lowest(S.EK), &fg);
        slot(6, kq, kq >= 0, f);
        fg = kq >= 0 ? fg : 0u;
    }
    {
        CzmSet P = S.P;
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            const int sq = czm_lowest(P);
            P = czm_without(P, sq);
            slot(7 + it, sq, sq >= 0, czm_pawn_field(S, side, sq >= 0 ? sq : 0));
        }
    }
    {
        const int a0 = czm_lowest(S.A), a1 = czm_highest(S.A), b0 = czm_lowest(S.B), b1 = czm_highest(S.B);
        slot(12, a0, a0 >= 0, czm_diag_good<0>(S, side, a0 >= 0 ? a0 : 0));
        slot(13, a1, a1 > a0, czm_diag_good<0>(S, side, a1 > a0 ? a1 : 0));
        slot(14, b0, b0 >= 0, czm_diag_good<1>(S, side, b0 >= 0 ? b0 : 0));
        slot(15, b1, b1 > b0, czm_diag_good<1>(S, side, b1 > b0 ? b1 : 0));
    }
    {   // the set: czm_position's 15 (bit, field) pairs from the same payloads
#pragma unroll
        for (int s = 0; s < 4; ++s) emit((int)T.base[q[s]], pay[s]);
#pragma unroll
        for (int s = 4; s < 6; ++s) {
            const uint32_t on = T.knon[q[s]];
            uint32_t f = 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) f |= ((pay[s] >> j) & 1u) << __builtin_popcount(on & czm_low(j));
            emit((int)T.base[q[s]] + 17, f);
        }
        emit((int)T.base[q[6]], pay[6] | fg);
#pragma unroll
        for (int s = 7; s < 12; ++s) emit((int)T.base[q[s]], pay[s]);
        uint64_t lits = 0ull;
#pragma unroll
        for (int s = 12; s < 16; ++s)
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t l = T.ab[s >= 14 ? 1 : 0][q[s] * 4 + d];
                lits |= (((pay[s] >> d) & 1u) != 0u && l != 0xFFu) ? 1ull << (l & 63u) : 0ull;
            }
        emit(CZM_NLIT_BASE, (uint32_t)lits & 0xFFFFu);
        emit(CZM_NLIT_BASE + 16, (uint32_t)(lits >> 16) & 0xFFFFu);
        emit(CZM_NLIT_BASE + 32, (uint32_t)(lits >> 32) & 0xFFFFu);
    }
    // the counts in square order: scratch word r collects the count of the piece of rank r (a missing piece adds 0 to word 0)
    // (the pieces' ranks are distinct, so a piece WRITES its count — a piece without moves, or a missing one, writes word 16
    // instead: 16 independent stores, then 16 loads, the sums in registers, 16 stores, 16 loads by rank; a read-modify-write per
    // piece was 16 LDS round trips one after the other)
    int rk[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) scr(s) = 0u;
    {
        const uint32_t o0 = (uint32_t)S.own.lo, o1 = (uint32_t)(S.own.lo >> 32), o2 = S.own.hi;
        const int p1 = __builtin_popcount(o0), p2 = p1 + __builtin_popcount(o1);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const uint32_t c = (uint32_t)__builtin_popcount(pay[s]);
            rk[s] = czm_rank_below(o0, o1, o2, p1, p2, q[s]) & 15;
            scr(c ? rk[s] : 16) = c;
        }
    }
    // From here on a place in the list is m = 2 * (n - CZM_IGNORE_SLOT): the byte offset of the n-th 16-bit slot from slot 128
    // (negative; see put above) — the prefix sums are kept in that form
    int mtot = -2 * CZM_IGNORE_SLOT;
    {
#pragma unroll
        for (int h = 0; h < 16; h += 8) {   // eight words at a time: sixteen live values cost the list-only kernel its 168-register budget
            uint32_t t[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) t[r] = scr(h + r);
#pragma unroll
            for (int r = 0; r < 8; ++r) { const int c = (int)t[r]; t[r] = (uint32_t)mtot; mtot += 2 * c; }
#pragma unroll
            for (int r = 0; r < 8; ++r) scr(h + r) = t[r];
        }
    }
    int off[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) off[s] = (int)scr(rk[s]);
    mid();
    // every piece writes its moves at its offset.  Round 6: no loops over set bits (a `while any lane has a bit left` loop ran as
    // long as the wave's richest position, ~140 iterations of ~12 instructions per group of 64 positions = 44 % of the list
    // kernel): every CANDIDATE of a piece is visited once, in the reference's order, with put(m, label, is-it-a-move).  For a
    // rook / cannon the eight same-rank candidates in emission order are bit p of  perm = reverse(field below x) | field from x up
    // (p < x: file x - 1 - p, the -x ray from the piece outwards; p >= x: bit p, the +x ray), their label offsets the nibbles of a
    // word made from x alone; the nine same-file candidates likewise from y.
    auto cand = [&](uint32_t b, int label, int &m) { m = put(m, label, b); };
    auto order_bits = [](uint32_t f, int k) -> uint32_t {   // bits below k reversed (bit p <- bit k - 1 - p), bits from k up in place
        const uint32_t lo = f & czm_low(k);
        return (czm_bitrev32(lo) >> ((32 - k) & 31)) | (f & ~czm_low(k));   // k = 0: lo = 0
    };
    auto order_nibbles = [](int k) -> uint32_t {            // nibble p = k - 1 - p below k, p from k up (k <= 8)
        const uint32_t m = k >= 8 ? 0xFFFFFFFFu : ((1u << (4 * k)) - 1u);
        return (0x76543210u & ~m) | (k ? (0x01234567u >> (4 * (8 - k))) : 0u);
    };
    auto slider = [&](int s) {    // -x, +x, -y, +y, each from the piece outwards
        const int y = q[s] / 9, x = q[s] - y * 9, base = T.base[q[s]];
        const uint32_t px = order_bits(pay[s] & 0xFFu, x), py = order_bits(pay[s] >> 8, y);
        const uint32_t wx = order_nibbles(x), wy = y == 9 ? 0x12345678u : order_nibbles(y);
        int m = off[s];
#pragma unroll
        for (int p = 0; p < 8; ++p) cand((px >> p) & 1u, base + (int)((wx >> (4 * p)) & 15u), m);
#pragma unroll
        for (int p = 0; p < 8; ++p) cand((py >> p) & 1u, base + 8 + (int)((wy >> (4 * p)) & 15u), m);
        cand((py >> 8) & 1u, base + 8 + (y == 9 ? 0 : 8), m);
    };
    auto ortho = [&](int s, bool pawn) {
        const int y = q[s] / 9, x = q[s] - y * 9, base = T.base[q[s]];
        const uint32_t rkf = pay[s] & 0xFFu, flf = pay[s] >> 8;
        const uint32_t xm = ((rkf << 1) >> x) & 1u, xp = (rkf >> x) & 1u;   // field bit x - 1 (none at x = 0), field bit x
        int m = off[s];
        if (pawn) {   // forward (the one bit of the file field), x + 1 (rank-field bit x), x - 1 (bit x - 1)
            cand(flf != 0u ? 1u : 0u, base + 8 + (flf ? czm_ctz32(flf) : 0), m);
            cand(xp, base + x, m);
            cand(xm, base + x - 1, m);
        } else {      // the king: x - 1, x + 1, y - 1, y + 1
            cand(xm, base + x - 1, m);
            cand(xp, base + x, m);
            cand(((flf << 1) >> y) & 1u, base + 8 + y - 1, m);
            cand((flf >> y) & 1u, base + 8 + y, m);
        }
    };
#pragma unroll
    for (int s = 0; s < 4; ++s) slider(s);
#pragma unroll
    for (int s = 4; s < 6; ++s) {   // knights: (2i, j) then (i, 2j) for i, j in (-1, +1)^2 = vocabulary jumps 1, 0, 3, 4, 5, 2, 7, 6
        const uint32_t on = T.knon[q[s]];
        const int base = T.base[q[s]] + 17;
        int m = off[s];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const int j = o == 0 ? 1 : o == 1 ? 0 : o == 2 ? 3 : o == 3 ? 4 : o == 4 ? 5 : o == 5 ? 2 : o == 6 ? 7 : 6;
            cand((pay[s] >> j) & 1u, base + __builtin_popcount(on & czm_low(j)), m);
        }
    }
    ortho(6, false);
#pragma unroll
    for (int s = 7; s < 12; ++s) ortho(s, true);
#pragma unroll
    for (int s = 12; s < 16; ++s) {
        const int kind = s >= 14 ? 1 : 0;
        int m = off[s];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t l = T.ab[kind][q[s] * 4 + d];
            const uint32_t g = (pay[s] >> d) & 1u;
            err |= (g != 0u) & (l == 0xFFu);
            cand(g, CZM_NLIT_BASE + (int)(l & 63u), m);
        }
    }
    // the flying general, last
    cand(fg != 0u ? 1u : 0u, (int)T.base[kq >= 0 ? kq : 0] + (fg ? __builtin_ctz(fg) : 0), mtot);
    const int total = (mtot >> 1) + CZM_IGNORE_SLOT;
    return err ? -1 : total;
}
