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

// Per-lane generator for the piece on `sq` (if it belongs to `side`), in the exact emission
// order of GameBoard.get_legal_moves (main.py:757-1095).  Writes labels to st[0..n).
__device__ __forceinline__ int czd_gen_piece(const uint8_t *b, int sq, int side, const int16_t *lut,
                                             uint16_t *st, bool &err) {
    const int c = b[sq];
    const int black = c > 7;
    if (c == 0 || black != side) return 0;
    const int t = black ? c - 7 : c;
    const int y = sq / 9, x = sq - y * 9;
    int n = 0;
    auto emit = [&](int ty, int tx) {
        int l = lut[sq * CZD_NSQ + ty * 9 + tx];
        if (l < 0) err = true; else st[n++] = (uint16_t)l;
    };
    auto enemy = [&](int d) { return d != 0 && ((d > 7) != side); };
    auto notown = [&](int d) { return d == 0 || ((d > 7) != side); };  // validate_move, main.py:727
    auto inb = [&](int ty, int tx) { return ty >= 0 && tx >= 0 && ty < 10 && tx < 9; };  // check_bounds :717
    switch (t) {
    case 3: {  // R/r  main.py:757-833: -x, +x, -y, +y
        for (int tx = x - 1; tx >= 0; --tx) { int d = b[y * 9 + tx]; if (d) { if (enemy(d)) emit(y, tx); break; } emit(y, tx); }
        for (int tx = x + 1; tx < 9; ++tx) { int d = b[y * 9 + tx]; if (d) { if (enemy(d)) emit(y, tx); break; } emit(y, tx); }
        for (int ty = y - 1; ty >= 0; --ty) { int d = b[ty * 9 + x]; if (d) { if (enemy(d)) emit(ty, x); break; } emit(ty, x); }
        for (int ty = y + 1; ty < 10; ++ty) { int d = b[ty * 9 + x]; if (d) { if (enemy(d)) emit(ty, x); break; } emit(ty, x); }
    } break;
    case 5: {  // N/n  main.py:835-856: (2i,j) with leg (i,0), then (i,2j) with leg (0,j)
        for (int i = -1; i <= 1; i += 2)
            for (int j = -1; j <= 1; j += 2) {
                int ty = y + 2 * i, tx = x + j;
                if (inb(ty, tx) && notown(b[ty * 9 + tx]) && b[(y + i) * 9 + x] == 0) emit(ty, tx);
                ty = y + i; tx = x + 2 * j;
                if (inb(ty, tx) && notown(b[ty * 9 + tx]) && b[y * 9 + x + j] == 0) emit(ty, tx);
            }
    } break;
    case 4: {  // B/b  main.py:857-888: two-step diagonals, eye empty, own half
        for (int i = -2; i <= 2; i += 4) {
            const int h = i / 2;
            int ty = y + i, tx = x + i;
            if (inb(ty, tx) && notown(b[ty * 9 + tx]) && (side ? ty >= 5 : ty <= 4) && b[(y + h) * 9 + x + h] == 0) emit(ty, tx);
            tx = x - i;
            if (inb(ty, tx) && notown(b[ty * 9 + tx]) && (side ? ty >= 5 : ty <= 4) && b[(y + h) * 9 + x - h] == 0) emit(ty, tx);
        }
    } break;
    case 2: {  // A/a  main.py:889-918: palace diagonals
        for (int i = -1; i <= 1; i += 2) {
            int ty = y + i, tx = x + i;
            if (inb(ty, tx) && notown(b[ty * 9 + tx]) && (side ? ty >= 7 : ty <= 2) && tx >= 3 && tx <= 5) emit(ty, tx);
            tx = x - i;
            if (inb(ty, tx) && notown(b[ty * 9 + tx]) && (side ? ty >= 7 : ty <= 2) && tx >= 3 && tx <= 5) emit(ty, tx);
        }
    } break;
    case 1: {  // K/k  main.py:919-946: (0,-1) (0,+1) (-1,0) (+1,0) inside the palace
        for (int i = 0; i < 2; ++i)
            for (int s = -1; s <= 1; s += 2) {
                int ty = y + i * s, tx = x + (1 - i) * s;
                if (inb(ty, tx) && notown(b[ty * 9 + tx]) && (side ? ty >= 7 : ty <= 2) && tx >= 3 && tx <= 5) emit(ty, tx);
            }
    } break;
    case 7: {  // C/c  main.py:947-1062: slide over empties; after one screen capture the first enemy
        bool hit = false;
        for (int tx = x - 1; tx >= 0; --tx) { int d = b[y * 9 + tx]; if (!hit) { if (d) hit = true; else emit(y, tx); } else if (d) { if (enemy(d)) emit(y, tx); break; } }
        hit = false;
        for (int tx = x + 1; tx < 9; ++tx) { int d = b[y * 9 + tx]; if (!hit) { if (d) hit = true; else emit(y, tx); } else if (d) { if (enemy(d)) emit(y, tx); break; } }
        hit = false;
        for (int ty = y - 1; ty >= 0; --ty) { int d = b[ty * 9 + x]; if (!hit) { if (d) hit = true; else emit(ty, x); } else if (d) { if (enemy(d)) emit(ty, x); break; } }
        hit = false;
        for (int ty = y + 1; ty < 10; ++ty) { int d = b[ty * 9 + x]; if (!hit) { if (d) hit = true; else emit(ty, x); } else if (d) { if (enemy(d)) emit(ty, x); break; } }
    } break;
    case 6: {  // P/p  main.py:1063-1095: black advances to y-1, red to y+1; sideways past the river
        if (side) {
            if (inb(y - 1, x) && notown(b[(y - 1) * 9 + x])) emit(y - 1, x);
            if (y < 5) {
                if (inb(y, x + 1) && notown(b[y * 9 + x + 1])) emit(y, x + 1);
                if (inb(y, x - 1) && notown(b[y * 9 + x - 1])) emit(y, x - 1);
            }
        } else {
            if (inb(y + 1, x) && notown(b[(y + 1) * 9 + x])) emit(y + 1, x);
            if (y > 4) {
                if (inb(y, x + 1) && notown(b[y * 9 + x + 1])) emit(y, x + 1);
                if (inb(y, x - 1) && notown(b[y * 9 + x - 1])) emit(y, x - 1);
            }
        }
    } break;
    default: break;
    }
    return n;
}

// Ordered pseudo-legal move list of one position (GameBoard.get_legal_moves, main.py:743-1109).
//   b     LDS board [96]; stage LDS [64*18] u16; out LDS [128] u16.
// Returns the move count (wave-uniform), or -1 on overflow / unlabeled move.
// Scan order = ascending sq (y outer, x inner, main.py:754-755): lanes take squares 0..63 then
// 64..89; a wave prefix sum of the per-piece counts places every piece's run.
__device__ __forceinline__ int czd_wave_movegen(const uint8_t *b, int side, const int16_t *lut,
                                                uint16_t *stage, uint16_t *out, int lane) {
    bool err = false;
    int base = 0;
    uint16_t *st = stage + lane * CZD_STAGE_STRIDE;
#pragma unroll 1
    for (int r = 0; r < 2; ++r) {
        const int sq = lane + 64 * r;
        int n = 0;
        if (sq < CZD_NSQ) n = czd_gen_piece(b, sq, side, lut, st, err);
        int total;
        const int off = base + czd_wave_excl_scan(n, lane, &total);
        if (off + n > CZD_MAXMOVES) { err = true; n = 0; }
        for (int k = 0; k < n; ++k) out[off + k] = st[k];  // own staging row: no cross-lane hazard
        base += total;
        if (base > CZD_MAXMOVES) base = CZD_MAXMOVES;
    }
    // flying general, main.py:1097-1107: kings on one file with nothing between -> mover's king captures
    const int c0 = b[lane], c1 = (lane + 64 < CZD_NSQ) ? b[lane + 64] : 0;
    const unsigned long long K0 = __ballot(c0 == 1), K1 = __ballot(c1 == 1);
    const unsigned long long k0 = __ballot(c0 == 8), k1 = __ballot(c1 == 8);
    const int Ksq = K0 ? __ffsll((long long)K0) - 1 : (K1 ? 64 + __ffsll((long long)K1) - 1 : -1);
    const int ksq = k0 ? __ffsll((long long)k0) - 1 : (k1 ? 64 + __ffsll((long long)k1) - 1 : -1);
    if (Ksq >= 0 && ksq >= 0 && (Ksq % 9) == (ksq % 9)) {
        bool face = true;
        for (int s = Ksq + 9; s < ksq; s += 9) face = face && (b[s] == 0);
        if (face) {
            const int src = side ? ksq : Ksq, dst = side ? Ksq : ksq;
            const int l = lut[src * CZD_NSQ + dst];
            if (l < 0 || base >= CZD_MAXMOVES) err = true;
            else { if (lane == 0) out[base] = (uint16_t)l; base += 1; }
        }
    }
    const bool any_err = __ballot(err) != 0ull;
    __syncthreads();
    return any_err ? -1 : base;
}

// MCTS_tree.generate_inputs (main.py:531-533): try_flip (:560-574) for black — reverse the rank
// order and swap case — then state_to_positions (:547-557) with its 9-stride read (quirk Q1).
// Writes [9][10][C] elements of T (float or bf16 bits), C >= 14, coalesced across the wave.
template <typename T>
__device__ __forceinline__ void czd_wave_encode_planes(const uint8_t *b, int side, int quirk_q1, T *out,
                                                       int C, T one, int lane) {
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
