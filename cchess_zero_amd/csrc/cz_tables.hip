// cz_tables.hip — host-side static tables of the move vocabulary and the Zobrist keys.
//
// Move vocabulary: chengstone/cchess-zero main.py:30-65 (create_uci_labels): for every source
// square (file-major: 'a'..'i' outer, rank 0..9 inner) all same-rank destinations, all same-file
// destinations, then the on-board knight jumps in the fixed offset order; then 16 advisor and
// 32 bishop literals.  label2i (main.py:217) becomes a 90x90 LUT; unflipped_index (main.py:214)
// is the rank mirror (digit d -> 9-d, main.py:23-27) looked up through that LUT.
#include "cz_internal.h"

#include <mutex>
#include <string.h>

namespace {

struct Builder {
    CzHostTables t;
    int n = 0;
    void push(int sx, int sy, int dx, int dy) {
        char *s = t.labels + n * 5;
        s[0] = char('a' + sx); s[1] = char('0' + sy); s[2] = char('a' + dx); s[3] = char('0' + dy); s[4] = 0;
        const int src = sy * 9 + sx, dst = dy * 9 + dx;
        t.lut[src * CZ_NSQ + dst] = (int16_t)n;
        t.srcdst[n] = (uint16_t)(src | (dst << 8));
        ++n;
    }
    void push_str(const char *s) { push(s[0] - 'a', s[1] - '0', s[2] - 'a', s[3] - '0'); }
};

uint64_t mix64(uint64_t &state) {  // splitmix64
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

CzHostTables *build() {
    static Builder b;
    for (int i = 0; i < CZ_NSQ * CZ_NSQ; ++i) b.t.lut[i] = -1;
    const int jump[8][2] = {{-2, -1}, {-1, -2}, {-2, 1}, {1, -2}, {2, -1}, {-1, 2}, {2, 1}, {1, 2}};
    for (int fx = 0; fx < 9; ++fx)
        for (int ry = 0; ry < 10; ++ry) {
            for (int dx = 0; dx < 9; ++dx) if (dx != fx) b.push(fx, ry, dx, ry);
            for (int dy = 0; dy < 10; ++dy) if (dy != ry) b.push(fx, ry, fx, dy);
            for (auto &j : jump) {
                const int dx = fx + j[0], dy = ry + j[1];
                if (dx >= 0 && dx < 9 && dy >= 0 && dy < 10) b.push(fx, ry, dx, dy);
            }
        }
    const char *advisor = "d7e8 e8d7 e8f9 f9e8 d0e1 e1d0 e1f2 f2e1 d2e1 e1d2 e1f0 f0e1 d9e8 e8d9 e8f7 f7e8";
    const char *bishop = "a2c4 c4a2 c0e2 e2c0 e2g4 g4e2 g0i2 i2g0 a7c9 c9a7 c5e7 e7c5 e7g9 g9e7 g5i7 i7g5 "
                         "a2c0 c0a2 c4e2 e2c4 e2g0 g0e2 g4i2 i2g4 a7c5 c5a7 c9e7 e7c9 e7g5 g5e7 g9i7 i7g9";
    for (const char *p = advisor; *p; p += (p[4] ? 5 : 4)) b.push_str(p);
    for (const char *p = bishop; *p; p += (p[4] ? 5 : 4)) b.push_str(p);
    if (b.n != CZ_NLABELS) { fprintf(stderr, "cchess_hip: vocabulary has %d labels, expected 2086\n", b.n); abort(); }
    for (int i = 0; i < CZ_NLABELS; ++i) {
        const int s = b.t.srcdst[i] & 0xFF, d = b.t.srcdst[i] >> 8;
        const int ms = (9 - s / 9) * 9 + s % 9, md = (9 - d / 9) * 9 + d % 9;
        b.t.unflip[i] = b.t.lut[ms * CZ_NSQ + md];
    }
    // Zobrist keys: fixed seed, piece-major then square, side key last (format defined by this project).
    uint64_t st = 0xC0FFEE1234567ull;
    for (int q = 0; q < CZ_NSQ; ++q) b.t.zob[q] = 0;
    for (int c = 1; c <= 14; ++c)
        for (int q = 0; q < CZ_NSQ; ++q) b.t.zob[c * CZ_NSQ + q] = mix64(st);
    b.t.zob[15 * CZ_NSQ] = mix64(st);
    return &b.t;
}

}  // namespace

const CzHostTables &cz_host_tables() {
    static CzHostTables *t = build();
    return *t;
}
