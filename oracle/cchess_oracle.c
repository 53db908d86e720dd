/*
 * cchess_oracle.c — scalar CPU restatement of the cchess-zero hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see cchess_oracle.h).  The product path is the HIP
 * library under cchess_zero_amd/csrc; nothing there includes or links this file.
 *
 * Arithmetic note: the reference runs under NumPy scalar
 * promotion rules.  This file restates the NumPy >= 2 (NEP 50) behaviour, which is
 * what the golden vectors were generated with (NumPy 2.2.6):
 *   - priors: float32 logits, tot_p = float32(1e-8) then sequential float32 adds in
 *     move order, P = float32(p / tot_p)                       (main.py:176-187)
 *   - W, Q: float32 (first np.float32 added makes W np.float32; python-float phases
 *     only ever hold small integers, exact in float32)           (main.py:189-194)
 *   - U = float64(float32(5*P)) * sqrt(float64(parent.N)) / (1 + N), Q+U in float64,
 *     first maximum wins (Python max()).                         (main.py:108-116,158)
 */
#include "cchess_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* piece codes: 1 + index in 'KARBNPCkarbnpc' (main.py:208) */
enum { T_K = 1, T_A = 2, T_R = 3, T_B = 4, T_N = 5, T_P = 6, T_C = 7 };
static const char PIECE_CHARS[] = ".KARBNPCkarbnpc";

/* ------------------------------------------------------------------ tables */
static char g_labels[CZO_NLABELS * 5];
static int16_t g_lut[CZO_NSQ * CZO_NSQ];
static int16_t g_unflip[CZO_NLABELS];
static uint16_t g_srcdst[CZO_NLABELS];
static int g_tables_ready = 0;

static int sq_of(char letter, char digit) { return (digit - '0') * 9 + (letter - 'a'); }

static void add_label(int *n, int l1, int n1, int l2, int n2) {
    char *s = g_labels + (*n) * 5;
    s[0] = (char)('a' + l1);
    s[1] = (char)('0' + n1);
    s[2] = (char)('a' + l2);
    s[3] = (char)('0' + n2);
    s[4] = 0;
    (*n)++;
}

static void build_tables(void) {
    if (g_tables_ready) return;
    int n = 0;
    /* create_uci_labels, main.py:30-65 */
    static const int kn[8][2] = {{-2, -1}, {-1, -2}, {-2, 1}, {1, -2}, {2, -1}, {-1, 2}, {2, 1}, {1, 2}};
    for (int l1 = 0; l1 < 9; l1++) {
        for (int n1 = 0; n1 < 10; n1++) {
            for (int t = 0; t < 9; t++) /* (t, n1): same rank */
                if (t != l1) add_label(&n, l1, n1, t, n1);
            for (int t = 0; t < 10; t++) /* (l1, t): same file */
                if (t != n1) add_label(&n, l1, n1, l1, t);
            for (int k = 0; k < 8; k++) { /* knight */
                int l2 = l1 + kn[k][0], n2 = n1 + kn[k][1];
                if (l2 >= 0 && l2 < 9 && n2 >= 0 && n2 < 10) add_label(&n, l1, n1, l2, n2);
            }
        }
    }
    static const char *adv[16] = {"d7e8", "e8d7", "e8f9", "f9e8", "d0e1", "e1d0", "e1f2", "f2e1",
                                  "d2e1", "e1d2", "e1f0", "f0e1", "d9e8", "e8d9", "e8f7", "f7e8"};
    static const char *bis[32] = {"a2c4", "c4a2", "c0e2", "e2c0", "e2g4", "g4e2", "g0i2", "i2g0",
                                  "a7c9", "c9a7", "c5e7", "e7c5", "e7g9", "g9e7", "g5i7", "i7g5",
                                  "a2c0", "c0a2", "c4e2", "e2c4", "e2g0", "g0e2", "g4i2", "i2g4",
                                  "a7c5", "c5a7", "c9e7", "e7c9", "e7g5", "g5e7", "g9i7", "i7g9"};
    for (int i = 0; i < 16; i++) { memcpy(g_labels + n * 5, adv[i], 5); n++; }
    for (int i = 0; i < 32; i++) { memcpy(g_labels + n * 5, bis[i], 5); n++; }
    if (n != CZO_NLABELS) { fprintf(stderr, "czo: label count %d != 2086\n", n); abort(); }

    for (int i = 0; i < CZO_NSQ * CZO_NSQ; i++) g_lut[i] = -1;
    for (int i = 0; i < n; i++) {
        const char *s = g_labels + i * 5;
        int src = sq_of(s[0], s[1]), dst = sq_of(s[2], s[3]);
        g_lut[src * CZO_NSQ + dst] = (int16_t)i; /* label2i, main.py:217 */
        g_srcdst[i] = (uint16_t)(src | (dst << 8));
    }
    /* flipped_uci_labels (main.py:23-27): digit d -> 9-d; unflipped_index (:214) */
    for (int i = 0; i < n; i++) {
        int s = g_srcdst[i] & 0xff, d = g_srcdst[i] >> 8;
        int fs = (9 - s / 9) * 9 + s % 9, fd = (9 - d / 9) * 9 + d % 9;
        g_unflip[i] = g_lut[fs * CZO_NSQ + fd];
    }
    g_tables_ready = 1;
}

const char *czo_labels(void) { build_tables(); return g_labels; }
const int16_t *czo_lut(void) { build_tables(); return g_lut; }
const int16_t *czo_unflip(void) { build_tables(); return g_unflip; }
const uint16_t *czo_label_srcdst(void) { build_tables(); return g_srcdst; }

/* ------------------------------------------------------------------ state strings */
int czo_fen_to_board(const char *fen, uint8_t *b) {
    /* board_to_pos_name, main.py:705-714: digits expand to runs of empties, '/' splits rows */
    memset(b, 0, CZO_NSQ);
    int y = 0, x = 0;
    for (const char *p = fen; *p; p++) {
        char c = *p;
        if (c == '/') {
            if (x != 9) return -1;
            y++; x = 0;
            continue;
        }
        if (y > 9) return -1;
        if (c >= '1' && c <= '9') { x += c - '0'; if (x > 9) return -1; continue; }
        const char *q = strchr(PIECE_CHARS + 1, c);
        if (!q || x >= 9) return -1;
        b[y * 9 + x] = (uint8_t)(q - PIECE_CHARS);
        x++;
    }
    return (y == 9 && x == 9) ? 0 : -1;
}

int czo_board_to_fen(const uint8_t *b, char *out) {
    /* re-compression of sim_do_action, main.py:691-699 */
    int n = 0;
    for (int y = 0; y < 10; y++) {
        int run = 0;
        for (int x = 0; x < 9; x++) {
            uint8_t c = b[y * 9 + x];
            if (!c) { run++; continue; }
            if (run) { out[n++] = (char)('0' + run); run = 0; }
            out[n++] = PIECE_CHARS[c];
        }
        if (run) out[n++] = (char)('0' + run);
        if (y != 9) out[n++] = '/';
    }
    out[n] = 0;
    return n;
}

/* ------------------------------------------------------------------ move generation */
static inline int inb(int y, int x) { return y >= 0 && x >= 0 && y < 10 && x < 9; } /* check_bounds :717 */
static inline int is_black(uint8_t c) { return c > 7; }
static inline int is_red(uint8_t c) { return c >= 1 && c <= 7; }
/* validate_move (main.py:727-740) as used by the mover: empty or an enemy piece */
static inline int not_own(uint8_t c, int side) { return c == 0 || (side ? is_red(c) : is_black(c)); }
static inline int is_enemy(uint8_t c, int side) { return c != 0 && (side ? is_red(c) : is_black(c)); }

typedef struct { uint16_t *out; int n; int err; } mlist;
static void emit(mlist *m, int sy, int sx, int ty, int tx) {
    int16_t l = g_lut[(sy * 9 + sx) * CZO_NSQ + ty * 9 + tx];
    if (l < 0 || m->n >= CZO_MAXMOVES) { m->err = 1; return; }
    m->out[m->n++] = (uint16_t)l;
}

int czo_legal_moves(const uint8_t *b, int side, uint16_t *out) {
    build_tables();
    mlist m = {out, 0, 0};
    int kx = -1, ky = -1, Kx = -1, Ky = -1;
    /* scan order y outer, x inner: main.py:754-755 */
    for (int y = 0; y < 10; y++) {
        for (int x = 0; x < 9; x++) {
            uint8_t c = b[y * 9 + x];
            if (!c) continue;
            int black = is_black(c);
            int t = black ? c - 7 : c;
            int mine = (black == side);
            switch (t) {
            case T_R: /* main.py:757-833 : -x, +x, -y, +y; stop at first piece, capture if enemy */
                if (!mine) break;
                for (int tx = x - 1; tx >= 0; tx--) { uint8_t d = b[y * 9 + tx]; if (d) { if (is_enemy(d, side)) emit(&m, y, x, y, tx); break; } emit(&m, y, x, y, tx); }
                for (int tx = x + 1; tx < 9; tx++) { uint8_t d = b[y * 9 + tx]; if (d) { if (is_enemy(d, side)) emit(&m, y, x, y, tx); break; } emit(&m, y, x, y, tx); }
                for (int ty = y - 1; ty >= 0; ty--) { uint8_t d = b[ty * 9 + x]; if (d) { if (is_enemy(d, side)) emit(&m, y, x, ty, x); break; } emit(&m, y, x, ty, x); }
                for (int ty = y + 1; ty < 10; ty++) { uint8_t d = b[ty * 9 + x]; if (d) { if (is_enemy(d, side)) emit(&m, y, x, ty, x); break; } emit(&m, y, x, ty, x); }
                break;
            case T_N: /* main.py:835-856 */
                if (!mine) break;
                for (int i = -1; i < 3; i += 2)
                    for (int j = -1; j < 3; j += 2) {
                        int ty = y + 2 * i, tx = x + j;
                        if (inb(ty, tx) && not_own(b[ty * 9 + tx], side) && b[(ty - i) * 9 + x] == 0) emit(&m, y, x, ty, tx);
                        ty = y + i; tx = x + 2 * j;
                        if (inb(ty, tx) && not_own(b[ty * 9 + tx], side) && b[y * 9 + (tx - j)] == 0) emit(&m, y, x, ty, tx);
                    }
                break;
            case T_B: /* main.py:857-888 : own half only (black toY>=5, red toY<=4), eye must be empty */
                if (!mine) break;
                for (int i = -2; i < 3; i += 4) {
                    int h = i / 2; /* python i//2 for i in {-2,2} */
                    int ty = y + i, tx = x + i;
                    if (inb(ty, tx) && not_own(b[ty * 9 + tx], side) && (side ? ty >= 5 : ty <= 4) && b[(y + h) * 9 + (x + h)] == 0) emit(&m, y, x, ty, tx);
                    ty = y + i; tx = x - i;
                    if (inb(ty, tx) && not_own(b[ty * 9 + tx], side) && (side ? ty >= 5 : ty <= 4) && b[(y + h) * 9 + (x - h)] == 0) emit(&m, y, x, ty, tx);
                }
                break;
            case T_A: /* main.py:889-918 : destination inside own palace */
                if (!mine) break;
                for (int i = -1; i < 3; i += 2) {
                    int ty = y + i, tx = x + i;
                    if (inb(ty, tx) && not_own(b[ty * 9 + tx], side) && (side ? ty >= 7 : ty <= 2) && tx >= 3 && tx <= 5) emit(&m, y, x, ty, tx);
                    ty = y + i; tx = x - i;
                    if (inb(ty, tx) && not_own(b[ty * 9 + tx], side) && (side ? ty >= 7 : ty <= 2) && tx >= 3 && tx <= 5) emit(&m, y, x, ty, tx);
                }
                break;
            case T_K: /* main.py:919-946 : king squares recorded for BOTH colours */
                if (black) { kx = x; ky = y; } else { Kx = x; Ky = y; }
                if (!mine) break;
                for (int i = 0; i < 2; i++)
                    for (int sign = -1; sign < 2; sign += 2) {
                        int j = 1 - i;
                        int ty = y + i * sign, tx = x + j * sign;
                        if (inb(ty, tx) && not_own(b[ty * 9 + tx], side) && (side ? ty >= 7 : ty <= 2) && tx >= 3 && tx <= 5) emit(&m, y, x, ty, tx);
                    }
                break;
            case T_C: /* main.py:947-1062 : slide over empties; after one screen capture first enemy */
                if (!mine) break;
                {
                    int hits = 0;
                    for (int tx = x - 1; tx >= 0; tx--) { uint8_t d = b[y * 9 + tx]; if (!hits) { if (d) hits = 1; else emit(&m, y, x, y, tx); } else if (d) { if (is_enemy(d, side)) emit(&m, y, x, y, tx); break; } }
                    hits = 0;
                    for (int tx = x + 1; tx < 9; tx++) { uint8_t d = b[y * 9 + tx]; if (!hits) { if (d) hits = 1; else emit(&m, y, x, y, tx); } else if (d) { if (is_enemy(d, side)) emit(&m, y, x, y, tx); break; } }
                    hits = 0;
                    for (int ty = y - 1; ty >= 0; ty--) { uint8_t d = b[ty * 9 + x]; if (!hits) { if (d) hits = 1; else emit(&m, y, x, ty, x); } else if (d) { if (is_enemy(d, side)) emit(&m, y, x, ty, x); break; } }
                    hits = 0;
                    for (int ty = y + 1; ty < 10; ty++) { uint8_t d = b[ty * 9 + x]; if (!hits) { if (d) hits = 1; else emit(&m, y, x, ty, x); } else if (d) { if (is_enemy(d, side)) emit(&m, y, x, ty, x); break; } }
                }
                break;
            case T_P: /* main.py:1063-1095 : black moves toward y-1, red toward y+1; sideways after the river */
                if (!mine) break;
                if (side) {
                    int ty = y - 1, tx = x;
                    if (inb(ty, tx) && not_own(b[ty * 9 + tx], side)) emit(&m, y, x, ty, tx);
                    if (y < 5) {
                        ty = y; tx = x + 1;
                        if (inb(ty, tx) && not_own(b[ty * 9 + tx], side)) emit(&m, y, x, ty, tx);
                        tx = x - 1;
                        if (inb(ty, tx) && not_own(b[ty * 9 + tx], side)) emit(&m, y, x, ty, tx);
                    }
                } else {
                    int ty = y + 1, tx = x;
                    if (inb(ty, tx) && not_own(b[ty * 9 + tx], side)) emit(&m, y, x, ty, tx);
                    if (y > 4) {
                        ty = y; tx = x + 1;
                        if (inb(ty, tx) && not_own(b[ty * 9 + tx], side)) emit(&m, y, x, ty, tx);
                        tx = x - 1;
                        if (inb(ty, tx) && not_own(b[ty * 9 + tx], side)) emit(&m, y, x, ty, tx);
                    }
                }
                break;
            default: break;
            }
        }
    }
    /* flying general, main.py:1097-1107 */
    if (Kx >= 0 && kx >= 0 && Kx == kx) {
        int face = 1;
        for (int i = Ky + 1; i < ky; i++)
            if (b[i * 9 + Kx]) face = 0;
        if (face) {
            if (side) emit(&m, ky, kx, Ky, Kx);
            else emit(&m, Ky, Kx, ky, kx);
        }
    }
    return m.err ? -1 : m.n;
}

int czo_apply_move(uint8_t *b, uint16_t label, uint8_t *captured) {
    build_tables();
    int src = g_srcdst[label] & 0xff, dst = g_srcdst[label] >> 8;
    uint8_t cap = b[dst]; /* is_kill_move: piece count delta, main.py:219-227 */
    b[dst] = b[src];      /* main.py:671-672 */
    b[src] = 0;
    if (captured) *captured = cap;
    int K = 0, k = 0;
    for (int i = 0; i < CZO_NSQ; i++) { K |= (b[i] == T_K); k |= (b[i] == T_K + 7); }
    return (K ? 0 : 1) | (k ? 0 : 2); /* state.find('K') == -1 / find('k') == -1, main.py:409-413 */
}

void czo_encode_planes(const uint8_t *b, int side, int quirk_q1, float *planes) {
    uint8_t f[CZO_NSQ];
    if (side) { /* try_flip, main.py:560-574: reverse the rank order, swap case */
        for (int y = 0; y < 10; y++)
            for (int x = 0; x < 9; x++) {
                uint8_t c = b[(9 - y) * 9 + x];
                f[y * 9 + x] = c == 0 ? 0 : (c > 7 ? c - 7 : c + 7);
            }
    } else memcpy(f, b, CZO_NSQ);
    memset(planes, 0, sizeof(float) * CZO_PLANE_ELEMS);
    if (quirk_q1) {
        /* state_to_positions, main.py:547-557: v = board_state[rank*9+file], rank<9, file<10 */
        for (int rank = 0; rank < 9; rank++)
            for (int file = 0; file < 10; file++) {
                uint8_t c = f[rank * 9 + file];
                if (c) planes[(rank * 10 + file) * 14 + (c - 1)] = 1.0f;
            }
    } else {
        /* what the shapes suggest was intended: plane[file x][rank y] */
        for (int x = 0; x < 9; x++)
            for (int y = 0; y < 10; y++) {
                uint8_t c = f[y * 9 + x];
                if (c) planes[(x * 10 + y) * 14 + (c - 1)] = 1.0f;
            }
    }
}

/* ------------------------------------------------------------------ Zobrist (project-defined) */
static uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t g_zob[15 * CZO_NSQ];
static uint64_t g_zob_side;
static int g_zob_ready = 0;
static void build_zobrist(void) {
    if (g_zob_ready) return;
    uint64_t s = 0xC0FFEE1234567ull; /* fixed seed: part of the format */
    for (int c = 1; c <= 14; c++)
        for (int q = 0; q < CZO_NSQ; q++) g_zob[c * CZO_NSQ + q] = splitmix64(&s);
    g_zob_side = splitmix64(&s);
    g_zob_ready = 1;
}
uint64_t czo_zobrist_key(int code, int sq) { build_zobrist(); return g_zob[code * CZO_NSQ + sq]; }
uint64_t czo_zobrist_side(void) { build_zobrist(); return g_zob_side; }
uint64_t czo_hash(const uint8_t *b, int side) {
    build_zobrist();
    uint64_t h = side ? g_zob_side : 0;
    for (int q = 0; q < CZO_NSQ; q++)
        if (b[q]) h ^= g_zob[b[q] * CZO_NSQ + q];
    return h;
}

/* ------------------------------------------------------------------ search */
typedef struct {
    float P, W, Q;       /* leaf_node.P/W/Q main.py:95-100 */
    int32_t N;
    int32_t parent;      /* -1 for a root */
    int32_t child_begin; /* -1 = not in `expanded` (main.py:247) */
    uint16_t child_count;
    uint16_t move;       /* label of the edge into this node */
} onode;

typedef struct {
    onode *nodes;
    int n_nodes, cap;
    int root;
    uint8_t board[CZO_NSQ];
    int side, rr;
    /* pending leaf between select and expand_backup */
    int kind; /* 0 idle, 1 eval+expand+backup, 2 terminal/draw backup, 3 root expansion */
    int leaf;
    float pend_value;
    int leaf_side;
    uint16_t moves[CZO_MAXMOVES];
    int nmoves;
    int status, sims, last_depth;
    /* k-wide search: pending slots */
    struct oslot { int kind, leaf, side, nmoves; uint16_t moves[CZO_MAXMOVES]; } *slots;
    int nslots;
} otree;

struct czo_search {
    int max_games, cap, G;
    otree *t;
};

czo_search *czo_search_create(int max_games, int cap) {
    build_tables();
    czo_search *s = (czo_search *)calloc(1, sizeof(*s));
    s->max_games = max_games; s->cap = cap; s->G = 0;
    s->t = (otree *)calloc((size_t)max_games, sizeof(otree));
    for (int i = 0; i < max_games; i++) {
        s->t[i].nodes = (onode *)malloc(sizeof(onode) * (size_t)cap);
        s->t[i].cap = cap;
    }
    return s;
}
void czo_search_destroy(czo_search *s) {
    if (!s) return;
    for (int i = 0; i < s->max_games; i++) { free(s->t[i].nodes); free(s->t[i].slots); }
    free(s->t); free(s);
}

static void fresh_root(otree *t) {
    t->n_nodes = 1; t->root = 0;
    onode *r = &t->nodes[0];
    r->P = 1.0f; /* p_ = 0.75 + 0.25*dirichlet([0.3]) == 1.0 (quirk Q4), main.py:238 */
    r->W = 0; r->Q = 0; r->N = 0; r->parent = -1; r->child_begin = -1; r->child_count = 0; r->move = 0xffff;
    t->kind = 0;
}

int czo_search_reset(czo_search *s, const uint8_t *boards, const uint8_t *side, const int32_t *rr, int G) {
    if (G > s->max_games) return -1;
    s->G = G;
    for (int g = 0; g < G; g++) {
        otree *t = &s->t[g];
        memcpy(t->board, boards + (size_t)g * CZO_NSQ, CZO_NSQ);
        t->side = side[g] ? 1 : 0;
        t->rr = rr ? rr[g] : 0;
        t->status = 0; t->sims = 0; t->last_depth = 0;
        fresh_root(t);
    }
    return 0;
}

/* select_new / get_Q_plus_U_new, main.py:108-116,158-159 */
static int select_child(const otree *t, int node) {
    const onode *p = &t->nodes[node];
    /* every non-root node on the current path carries its virtual loss (N += 3, main.py:403)
     * while its children are being scored, so parent.N is inflated by 3 below the root;
     * Q is NOT recomputed under virtual loss (quirk Q6). */
    int vl = (node != t->root) ? 3 : 0;
    double sq = sqrt((double)(p->N + vl));
    int best = -1; double bestv = 0;
    for (int i = 0; i < p->child_count; i++) {
        const onode *c = &t->nodes[p->child_begin + i];
        float cp = 5.0f * c->P; /* c_puct(int) * np.float32 -> float32 */
        double u = (double)cp * sq / (double)(1 + c->N);
        double v = (double)c->Q + u;
        if (best < 0 || v > bestv) { best = p->child_begin + i; bestv = v; } /* first max wins */
    }
    return best;
}

int czo_search_select(czo_search *s, int mode, float *planes, uint8_t *needs_eval) {
    for (int g = 0; g < s->G; g++) {
        otree *t = &s->t[g];
        float *pl = planes ? planes + (size_t)g * CZO_PLANE_ELEMS : NULL;
        if (pl) memset(pl, 0, sizeof(float) * CZO_PLANE_ELEMS);
        if (needs_eval) needs_eval[g] = 0;
        t->kind = 0;
        if (t->status & ~8) continue; /* a failed tree stays parked (bit 3 is informational) */
        uint8_t b[CZO_NSQ];
        memcpy(b, t->board, CZO_NSQ);
        int side = t->side, rr = t->rr, node = t->root, depth = 0;
        if (t->nodes[node].child_begin < 0) {
            /* MCTS_tree.main root expansion, main.py:475-487 (value discarded, no backup) */
            t->kind = 3; t->leaf = node;
        } else if (mode == 0) {
            continue;
        } else {
            for (;;) {
                onode *p = &t->nodes[node];
                if (p->child_begin < 0) { t->kind = 1; t->leaf = node; break; } /* main.py:357 */
                if (p->child_count == 0) { t->status |= 2; break; }               /* max() of empty, quirk Q7 */
                int c = select_child(t, node);                                    /* main.py:391 */
                uint8_t cap;
                int term = czo_apply_move(b, t->nodes[c].move, &cap);
                side ^= 1;                                                        /* main.py:392 */
                rr = cap ? 0 : rr + 1;                                            /* main.py:393-396 */
                depth++;
                /* virtual loss (main.py:403-404, 426-427): one simulation in flight per tree, so nobody
                 * observes N+3; its float32 rounding effect on W is applied in expand_backup */
                if (term) {
                    /* main.py:409-414; current_player is the side to move at the child */
                    float value = 0;
                    if (term & 1) value = side ? 1.0f : -1.0f;
                    if (term & 2) value = side ? -1.0f : 1.0f;
                    value = value * -1;
                    t->kind = 2; t->leaf = c; t->pend_value = value; break;
                } else if (rr >= 60) {                                            /* main.py:415-416 */
                    t->kind = 2; t->leaf = c; t->pend_value = 0.0f; break;
                }
                node = c;
            }
        }
        t->last_depth = depth;
        if (t->kind == 1 || t->kind == 3) {
            t->leaf_side = side;
            int n = czo_legal_moves(b, side, t->moves);                           /* main.py:374 / 483 */
            if (n < 0) { t->status |= 4; t->kind = 0; continue; }
            t->nmoves = n;
            if (pl) czo_encode_planes(b, side, 1, pl);                            /* main.py:362 / 477 */
            if (needs_eval) needs_eval[g] = 1;
        }
    }
    return 0;
}

int czo_search_expand_backup(czo_search *s, const float *logits, const float *value) {
    for (int g = 0; g < s->G; g++) {
        otree *t = &s->t[g];
        if (t->kind == 0) continue;
        float v;
        if (t->kind == 1 || t->kind == 3) {
            /* leaf_node.expand, main.py:175-187; flip_policy for black, main.py:371-372,1153-1155 */
            const float *lg = logits + (size_t)g * CZO_NLABELS;
            if (t->n_nodes + t->nmoves > t->cap) {
                t->status |= 1;
            } else {
                onode *leaf = &t->nodes[t->leaf];
                int begin = t->n_nodes;
                float tot = (float)1e-8;
                for (int i = 0; i < t->nmoves; i++) {
                    uint16_t l = t->moves[i];
                    float p = lg[t->leaf_side ? g_unflip[l] : l];
                    onode *c = &t->nodes[begin + i];
                    c->P = p; c->W = 0; c->Q = 0; c->N = 0; c->parent = t->leaf;
                    c->child_begin = -1; c->child_count = 0; c->move = l;
                    tot = tot + p;
                }
                for (int i = 0; i < t->nmoves; i++) t->nodes[begin + i].P = t->nodes[begin + i].P / tot;
                leaf->child_begin = begin; leaf->child_count = (uint16_t)t->nmoves;
                t->n_nodes += t->nmoves;
            }
            if (t->kind == 3) { t->kind = 0; continue; }
            v = value[g] * -1; /* return value[0] * -1, main.py:384 */
        } else {
            v = t->pend_value;
        }
        /* back_up_value along the recursion unwind, main.py:189-194, 432-435.
         * The root is never updated (quirk Q2). */
        int n = t->leaf;
        while (n != t->root) {
            onode *x = &t->nodes[n];
            /* virtual loss add/remove (main.py:403-404, 426-427): N nets to zero, but in
             * float32 (W - 3) + 3 != W in general, so the rounding must be reproduced. */
            x->W = x->W + -3.0f;
            x->W = x->W + 3.0f;
            x->N += 1;
            x->W = x->W + v;
            x->Q = x->W / (float)x->N;
            v = v * -1;
            n = x->parent;
        }
        t->sims++;
        t->kind = 0;
    }
    return 0;
}

int czo_search_root_stats(const czo_search *s, uint16_t *label, int32_t *N, float *Q, float *P, float *W, uint16_t *count) {
    for (int g = 0; g < s->G; g++) {
        const otree *t = &s->t[g];
        const onode *r = &t->nodes[t->root];
        int n = r->child_begin < 0 ? 0 : r->child_count;
        if (count) count[g] = (uint16_t)n;
        for (int i = 0; i < CZO_MAXMOVES; i++) {
            size_t o = (size_t)g * CZO_MAXMOVES + i;
            const onode *c = i < n ? &t->nodes[r->child_begin + i] : NULL;
            if (label) label[o] = c ? c->move : 0xffff;
            if (N) N[o] = c ? c->N : 0;
            if (Q) Q[o] = c ? c->Q : 0;
            if (P) P[o] = c ? c->P : 0;
            if (W) W[o] = c ? c->W : 0;
        }
    }
    return 0;
}

int czo_search_advance(czo_search *s, const uint16_t *played) {
    for (int g = 0; g < s->G; g++) {
        otree *t = &s->t[g];
        uint16_t l = played[g];
        if (l >= CZO_NLABELS) continue; /* 0xffff = leave this game alone */
        const onode *r = &t->nodes[t->root];
        int found = -1;
        if (r->child_begin >= 0)
            for (int i = 0; i < r->child_count; i++)
                if (t->nodes[r->child_begin + i].move == l) { found = r->child_begin + i; break; }
        uint8_t cap;
        czo_apply_move(t->board, l, &cap);          /* main.py:1522 */
        t->side ^= 1;                               /* main.py:1524 */
        t->rr = cap ? 0 : t->rr + 1;                /* main.py:1525-1528 */
        if (found >= 0) {
            t->root = found;                        /* update_tree, main.py:272-276 */
            t->nodes[found].parent = -1;
        } else {
            t->status |= 8;                         /* reference: KeyError */
            fresh_root(t);
        }
        t->sims = 0;
    }
    return 0;
}

int czo_search_status(const czo_search *s, int32_t *status, int32_t *nodes_used, int32_t *sims) {
    for (int g = 0; g < s->G; g++) {
        if (status) status[g] = s->t[g].status;
        if (nodes_used) nodes_used[g] = s->t[g].n_nodes;
        if (sims) sims[g] = s->t[g].sims;
    }
    return 0;
}

int czo_search_root_state(const czo_search *s, uint8_t *boards, uint8_t *side, int32_t *rr) {
    for (int g = 0; g < s->G; g++) {
        if (boards) memcpy(boards + (size_t)g * CZO_NSQ, s->t[g].board, CZO_NSQ);
        if (side) side[g] = (uint8_t)s->t[g].side;
        if (rr) rr[g] = s->t[g].rr;
    }
    return 0;
}

int czo_search_last_depth(const czo_search *s, int32_t *depth) {
    for (int g = 0; g < s->G; g++) depth[g] = s->t[g].last_depth;
    return 0;
}

/* Canonical pre-order dump of one tree for whole-tree parity checks.
 * record = 7 x int32: depth, label, N, bits(W), bits(Q), bits(P), child_count (-1 = unexpanded).
 * Returns the number of records (may exceed max_records; only max_records are written). */
static int f2i(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static void dump_rec(const otree *t, int node, int depth, int32_t *out, int max_records, int *n) {
    const onode *p = &t->nodes[node];
    if (p->child_begin < 0) return;
    for (int i = 0; i < p->child_count; i++) {
        const onode *c = &t->nodes[p->child_begin + i];
        if (*n < max_records) {
            int32_t *r = out + (size_t)(*n) * 7;
            r[0] = depth; r[1] = c->move; r[2] = c->N; r[3] = f2i(c->W); r[4] = f2i(c->Q); r[5] = f2i(c->P);
            r[6] = c->child_begin < 0 ? -1 : c->child_count;
        }
        (*n)++;
        dump_rec(t, p->child_begin + i, depth + 1, out, max_records, n);
    }
}
int czo_search_tree_dump(const czo_search *s, int g, int32_t *out, int max_records) {
    int n = 0;
    dump_rec(&s->t[g], s->t[g].root, 0, out, max_records, &n);
    return n;
}

/* ---- k simulations in flight per tree: restatement of the schedule of k_select_k / k_expand_backup_k
 * (cchess_zero_amd/csrc/cz_search.hip), which batches the reference's search_threads coroutines
 * (main.py:250,337-348) with its virtual loss (:231,403-404,426-427).  The reference's own interleaving
 * depends on wall-clock sleeps (:355,452), so this — not a golden tree — is the parity target for k > 1. */
static int select_child_vl(const otree *t, int node) {
    const onode *p = &t->nodes[node];
    double sq = sqrt((double)p->N); /* physical virtual losses included */
    int best = -1; double bestv = 0;
    for (int i = 0; i < p->child_count; i++) {
        const onode *c = &t->nodes[p->child_begin + i];
        float cp = 5.0f * c->P;
        double u = (double)cp * sq / (double)(1 + c->N);
        double v = (double)c->Q + u;
        if (best < 0 || v > bestv) { best = p->child_begin + i; bestv = v; }
    }
    return best;
}

static int g_sim_target = 0;
/* per-tree budget of completed + in-flight simulations for the k > 1 schedule (mirrors cz_search_set_sim_target) */
int czo_search_set_sim_target(czo_search *s, int target) { (void)s; g_sim_target = target; return 0; }

int czo_search_select_k(czo_search *s, int mode, int K, float *planes, uint8_t *needs_eval) {
    for (int g = 0; g < s->G; g++) {
        otree *t = &s->t[g];
        if (t->nslots < K) { free(t->slots); t->slots = calloc((size_t)K, sizeof(*t->slots)); t->nslots = K; }
        int stop = (t->status & ~8) != 0;
        const int budget = g_sim_target > 0 ? g_sim_target - t->sims : 0x7FFFFFFF;
        int issued = 0;
        for (int j = 0; j < K; j++) {
            if (mode != 0 && issued >= budget) stop = 1;
            size_t slot = (size_t)g * K + j;
            float *pl = planes ? planes + slot * CZO_PLANE_ELEMS : NULL;
            if (pl) memset(pl, 0, sizeof(float) * CZO_PLANE_ELEMS);
            if (needs_eval) needs_eval[slot] = 0;
            struct oslot *sl = &t->slots[j];
            sl->kind = 0;
            if (stop) continue;
            uint8_t b[CZO_NSQ];
            memcpy(b, t->board, CZO_NSQ);
            int side = t->side, rr = t->rr, node = t->root, depth = 0;
            if (t->nodes[node].child_begin < 0) { sl->kind = 3; sl->leaf = node; stop = 1; }
            else if (mode == 0) { stop = 1; }
            else {
                for (;;) {
                    onode *p = &t->nodes[node];
                    int abandon = 0;
                    if (p->child_begin == -1) { sl->kind = 1; sl->leaf = node; p->child_begin = -2; issued++; break; }
                    if (p->child_begin == -2) abandon = 1;
                    else if (p->child_count == 0) { t->status |= 2; abandon = 1; }
                    if (abandon) {
                        for (int n = node; n != t->root; n = t->nodes[n].parent) { t->nodes[n].N -= 3; t->nodes[n].W = t->nodes[n].W + 3.0f; }
                        stop = 1; break;
                    }
                    int c = select_child_vl(t, node);
                    uint8_t cap;
                    int term = czo_apply_move(b, t->nodes[c].move, &cap);
                    t->nodes[c].N += 3; t->nodes[c].W = t->nodes[c].W + -3.0f;
                    side ^= 1; rr = cap ? 0 : rr + 1; depth++;
                    if (term || rr >= 60) {
                        float value = 0;
                        if (term) { if (term & 1) value = side ? 1.0f : -1.0f; if (term & 2) value = side ? -1.0f : 1.0f; value = value * -1; }
                        float x = value;
                        for (int n = c; n != t->root; n = t->nodes[n].parent) {
                            onode *q = &t->nodes[n];
                            int cnt = q->N - 3 + 1;
                            float w = q->W + 3.0f; w = w + x;
                            q->N = cnt; q->W = w; q->Q = w / (float)cnt; x = x * -1;
                        }
                        t->sims++;
                        issued++;
                        break;
                    }
                    node = c;
                }
            }
            if (sl->kind == 1 || sl->kind == 3) {
                sl->side = side;
                int n = czo_legal_moves(b, side, sl->moves);
                if (n < 0) {
                    t->status |= 4;
                    if (sl->kind == 1) {
                        issued--;
                        t->nodes[sl->leaf].child_begin = -1;
                        for (int m = sl->leaf; m != t->root; m = t->nodes[m].parent) { t->nodes[m].N -= 3; t->nodes[m].W = t->nodes[m].W + 3.0f; }
                    }
                    sl->kind = 0; stop = 1; continue;
                }
                sl->nmoves = n;
                t->last_depth = depth;
                if (pl) czo_encode_planes(b, side, 1, pl);
                if (needs_eval) needs_eval[slot] = 1;
            }
        }
    }
    return 0;
}

int czo_search_expand_backup_k(czo_search *s, int K, const float *logits, const float *value) {
    for (int g = 0; g < s->G; g++) {
        otree *t = &s->t[g];
        for (int j = 0; j < K && j < t->nslots; j++) {
            struct oslot *sl = &t->slots[j];
            if (!sl->kind) continue;
            size_t slot = (size_t)g * K + j;
            const float *lg = logits + slot * CZO_NLABELS;
            if (t->n_nodes + sl->nmoves > t->cap) {
                t->status |= 1;
                t->nodes[sl->leaf].child_begin = -1;
            } else {
                int begin = t->n_nodes;
                float tot = (float)1e-8;
                for (int i = 0; i < sl->nmoves; i++) {
                    uint16_t l = sl->moves[i];
                    float p = lg[sl->side ? g_unflip[l] : l];
                    onode *c = &t->nodes[begin + i];
                    c->P = p; c->W = 0; c->Q = 0; c->N = 0; c->parent = sl->leaf; c->child_begin = -1; c->child_count = 0; c->move = l;
                    tot = tot + p;
                }
                for (int i = 0; i < sl->nmoves; i++) t->nodes[begin + i].P = t->nodes[begin + i].P / tot;
                t->nodes[sl->leaf].child_begin = begin; t->nodes[sl->leaf].child_count = (uint16_t)sl->nmoves;
                t->n_nodes += sl->nmoves;
            }
            if (sl->kind == 1) {
                float x = value[slot] * -1;
                for (int n = sl->leaf; n != t->root; n = t->nodes[n].parent) {
                    onode *q = &t->nodes[n];
                    int cnt = q->N - 3 + 1;
                    float w = q->W + 3.0f; w = w + x;
                    q->N = cnt; q->W = w; q->Q = w / (float)cnt; x = x * -1;
                }
                t->sims++;
            }
            sl->kind = 0;
        }
    }
    return 0;
}
