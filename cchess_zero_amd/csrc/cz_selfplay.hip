// cz_selfplay.hip — the per-ply bookkeeping of self-play for G concurrent games, device-resident.
//
// Restates, for every game slot of a cz_ctx at once (chengstone/cchess-zero main.py):
//   cchess_main.get_action   :1332-1358   visits -> softmax(log N / T), 0.75 pi + 0.25 Dirichlet(0.3) sampling
//   cchess_main.selfplay     :1493-1554   per-ply (state, pi, mover) lists, game end tests, z assignment, reload
// One wave64 workgroup per game.  Nothing here needs the host: the random numbers come in as device arrays, the
// finished games' records leave through a device ring, finished slots are re-seeded in place, so a loop of
// (search, choose, advance, adjudicate, flush) launches keeps every slot busy for as long as it runs.
//
// Record (CZ_REC_BYTES = 608, include/cchess_hip.h): the root position BEFORE the move, the mover, the root's
// children in generation order with their visit counts, the game result from the mover's point of view.  pi is not
// stored: the reference's pi = softmax(1/T * log(visits)) (main.py:1341) is a pure function of the visit counts, so
// the host recomputes it in float64 with the reference's own expression and gets it bit for bit.
#include "cz_internal.h"

#include <math.h>

namespace {

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}
__device__ __forceinline__ double wave_max(double x) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x = fmax(x, __shfl_xor(x, d, 64));
    return x;
}
__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
    double x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

__device__ __forceinline__ void seed_slot(const CzTrees &t, const CzSelfplay &sp, int g, int lane) {
    // MCTS_tree.reload + GameBoard.reload (main.py:255-259, 582-588): a fresh, unexpanded root on the slot's start position
    for (int i = lane; i < CZD_BOARD_LDS; i += 64)
        t.root_board[(size_t)g * CZD_BOARD_LDS + i] = sp.start_board[(size_t)g * CZD_BOARD_LDS + i];
    if (lane == 0) {
        t.root_side[g] = sp.start_side[g];
        t.root_rr[g] = sp.start_rr[g];
        t.root_node[g] = 0; t.n_nodes[g] = 1; t.status[g] = 0; t.sims[g] = 0; t.last_depth[g] = 0; t.root_ply[g] = 0;
        init_root(view_of(t, g), 0);
        sp.ply[g] = 0; sp.stalled[g] = 0;
    }
    ec_clear_tree(t, g, lane, 64);
}

__global__ __launch_bounds__(64) void k_sp_seed(CzTrees t, CzSelfplay sp, int G, const uint8_t *__restrict__ boards,
                                                const uint8_t *__restrict__ side, const int32_t *__restrict__ rr) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G) return;
    // the position new games of this slot start from: given, or the tree's current root position
    for (int i = lane; i < CZD_BOARD_LDS; i += 64)
        sp.start_board[(size_t)g * CZD_BOARD_LDS + i] =
            boards ? (i < CZ_NSQ ? boards[(size_t)g * CZ_NSQ + i] : (uint8_t)0) : t.root_board[(size_t)g * CZD_BOARD_LDS + i];
    if (lane == 0) {
        sp.start_side[g] = boards ? (side[g] ? 1 : 0) : t.root_side[g];
        sp.start_rr[g] = boards ? (rr ? rr[g] : 0) : t.root_rr[g];
        sp.ply[g] = 0; sp.stalled[g] = 0; sp.active[g] = 1;
        if (g == 0)
            for (int k = 0; k < CZ_SP_NSTATS; ++k) sp.stats[k] = 0;
    }
}

// get_action (main.py:1337-1351) + the record append of selfplay (:1504-1518) for every active game.
__global__ __launch_bounds__(64) void k_sp_choose(CzTrees t, CzSelfplay sp, int G, const float *__restrict__ gamma,
                                                  const float *__restrict__ u, const uint16_t *__restrict__ forced,
                                                  double inv_temp, float eps, int min_sims, uint16_t *__restrict__ played) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G) return;
    if (!sp.active[g]) { if (lane == 0) played[g] = 0xFFFF; return; }
    // asynchronous plies (min_sims > 0): only the games whose search has had its playouts move now — or cannot go on
    // (node pool full: the move is chosen from the visits it has; rules overflow: dropped below)
    if (min_sims > 0 && t.sims[g] < min_sims && (t.status[g] & (CZ_ST_POOL_EXHAUSTED | CZ_ST_NO_MOVES | CZ_ST_MOVE_OVERFLOW)) == 0) {
        if (lane == 0) played[g] = 0xFFFF;
        return;
    }
    if (lane == 0) atomicAdd((unsigned long long *)&sp.stats[CZ_SP_SIMS], (unsigned long long)t.sims[g]);
    const TreeView v = view_of(t, g);
    const int root = t.root_node[g];
    const int cb = v.child_begin[root];
    const int n = cb < 0 ? 0 : (int)v.child_count[root];
    if (n == 0 || (t.status[g] & (CZ_ST_NO_MOVES | CZ_ST_MOVE_OVERFLOW)) != 0) {
        // no child to play (node pool exhausted at the root, or a rules overflow): the game cannot continue; the
        // adjudication drops it and re-seeds the slot (the reference has no node limit, so it has no such case)
        if (lane == 0) { played[g] = 0xFFFF; sp.stalled[g] = 1; }
        return;
    }
    int N[2] = {0, 0};
    uint16_t lab[2] = {0xFFFF, 0xFFFF};
    double x[2], e[2], p[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = lane + 64 * r;
        if (i < n) { N[r] = v.N[cb + i]; lab[r] = v.move[cb + i]; }
        // probs = softmax(1.0 / temperature * np.log(visits)), main.py:1341: log(0) = -inf -> probability 0
        x[r] = (i < n && N[r] > 0) ? inv_temp * log((double)N[r]) : -INFINITY;
    }
    const double m = wave_max(fmax(x[0], x[1]));
#pragma unroll
    for (int r = 0; r < 2; ++r) e[r] = (x[r] == -INFINITY || m == -INFINITY) ? 0.0 : exp(x[r] - m);
    const double se = wave_sum(e[0] + e[1]);
    double gm[2] = {0.0, 0.0};
    if (gamma && eps > 0.f) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = lane + 64 * r;
            if (i < n) gm[r] = (double)gamma[(size_t)g * CZD_MAXMOVES + i];
        }
    }
    const double sg = wave_sum(gm[0] + gm[1]);
    const bool noise = gamma && eps > 0.f && sg > 0.0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = lane + 64 * r;
        // with no visit at all (zero playouts) the reference's softmax is NaN and np.random.choice raises; play uniformly
        const double pi = se > 0.0 ? e[r] / se : (i < n ? 1.0 / (double)n : 0.0);
        // 0.75 * probs + 0.25 * np.random.dirichlet(0.3 * np.ones(len(probs))), main.py:1346
        p[r] = i < n ? (noise ? (1.0 - (double)eps) * pi + (double)eps * (gm[r] / sg) : pi) : 0.0;
    }
    // np.random.choice(actions, p = ...): inverse CDF over the children in generation order
    const double c0 = wave_incl_scan(p[0], lane);
    const double t0 = __shfl(c0, 63, 64);
    const double c1 = t0 + wave_incl_scan(p[1], lane);
    const double total = __shfl(c1, 63, 64);
    const double target = (double)u[g] * total;
    const unsigned long long h0 = __ballot(p[0] > 0.0 && c0 > target), h1 = __ballot(p[1] > 0.0 && c1 > target);
    int pick;
    if (h0) pick = __ffsll((long long)h0) - 1;
    else if (h1) pick = 64 + __ffsll((long long)h1) - 1;
    else {   // rounding at the upper end: the last child with a positive probability
        const unsigned long long q1 = __ballot(p[1] > 0.0), q0 = __ballot(p[0] > 0.0);
        pick = q1 ? 127 - __clzll((long long)q1) : (q0 ? 63 - __clzll((long long)q0) : 0);
    }
    int mv = __shfl(pick < 64 ? (int)lab[0] : (int)lab[1], pick & 63, 64);
    if (forced && forced[g] < CZ_NLABELS) mv = forced[g];
    // the record of this ply: state before the move, mover, children and their visits
    const int ply = sp.ply[g];
    if (ply < sp.max_plies) {
        uint8_t *rec = sp.hist + ((size_t)g * sp.max_plies + ply) * CZ_REC_BYTES;
        const uint8_t *rb = t.root_board + (size_t)g * CZD_BOARD_LDS;
        for (int i = lane; i < CZ_NSQ; i += 64) rec[i] = rb[i];
        uint16_t *labs = reinterpret_cast<uint16_t *>(rec + CZ_REC_LABELS), *vis = reinterpret_cast<uint16_t *>(rec + CZ_REC_VISITS);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = lane + 64 * r;
            labs[i] = lab[r];
            vis[i] = (uint16_t)(N[r] > 65535 ? 65535 : N[r]);
        }
        const bool sat = __ballot(N[0] > 65535 || N[1] > 65535) != 0ull;
        if (lane == 0) {
            rec[CZ_REC_SIDE] = t.root_side[g];
            rec[CZ_REC_COUNT] = (uint8_t)n;
            rec[CZ_REC_Z] = 0;
            rec[CZ_REC_FLAGS] = sat ? 1 : 0;   // bit 0: a visit count saturated the 16-bit field
            *reinterpret_cast<uint16_t *>(rec + CZ_REC_PLY) = (uint16_t)ply;
        }
    }
    if (lane == 0) { played[g] = (uint16_t)mv; sp.ply[g] = ply + 1; }
}

// The game-end tests of selfplay (main.py:1532-1545) on the position after the move, z for every recorded ply, and —
// reseed != 0 — MCTS_tree.reload / GameBoard.reload for the next game of the slot (:1549-1551, :1494).
// fin_n[g] = number of records the finished game hands to the ring (0: not finished, or dropped).
__global__ __launch_bounds__(64) void k_sp_adjudicate(CzTrees t, CzSelfplay sp, int G, int reseed, const uint16_t *__restrict__ played,
                                                      int32_t *__restrict__ fin_n) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G) return;
    if (!sp.active[g]) { if (lane == 0) fin_n[g] = 0; return; }
    // asynchronous plies: only the slots that just moved (or stalled) can have ended their game
    if (played && played[g] == 0xFFFF && !sp.stalled[g]) { if (lane == 0) fin_n[g] = 0; return; }
    const uint8_t *rb = t.root_board + (size_t)g * CZD_BOARD_LDS;
    const int c0 = rb[lane], c1 = (lane + 64 < CZ_NSQ) ? rb[lane + 64] : 0;
    const bool Kmiss = (__ballot(c0 == 1) | __ballot(c1 == 1)) == 0ull;
    const bool kmiss = (__ballot(c0 == 8) | __ballot(c1 == 8)) == 0ull;
    const int ply = sp.ply[g];
    const bool stalled = sp.stalled[g] != 0 || (t.status[g] & CZ_ST_BAD_ADVANCE) != 0;
    const bool decided = Kmiss || kmiss;
    const bool draw = !decided && (t.root_rr[g] >= 60 || ply >= sp.max_plies);
    if (!(decided || draw || stalled)) { if (lane == 0) fin_n[g] = 0; return; }
    // winner: 'K' missing -> "b", 'k' missing -> "w" (main.py:1534-1537); side codes 1 / 0
    const int winner = Kmiss ? 1 : 0;
    const int n = stalled ? 0 : min(ply, sp.max_plies);
    for (int j = lane; j < n; j += 64) {
        uint8_t *rec = sp.hist + ((size_t)g * sp.max_plies + j) * CZ_REC_BYTES;
        // z[current_players == winner] = 1, else -1 (main.py:1538-1539); zeros for a tie (:1543)
        const int z = decided ? (rec[CZ_REC_SIDE] == winner ? 1 : -1) : 0;
        rec[CZ_REC_Z] = (uint8_t)(int8_t)z;
    }
    if (lane == 0) {
        fin_n[g] = n;
        atomicAdd((unsigned long long *)&sp.stats[CZ_SP_GAMES], 1ull);
        if (stalled) atomicAdd((unsigned long long *)&sp.stats[CZ_SP_STALLED], 1ull);
        else if (decided) atomicAdd((unsigned long long *)&sp.stats[winner ? CZ_SP_BLACK_WINS : CZ_SP_RED_WINS], 1ull);
        else atomicAdd((unsigned long long *)&sp.stats[CZ_SP_DRAWS], 1ull);
        atomicAdd((unsigned long long *)&sp.stats[CZ_SP_PLIES], (unsigned long long)n);
    }
    if (reseed) seed_slot(t, sp, g, lane);
    else if (lane == 0) sp.active[g] = 0;
}

// Copies the records of the games k_sp_adjudicate finished to ring[(offset[g] + j) % ring_records].  The offsets are an
// exclusive prefix sum of fin_n computed by the caller (record order = game order: deterministic, no atomics).
__global__ __launch_bounds__(64) void k_sp_flush(CzSelfplay sp, int G, const int32_t *__restrict__ fin_n,
                                                 const long long *__restrict__ offset, uint8_t *__restrict__ ring,
                                                 long long ring_records, const long long *__restrict__ read_cursor) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G) return;
    const int n = fin_n[g];
    if (n <= 0) return;
    const long long off = offset[g];
    if (read_cursor && off + n - *read_cursor > ring_records) {   // would overwrite records nobody has read yet
        if (lane == 0) atomicAdd((unsigned long long *)&sp.stats[CZ_SP_DROPPED], (unsigned long long)n);
        return;
    }
    for (int j = 0; j < n; ++j) {
        const uint4 *src = reinterpret_cast<const uint4 *>(sp.hist + ((size_t)g * sp.max_plies + j) * CZ_REC_BYTES);
        uint4 *dst = reinterpret_cast<uint4 *>(ring + (size_t)((off + j) % ring_records) * CZ_REC_BYTES);
        if (lane < CZ_REC_BYTES / 16) dst[lane] = src[lane];
    }
}

}  // namespace

int czk_selfplay_seed(cz_ctx *c, const uint8_t *boards, const uint8_t *side, const int32_t *rr) {
    hipLaunchKernelGGL(k_sp_seed, dim3(c->G), dim3(64), 0, c->stream, c->t, c->sp, c->G, boards, side, rr);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_selfplay_choose(cz_ctx *c, const float *gamma, const float *u, const uint16_t *forced, double temperature, float eps,
                        int min_sims, uint16_t *played) {
    hipLaunchKernelGGL(k_sp_choose, dim3(c->G), dim3(64), 0, c->stream, c->t, c->sp, c->G, gamma, u, forced, 1.0 / temperature, eps, min_sims, played);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_selfplay_adjudicate(cz_ctx *c, int reseed, const uint16_t *played, int32_t *fin_n) {
    hipLaunchKernelGGL(k_sp_adjudicate, dim3(c->G), dim3(64), 0, c->stream, c->t, c->sp, c->G, reseed, played, fin_n);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_selfplay_flush(cz_ctx *c, const int32_t *fin_n, const long long *offset, uint8_t *ring, long long ring_records,
                       const long long *read_cursor) {
    hipLaunchKernelGGL(k_sp_flush, dim3(c->G), dim3(64), 0, c->stream, c->sp, c->G, fin_n, offset, ring, ring_records, read_cursor);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}
