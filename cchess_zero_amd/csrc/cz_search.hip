// cz_search.hip — lock-step MCTS over G independent game trees, one wave64 workgroup per tree.
//
// Restates (chengstone/cchess-zero main.py, search_threads = 1 semantics):
//   leaf_node                :93-206    MCTS_tree.main      :473-493
//   start_tree_search        :350-435   update_tree         :272-276
// Arithmetic follows the reference under NumPy >= 2 scalar promotion:
//   priors   float32, tot_p = float32(1e-8) + sequential float32 adds in move order
//   W, Q     float32; virtual loss applied as (W + -3) + 3 in float32 before W += v
//   score    float64: Q + float64(float32(5*P)) * sqrt(float64(parent.N [+3 below the root])) / (1+N)
//   argmax   first maximum in generation order (Python max()), NaN never wins unless it is first
// The HBM traffic per simulation is a few KB (three coalesced sibling loads per level, one board,
// one 2086-logit gather); everything a wave re-reads sits in LDS.
#include "cz_internal.h"

#include <math.h>

namespace {

// ---- reset: MCTS_tree.__init__ / reload, main.py:235-259 ---------------------------------------
__device__ __forceinline__ void reset_tree(const CzTrees &t, int g, int lane, const uint8_t *__restrict__ boards,
                                           const uint8_t *__restrict__ side, const int32_t *__restrict__ rr) {
    for (int i = lane; i < CZD_BOARD_LDS; i += 64)
        t.root_board[(size_t)g * CZD_BOARD_LDS + i] = i < CZ_NSQ ? boards[(size_t)g * CZ_NSQ + i] : 0;
    if (lane == 0) {
        t.root_side[g] = side[g] ? 1 : 0;
        t.root_rr[g] = rr ? rr[g] : 0;
        t.root_node[g] = 0; t.n_nodes[g] = 1; t.status[g] = 0; t.sims[g] = 0; t.last_depth[g] = 0; t.root_ply[g] = 0;
        t.pend_kind[g] = 0; t.pend_leaf[g] = 0; t.pend_value[g] = 0.f; t.pend_side[g] = 0; t.pend_nmoves[g] = 0;
        init_root(view_of(t, g), 0);
    }
    ec_clear_tree(t, g, lane, 64);
}

__global__ __launch_bounds__(64) void k_reset(CzTrees t, const uint8_t *__restrict__ boards,
                                              const uint8_t *__restrict__ side, const int32_t *__restrict__ rr, int G,
                                              const uint8_t *__restrict__ which) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G || (which && !which[g])) return;   // cz_search_reload: only the trees whose game is over start afresh
    reset_tree(t, g, lane, boards, side, rr);
}

// ---- greedy re-rooting driver: the temperature -> 0 limit of get_action (main.py:1332-1341: softmax(log(visits) / 1e-3) is
// the most visited child) for the trees whose search is complete, and check_end + reload (main.py:1380-1392,255-258) for the
// games that are then over.  bench.py's search loop and any "play the strongest move" loop: no host round trip, no torch op.
__global__ __launch_bounds__(64) void k_pick_ready(CzTrees t, int G, int32_t *__restrict__ thr, int next_thr,
                                                   uint16_t *__restrict__ played, uint8_t *__restrict__ ready_out,
                                                   unsigned long long *__restrict__ banked) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G) return;
    const int sims = t.sims[g];
    const bool ready = sims >= thr[g] || (t.status[g] & CZ_ST_POOL_EXHAUSTED) != 0;
    uint16_t label = 0xFFFF;
    if (ready) {
        const TreeView v = view_of(t, g);
        const int root = t.root_node[g];
        const int cb = v.child_begin[root];
        const int n = cb < 0 ? 0 : v.child_count[root];
        // first maximum of N in generation order (Python max() over root.child.items())
        int bn = -1, bi = 0x7fffffff;
        for (int i = lane; i < n; i += 64) {
            const int x = v.N[cb + i];
            if (x > bn) { bn = x; bi = i; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int on = __shfl_xor(bn, d, 64), oi = __shfl_xor(bi, d, 64);
            if (on > bn || (on == bn && oi < bi)) { bn = on; bi = oi; }
        }
        if (n > 0) label = v.move[cb + bi];
        if (lane == 0) {
            thr[g] = next_thr;
            if (banked) atomicAdd(banked, (unsigned long long)sims);
        }
    }
    if (lane == 0) { played[g] = label; if (ready_out) ready_out[g] = ready ? 1 : 0; }
}

__global__ __launch_bounds__(64) void k_reload_finished(CzTrees t, int G, const uint8_t *__restrict__ ready,
                                                        const uint16_t *__restrict__ played, const uint8_t *__restrict__ boards,
                                                        const uint8_t *__restrict__ side, const int32_t *__restrict__ rr,
                                                        unsigned long long *__restrict__ reloaded) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G || !ready[g]) return;
    const uint8_t *b = t.root_board + (size_t)g * CZD_BOARD_LDS;
    const int c0 = b[lane], c1 = lane + 64 < CZ_NSQ ? b[lane + 64] : 0;
    const bool hasK = __ballot(c0 == 1 || c1 == 1) != 0ull, hask = __ballot(c0 == 8 || c1 == 8) != 0ull;
    // check_end (main.py:1380-1392): a king is gone or 60 plies without a capture; a root without a move to play is over too
    const bool over = !hasK || !hask || t.root_rr[g] >= 60 || played[g] == 0xFFFF;
    if (!over) return;
    reset_tree(t, g, lane, boards, side, rr);
    if (lane == 0 && reloaded) atomicAdd(reloaded, 1ull);
}

// ---- K4: selection descent + leaf preparation ----------------------------------------------------
struct Cand { double s; int i; };
__device__ __forceinline__ Cand better(Cand a, Cand b) {
    // first maximum wins: larger score, ties to the smaller index
    if (b.s > a.s || (b.s == a.s && b.i < a.i)) return b;
    return a;
}

// Zobrist key of the position in LDS (the same function as cz_hash; 0 is reserved for "empty" in the evaluation cache)
__device__ __forceinline__ unsigned long long wave_position_key(const uint8_t *b, int side, const uint64_t *__restrict__ zob, int lane) {
    unsigned long long h = 0ull;
    const int c0 = b[lane];
    if (c0) h = zob[c0 * CZ_NSQ + lane];
    if (lane + 64 < CZ_NSQ) { const int c1 = b[lane + 64]; if (c1) h ^= zob[c1 * CZ_NSQ + lane + 64]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) h ^= __shfl_xor(h, d, 64);
    if (side) h ^= zob[15 * CZ_NSQ];
    // the key is wave-uniform: keep it in scalar registers (the select kernel sits exactly at its 64-VGPR budget)
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)h), hi = __builtin_amdgcn_readfirstlane((uint32_t)(h >> 32));
    h = ((unsigned long long)hi << 32) | lo;
    return h ? h : 1ull;
}
__device__ __forceinline__ int ec_bucket(unsigned long long key) { return (int)((key >> 17) & (CZ_EC_BUCKETS - 1)); }
__device__ __forceinline__ uint32_t xc_bucket(unsigned long long key) { return (uint32_t)(key >> 24); }   // cross-tree table: other key bits
// The position itself, packed: 90 squares x 4 bits (piece codes 0..14) in 12 dwords, side to move in the top nibble of the
// last one.  Lane j < 12 packs squares 8j .. 8j+7 (LDS bytes 90..95 of the board are zero).  A cache entry carries this
// image of its node's position and a hit is only taken when it equals the leaf's: the 64-bit key finds the candidate, the
// position decides — a key collision cannot lend a wrong node.
__device__ __forceinline__ uint32_t wave_pack_board(const uint8_t *b, int side, int lane) {
    uint32_t pk = 0u;
    if (lane < 12) {
        const uint32_t w0 = ((const uint32_t *)b)[2 * lane], w1 = ((const uint32_t *)b)[2 * lane + 1];
        pk = (w0 & 15u) | ((w0 >> 4) & 0xF0u) | ((w0 >> 8) & 0xF00u) | ((w0 >> 12) & 0xF000u);
        pk |= ((w1 & 15u) | ((w1 >> 4) & 0xF0u) | ((w1 >> 8) & 0xF00u) | ((w1 >> 12) & 0xF000u)) << 16;
        if (lane == 11) pk |= (uint32_t)side << 28;
    }
    return pk;
}

// COMPACT: the leaf planes of the trees that need a net evaluation are written to consecutive rows handed out by an
// atomic counter (t.evcnt[parity]); t.slot_of[g] records the row (or -1), the other counter is zeroed for the next
// step.  Terminal / drawn / parked trees then cost the net nothing (they are 9 % of the simulations on the bench
// workload), and which row a tree gets does not matter: every row of the net is computed independently.
// Terminal simulations complete INSIDE the selection kernel (`extra` > 0).  A simulation that ends on a king capture or
// on the 60-ply rule (main.py:409-416) needs no net evaluation — the reference returns before push_queue — so there is no
// reason to spend one of the net batch's rows (and a whole lock-step) on it: the wave backs the value up along the path
// it has just walked (same float32 arithmetic as k_expand_backup), counts the simulation and starts the next descent, up
// to `extra` times per launch, until it stands on a leaf that does need the net.  Simulations of one tree are still
// strictly sequential, so the tree after N simulations is bit-identical to the one-simulation-per-step schedule; what
// changes is that a step now completes 1 / (1 - f) simulations per net row (f = share of terminal simulations, 8.5 % on
// the bench workload).  sim_target > 0 stops a tree at that many completed simulations (cz_search_set_sim_target).
template <typename T, bool COMPACT, bool CACHE, bool XC = false>
__device__ __forceinline__ void select_body(const CzTrees &t, const CzTables &tab, int G, int mode,
                                            const uint8_t *__restrict__ active, T *__restrict__ planes, int C,
                                            T one, uint8_t *__restrict__ needs_eval, int parity, int sim_target, int extra) {
    __shared__ __attribute__((aligned(16))) uint8_t b[CZD_BOARD_LDS];
    __shared__ __attribute__((aligned(16))) uint8_t b0[CZD_BOARD_LDS];   // the root board, to restart a descent from
    __shared__ uint16_t stage[64 * CZD_STAGE_STRIDE];
    __shared__ uint16_t mv[CZD_MAXMOVES];
    __shared__ int path_s[CZ_PATH_MAX];
    __shared__ CzdGroupLds GL;
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G) return;
    T *pl = planes ? planes + (size_t)g * 90 * C : nullptr;
    const bool parked = (active && !active[g]) || (t.status[g] & ~CZ_ST_BAD_ADVANCE) != 0;
    int kind = 0, leaf = 0, depth = 0, done_here = 0;
    float pend = 0.f;
    unsigned long long key = 0ull;
    uint32_t pk = 0u;   // evaluation cache: the leaf position packed (wave_pack_board), lanes 0..11
    int side = t.root_side[g];
    if (!parked) {
        for (int i = lane; i < CZD_BOARD_LDS / 4; i += 64) {
            const uint32_t w = ((const uint32_t *)(t.root_board + (size_t)g * CZD_BOARD_LDS))[i];
            ((uint32_t *)b)[i] = w;
            ((uint32_t *)b0)[i] = w;
        }
        __syncthreads();
        const TreeView v = view_of(t, g);
        const int root = t.root_node[g];
        const int root_rr = t.root_rr[g], root_side = side, s0 = t.sims[g];
        // kings present on the root board ('K' = 1, 'k' = 8)
        const int c0 = b[lane], c1 = (lane + 64 < CZ_NSQ) ? b[lane + 64] : 0;
        const bool Kmiss0 = (__ballot(c0 == 1) | __ballot(c1 == 1)) == 0ull;
        const bool kmiss0 = (__ballot(c0 == 8) | __ballot(c1 == 8)) == 0ull;
        // wave-uniform values are moved to scalar registers explicitly (readfirstlane / readlane): the compiler cannot prove
        // that the result of the butterfly argmax is uniform, and the kernel sits at its 64-VGPR budget
        const int rcb = __builtin_amdgcn_readfirstlane(v.child_begin[root]);
        if (rcb < 0) {
            kind = 3; leaf = root;  // MCTS_tree.main root expansion, main.py:475-487
        } else if (mode != 0) {
            // the root's own record does not change while simulations complete (the root is never backed up, quirk Q2)
            const int rcc = __builtin_amdgcn_readfirstlane((int)v.child_count[root]), rN = __builtin_amdgcn_readfirstlane(v.N[root]);
            // per-launch budgets of simulations completed inside this launch: `extra` terminal ones (cz_search_set_terminal_extra)
            // and, independently of it, CZ_EC_BUDGET evaluation-cache hits — the cache works with terminal_extra = 0 too
            int extra_left = extra, cache_left = CACHE ? CZ_EC_BUDGET : 0;
            for (;;) {   // one descent per iteration
                if (sim_target > 0 && s0 + done_here >= sim_target) { kind = 0; break; }   // this tree has had its playouts
                int rr = root_rr, node = root;
                bool Kmiss = Kmiss0, kmiss = kmiss0;
                side = root_side; depth = 0; kind = 0;
                // One dependent HBM round trip per tree level: every lane fetches, together with the statistics of the
                // child it scores, that child's move label and expansion record (child_begin, child_count); the
                // winner's are then taken from its lane, so the next level starts without touching memory again.
                int cb = rcb, cc = rcc, nN = rN;
                for (;;) {
                    if (cb < 0) { kind = 1; leaf = node; break; }  // not in `expanded`, main.py:357
                    if (cc == 0) { if (lane == 0) t.status[g] |= CZ_ST_NO_MOVES; break; }  // max() of empty, quirk Q7
                    // select_new / get_Q_plus_U_new, main.py:108-116,158-159.  Non-root nodes on the path
                    // carry their virtual loss (N += 3, main.py:403) while their children are scored.
                    const double sq = sqrt((double)(nN + (node != root ? 3 : 0)));
                    Cand best; best.s = -INFINITY; best.i = 0x7FFFFFFF;
                    bool first_nan = false;
                    int cN[2] = {0, 0}, cBeg[2] = {-1, -1}, cCnt[2] = {0, 0}, cSd[2] = {0, 0};
                    float cP[2] = {0.f, 0.f}, cQ[2] = {0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int i = lane + 64 * r;
                        if (i < cc) {
                            cP[r] = v.P[cb + i]; cN[r] = v.N[cb + i]; cQ[r] = v.Q[cb + i];
                            cBeg[r] = v.child_begin[cb + i]; cCnt[r] = v.child_count[cb + i];
                            cSd[r] = v.sd[cb + i];   // (src, dst) stored with the node: no label -> table round trip on the descent
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int i = lane + 64 * r;
                        if (i < cc) {
                            const float cp = 5.0f * cP[r];
                            const double u = (double)cp * sq / (double)(1 + cN[r]);
                            double s = (double)cQ[r] + u;
                            if (s != s) { if (i == 0) first_nan = true; s = -INFINITY; }
                            Cand c; c.s = s; c.i = i;
                            best = better(best, c);
                        }
                    }
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) {
                        Cand o; o.s = __shfl_xor(best.s, d, 64); o.i = __shfl_xor(best.i, d, 64);
                        best = better(best, o);
                    }
                    int bi = __builtin_amdgcn_readfirstlane(best.i);
                    if (__builtin_amdgcn_readfirstlane((int)first_nan)) bi = 0;  // a NaN first element is never displaced by `>`
                    const int c = cb + bi;
                    const int wl = bi & 63, wr = bi >> 6;   // the winner sits in lane wl, round wr
                    const int sd = __builtin_amdgcn_readlane(wr ? cSd[1] : cSd[0], wl);
                    const int nbeg = __builtin_amdgcn_readlane(wr ? cBeg[1] : cBeg[0], wl);
                    const int ncnt = __builtin_amdgcn_readlane(wr ? cCnt[1] : cCnt[0], wl);
                    const int nn = __builtin_amdgcn_readlane(wr ? cN[1] : cN[0], wl);
                    const int src = sd & 0xFF, dst = sd >> 8;
                    const int cap = b[dst];
                    __syncthreads();
                    if (lane == 0) {
                        b[dst] = b[src]; b[src] = 0;   // sim_do_action, main.py:671-672
                        if (depth < CZ_PATH_MAX) { path_s[depth] = c; t.pend_path[(size_t)g * CZ_PATH_MAX + depth] = c; }
                    }
                    __syncthreads();
                    side ^= 1;                  // main.py:392
                    rr = cap ? 0 : rr + 1;      // main.py:393-396
                    ++depth;
                    if (cap == 1) Kmiss = true;
                    if (cap == 8) kmiss = true;
                    if (Kmiss || kmiss) {
                        // main.py:409-414; `side` is the player to move at the child
                        float value = 0.f;
                        if (Kmiss) value = side ? 1.0f : -1.0f;
                        if (kmiss) value = side ? -1.0f : 1.0f;
                        kind = 2; leaf = c; pend = value * -1.0f; break;
                    } else if (rr >= 60) {      // main.py:415-416
                        kind = 2; leaf = c; pend = 0.f; break;
                    }
                    node = c; cb = nbeg; cc = ncnt; nN = nn;
                }
                const bool may = extra_left > 0 && depth <= CZ_PATH_MAX;
                const bool may_cache = cache_left > 0 && depth <= CZ_PATH_MAX;
                bool full = false;
                if (CACHE && kind == 1) {
                    // Evaluation cache.  The net is a pure function of (board, side to move) and every row of a batch is
                    // computed independently of the others, so a position this tree has evaluated before would get the
                    // same priors and value, bit for bit.  An expanded node with the same key lends its children's move
                    // labels and priors and the value its evaluation backed up: the simulation completes here, without
                    // a net row, and the tree is the one the net would have produced.
                    key = wave_position_key(b, side, tab.zob, lane) & t.ec_key_mask;
                    key = key ? key : 1ull;
                    pk = wave_pack_board(b, side, lane);
                    bool hit = false;
                    int src = 0;
                    unsigned long long xk = 0ull;   // XC: the cross-tree bucket's keys, requested together with the tree's own
                    if (may_cache) {
                        const size_t eb0 = (size_t)g * CZ_EC_ENTRIES + (size_t)ec_bucket(key) * 64;
                        const unsigned long long ek = t.ec_key[eb0 + lane];
                        if (XC) xk = czx_key(t)[(size_t)(xc_bucket(key) & t.xc_mask) * 64 + lane];   // one wait for both probes
                        const int en = t.ec_node[eb0 + lane];
                        const float ev = t.ec_val[eb0 + lane];
                        // every entry of the bucket with the leaf's key is a candidate (after a key collision two positions
                        // share a key): the stored position decides
                        for (unsigned long long m = __ballot(ek == key); m; m &= m - 1ull) {
                            const int hl = __ffsll((long long)m) - 1;
                            // the candidate's position against the leaf's: 12 lanes, one dword each
                            const uint32_t lb = lane < 12 ? t.ec_board[(eb0 + hl) * 12 + lane] : 0u;
                            if (__ballot(lb != pk) == 0ull) {
                                src = __builtin_amdgcn_readlane(en, hl);
                                pend = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ev), hl));
                                hit = true;
                                break;
                            }
                            if (lane == 0) t.ec_collisions[g] += 1u;   // same key, another position: not this entry
                        }
                        if (lane == 0) { t.ec_hits[g] += hit ? 1u : 0u; t.ec_lookups[g] += 1u; }
                    }
                    // second level (XC kernels): the context's cross-tree table (cz_search_set_xcache) — a position ANOTHER tree
                    // has had evaluated.  Its entries are self-contained (labels, (src, dst), priors, value) and read-only here.
                    long long xe = -1;
                    if (XC && !hit && may_cache) {
                        const size_t xb0 = (size_t)(xc_bucket(key) & t.xc_mask) * 64;
                        for (unsigned long long m = __ballot(xk == key); m; m &= m - 1ull) {
                            const int hl = __ffsll((long long)m) - 1;
                            const uint32_t lb = lane < 12 ? czx_board(t)[(xb0 + hl) * 12 + lane] : 0u;
                            if (__ballot(lb != pk) == 0ull) { xe = (long long)(xb0 + hl); break; }
                        }
                        if (xe >= 0) { pend = czx_val(t)[xe]; hit = true; }
                        if (lane == 0) { uint32_t *xs = czx_tree_stats(t) + (size_t)g * 8; xs[1] += 1u; if (xe >= 0) xs[0] += 1u; }
                    }
                    if (!hit) break;
                    // the lender: an expanded node of this tree (its children's arrays) or a cross-tree entry's arrays
                    const int scb = (XC && xe >= 0) ? 0 : v.child_begin[src];
                    const int n = (XC && xe >= 0) ? (int)(czx_cnt(t)[xe] & 0xFFFFu) : (int)v.child_count[src];
                    const int begin = t.n_nodes[g];
                    if (begin + n <= t.cap) {   // leaf_node.expand with the lender's priors
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int i = lane + 64 * r;
                            if (i < n) {
                                const int c = begin + i;
                                if (XC && xe >= 0) {
                                    const size_t o = (size_t)xe * CZD_MAXMOVES + i;
                                    v.P[c] = czx_P(t)[o]; v.move[c] = czx_moves(t)[o]; v.sd[c] = czx_sd(t)[o];
                                } else {
                                    v.P[c] = v.P[scb + i]; v.move[c] = v.move[scb + i]; v.sd[c] = v.sd[scb + i];
                                }
                                v.W[c] = 0.f; v.Q[c] = 0.f; v.N[c] = 0; v.parent[c] = leaf; v.child_begin[c] = -1;
                                v.child_count[c] = 0;
                            }
                        }
                        if (lane == 0) { v.child_begin[leaf] = begin; v.child_count[leaf] = (uint16_t)n; t.n_nodes[g] = begin + n; }
                    } else {
                        if (lane == 0) t.status[g] |= CZ_ST_POOL_EXHAUSTED;
                        full = true;   // this simulation is still backed up; the tree then stops, as it does after k_expand_backup
                    }
                } else if (!(kind == 2 && may)) break;
                // back_up_value along the path (main.py:189-194,426-435), exactly as k_expand_backup does it: lane d owns
                // the node of level d; (W + -3) + 3 reproduces the float32 rounding of the virtual loss
                if (lane < depth) {
                    const int n = path_s[lane];
                    const float x = ((depth - 1 - lane) & 1) ? pend * -1.0f : pend;
                    float w = v.W[n];
                    w = w + -3.0f;
                    w = w + 3.0f;
                    const int cnt = v.N[n] + 1;
                    w = w + x;
                    v.N[n] = cnt; v.W[n] = w; v.Q[n] = w / (float)cnt;
                }
                ++done_here;
                if (CACHE && kind == 1) --cache_left; else --extra_left;
                if (full) { kind = 0; break; }
                __threadfence_block();
                __syncthreads();
                for (int i = lane; i < CZD_BOARD_LDS / 4; i += 64) ((uint32_t *)b)[i] = ((const uint32_t *)b0)[i];
                __syncthreads();
            }
            if (lane == 0 && done_here) t.sims[g] = s0 + done_here;
        }
    }
    int nmoves = 0;
    if (kind == 1 || kind == 3) {
        nmoves = czd_wave_movegen(b, side, tab.lut, GL, stage, mv, lane);  // main.py:374 / :483
        if (nmoves < 0) { if (lane == 0) t.status[g] |= CZ_ST_MOVE_OVERFLOW; kind = 0; nmoves = 0; }
    }
    if constexpr (COMPACT) {
        int slot = -1;
        if (kind == 1 || kind == 3) {
            if (lane == 0) slot = atomicAdd(&t.evcnt[parity], 1);
            slot = __shfl(slot, 0, 64);
        }
        pl = slot >= 0 ? planes + (size_t)slot * 90 * C : nullptr;
        if (lane == 0) {
            t.slot_of[g] = slot;
            if (g == 0) t.evcnt[parity ^ 1] = 0;   // nobody touches the other counter until the next step's select
        }
    }
    if (kind == 1 || kind == 3) {
        for (int i = lane; i < nmoves; i += 64) t.pend_moves[(size_t)g * CZD_MAXMOVES + i] = mv[i];
        if (pl) czd_wave_encode_planes<T>(b, side, 1, pl, C, one, lane);  // generate_inputs, main.py:362 / :477
    } else if (pl && !COMPACT) {
        for (int e = lane; e < 90 * C; e += 64) pl[e] = (T)0;
    }
    if (CACHE && (kind == 1 || kind == 3)) {
        if (kind == 3) {
            key = wave_position_key(b, side, tab.zob, lane) & t.ec_key_mask;
            key = key ? key : 1ull;
            pk = wave_pack_board(b, side, lane);
        }
        if (lane == 0) t.pend_key[g] = key;
        if (lane < 12) t.pend_board[(size_t)g * 12 + lane] = pk;   // k_expand_backup files it with the cache entry
    }
    if (lane == 0) {
        t.pend_kind[g] = kind; t.pend_leaf[g] = leaf; t.pend_value[g] = pend;
        t.pend_side[g] = (uint8_t)side; t.pend_nmoves[g] = (uint16_t)nmoves;
        t.pend_depth[g] = depth;
        if (!parked) t.last_depth[g] = (int16_t)depth;
        if (needs_eval) needs_eval[g] = (kind == 1 || kind == 3) ? 1 : 0;
    }
}

// The two kernels around select_body.  The product step (no compaction) is capped at 80 SGPRs: 8192 trees are exactly 32
// waves per CU, and at the 82+ SGPRs hipcc picks only 7 waves fit a SIMD (MI355X_MICROARCH.md: floor(800 / (ceil(sgpr / 16)
// * 16 + 16))), so the 4 left-over waves per CU cost a second pass.  The compact variant (parking mode of self-play: the
// batch is not full anyway) keeps hipcc's own allocation — under the cap it would spill to scratch.
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(80))) void k_select(CzTrees t, CzTables tab, int G, int mode,
                                               const uint8_t *__restrict__ active, T *__restrict__ planes, int C,
                                               T one, uint8_t *__restrict__ needs_eval, int parity, int sim_target, int extra) {
    select_body<T, false, false>(t, tab, G, mode, active, planes, C, one, needs_eval, parity, sim_target, extra);
}
template <typename T>
__global__ __launch_bounds__(64) void k_select_compact(CzTrees t, CzTables tab, int G, int mode,
                                                       const uint8_t *__restrict__ active, T *__restrict__ planes, int C,
                                                       T one, uint8_t *__restrict__ needs_eval, int parity, int sim_target, int extra) {
    select_body<T, true, false>(t, tab, G, mode, active, planes, C, one, needs_eval, parity, sim_target, extra);
}
// With the evaluation cache (cz_search_set_eval_cache): the same descent, plus the lookup at the leaf and the expansion from
// a lender node inside the launch.
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(80))) void k_select_cache(CzTrees t, CzTables tab, int G, int mode,
                                                     const uint8_t *__restrict__ active, T *__restrict__ planes, int C,
                                                     T one, uint8_t *__restrict__ needs_eval, int parity, int sim_target, int extra) {
    select_body<T, false, true>(t, tab, G, mode, active, planes, C, one, needs_eval, parity, sim_target, extra);
}
// ... and with the cross-tree level behind it (cz_search_set_xcache).  Their own instantiations: the extra probe costs registers
// (65 VGPRs: 7 waves per SIMD) that the plain cache kernels (59) do not pay.
template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(80))) void k_select_xcache(CzTrees t, CzTables tab, int G, int mode,
                                                      const uint8_t *__restrict__ active, T *__restrict__ planes, int C,
                                                      T one, uint8_t *__restrict__ needs_eval, int parity, int sim_target, int extra) {
    select_body<T, false, true, true>(t, tab, G, mode, active, planes, C, one, needs_eval, parity, sim_target, extra);
}
template <typename T>
__global__ __launch_bounds__(64) void k_select_compact_xcache(CzTrees t, CzTables tab, int G, int mode,
                                                              const uint8_t *__restrict__ active, T *__restrict__ planes, int C,
                                                              T one, uint8_t *__restrict__ needs_eval, int parity, int sim_target, int extra) {
    select_body<T, true, true, true>(t, tab, G, mode, active, planes, C, one, needs_eval, parity, sim_target, extra);
}
template <typename T>
__global__ __launch_bounds__(64) void k_select_compact_cache(CzTrees t, CzTables tab, int G, int mode,
                                                             const uint8_t *__restrict__ active, T *__restrict__ planes, int C,
                                                             T one, uint8_t *__restrict__ needs_eval, int parity, int sim_target, int extra) {
    select_body<T, true, true>(t, tab, G, mode, active, planes, C, one, needs_eval, parity, sim_target, extra);
}

// ---- K5 + K6: expansion and value backup ---------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T x);
template <> __device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ float to_f32<uint16_t>(uint16_t x) { return czd_bf16_bits_to_f32(x); }

// FC = false: the raw prior of a move is read from a full logits row [2086] of type T.
// FC = true : `logits` is instead the head-conv output z [G][90][3] f32 and the policy FC (policy_value_network.py:62-63,
//             180 -> 2086, weight fcw [2086][180] f32, bias fcb) is evaluated in place for the <= 128 labels the
//             expansion needs — the search never looks at the other ~2046 logits of a position, so computing and
//             storing them (68 MB per step at 8192 trees) is wasted work.  Arithmetic, so that the CPU oracle can
//             restate it bit for bit (float32, multiply and add rounded separately, -ffp-contract=off):
//               x[k] = z[k / 2][k % 2] for k < 180 (the (h,w,c) flatten), p[k] = w[k] * x[k], p[180..191] = 0;
//               s[l] = ((p[12l] + p[12l+1]) + ...) + p[12l+11] for l = 0..15 (s[15] = 0);
//               s[l] += s[15-l]; s[l] += s[(l & 8) | (7 - (l & 7))]; s[l] += s[(l & 12) | (3 - (l & 3))]; s[l] += s[l ^ 1]
//               (each step on all 16 values at once); logit = s[0] + bias.
// register caps: 8192 trees are 32 waves per CU, which only fit at <= 64 VGPRs and <= 80 SGPRs per wave (see k_select)
template <typename T, bool FC>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(80), amdgpu_num_vgpr(64))) void k_expand_backup(CzTrees t, CzTables tab, int G, const T *__restrict__ logits,
                                                      const T *__restrict__ value, const float *__restrict__ fcw,
                                                      const float *__restrict__ fcb, int compact_parity) {
    __shared__ float pr[CZD_MAXMOVES];
    __shared__ float tot_s;
    __shared__ __attribute__((aligned(16))) float xin[FC ? 192 : 4];
    __shared__ uint16_t ridx[FC ? CZD_MAXMOVES : 2];
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G) return;
    // compact batches (compact_parity >= 0): tree g's leaf sits in row slot_of[g] of z / value; tree 0 also books the
    // step's row count into the running totals (the flop accounting of bench.py)
    int row = g;
    if (FC && compact_parity >= 0) {
        row = t.slot_of[g];
        if (g == 0 && lane == 0) { t.evtotal[0] += (unsigned long long)t.evcnt[compact_parity]; t.evtotal[1] += 1ull; }
    }
    const int kind = t.pend_kind[g];
    if (kind == 0) return;
    const TreeView v = view_of(t, g);
    const int leaf = t.pend_leaf[g];
    float val;
    if (kind == 1 || kind == 3) {
        // leaf_node.expand, main.py:175-187; flip_policy for black, main.py:371-372,1153-1155
        const int n = t.pend_nmoves[g];
        const int sd = t.pend_side[g];
        const int begin = t.n_nodes[g];
        const bool fits = begin + n <= t.cap;
        if (fits) {
            uint16_t lab[2];
            if constexpr (FC) {
                const float *zg = reinterpret_cast<const float *>(logits) + (size_t)row * 270;
                for (int i = lane; i < 270; i += 64) {
                    const float zv = zg[i];
                    const int cell = i / 3, ch = i - cell * 3;
                    if (ch < 2) xin[cell * 2 + ch] = zv;
                }
                if (lane < 12) xin[180 + lane] = 0.f;
                // the weight row of every move, resolved for all moves at once (label -> flip_policy's unflip for black):
                // two dependent round trips for the whole node instead of two per group of four moves
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int i = lane + 64 * r;
                    lab[r] = i < n ? t.pend_moves[(size_t)g * CZD_MAXMOVES + i] : (uint16_t)0;
                    if (i < n) ridx[i] = (uint16_t)(sd ? tab.unflip[lab[r]] : lab[r]);
                }
                __syncthreads();
                // four moves per pass, 16 lanes per move: lane l of a group owns inputs k = 12 l .. 12 l + 11 (three
                // coalesced float4 of the weight row), sums its 12 products in order, then the 16 partial sums are
                // folded with four symmetric DPP steps (i <-> 15-i, i <-> 7-i within halves, i <-> 3-i within quads,
                // i <-> i^1); every lane of the group ends up with the same total.  One lane per move gathering its
                // whole row (45 strided loads) made the texture addresser the bottleneck.
                const int l16 = lane & 15, grp = lane >> 4;
                float xr[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) xr[k] = xin[l16 * 12 + k];
                // THREE passes (12 moves) per iteration: all their weight rows are requested before any is reduced, so a
                // 40-move node waits for four gathers instead of ten (four passes would need 76 registers: 6 waves per SIMD)
                constexpr int NH = 3;
#pragma unroll 1
                for (int base = 0; base < n; base += 4 * NH) {
                    int idx[NH];
                    float sum[NH];
                    float4 wa[NH], wb[NH], wc4[NH];
                    bool on[NH];
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        const int i = base + 4 * h + grp;
                        on[h] = i < n;
                        idx[h] = on[h] ? (int)ridx[i] : 0;
                        sum[h] = 0.f;
                    }
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        wa[h] = wb[h] = wc4[h] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (on[h] && l16 < 15) {
                            const float4 *w4 = reinterpret_cast<const float4 *>(fcw + (size_t)idx[h] * 180 + l16 * 12);
                            wa[h] = w4[0]; wb[h] = w4[1]; wc4[h] = w4[2];
                        }
                    }
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        if (on[h] && l16 < 15) {
                            const float4 a = wa[h], bq = wb[h], cq = wc4[h];
                            float sm = a.x * xr[0];
                            sm = sm + a.y * xr[1];
                            sm = sm + a.z * xr[2];
                            sm = sm + a.w * xr[3];
                            sm = sm + bq.x * xr[4];
                            sm = sm + bq.y * xr[5];
                            sm = sm + bq.z * xr[6];
                            sm = sm + bq.w * xr[7];
                            sm = sm + cq.x * xr[8];
                            sm = sm + cq.y * xr[9];
                            sm = sm + cq.z * xr[10];
                            sm = sm + cq.w * xr[11];
                            sum[h] = sm;
                        }
                    }
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        float sm = sum[h];
                        sm = sm + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sm), 0x140, 0xF, 0xF, false));  // row_mirror
                        sm = sm + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sm), 0x141, 0xF, 0xF, false));  // row_half_mirror
                        sm = sm + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sm), 0x1B, 0xF, 0xF, false));   // quad_perm [3,2,1,0]
                        sm = sm + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sm), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
                        if (on[h] && l16 == 0) pr[base + 4 * h + grp] = sm + fcb[idx[h]];
                    }
                }
            } else {
                const T *lg = logits + (size_t)g * CZ_NLABELS;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int i = lane + 64 * r;
                    lab[r] = 0;
                    if (i < n) {
                        lab[r] = t.pend_moves[(size_t)g * CZD_MAXMOVES + i];
                        pr[i] = to_f32<T>(lg[sd ? tab.unflip[lab[r]] : lab[r]]);
                    }
                }
            }
            __syncthreads();
            if (lane == 0) {
                float tot = (float)1e-8;  // tot_p = 1e-8; tot_p += mov_p (np.float32), main.py:176,184
                for (int i = 0; i < n; ++i) tot = tot + pr[i];
                tot_s = tot;
            }
            __syncthreads();
            const float tot = tot_s;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = lane + 64 * r;
                if (i < n) {
                    const int c = begin + i;
                    v.P[c] = pr[i] / tot;  // n.P /= tot_p, main.py:186-187
                    v.W[c] = 0.f; v.Q[c] = 0.f; v.N[c] = 0; v.parent[c] = leaf; v.child_begin[c] = -1;
                    v.child_count[c] = 0; v.move[c] = lab[r]; v.sd[c] = tab.srcdst[lab[r]];
                }
            }
            if (lane == 0) { v.child_begin[leaf] = begin; v.child_count[leaf] = (uint16_t)n; t.n_nodes[g] = begin + n; }
        } else if (lane == 0) {
            t.status[g] |= CZ_ST_POOL_EXHAUSTED;
        }
        val = to_f32<T>(value[FC ? row : g]) * -1.0f;  // return value[0] * -1, main.py:384
        if (t.ec_key && fits) {
            // evaluation cache: this node now holds the priors of its position (its children's P), `val` is what its
            // evaluation backs up.  First empty entry of the key's bucket; a full bucket replaces a pseudo-random entry.
            const unsigned long long key = t.pend_key[g];
            const size_t eb0 = (size_t)g * CZ_EC_ENTRIES + (size_t)ec_bucket(key) * 64;
            const unsigned long long ek = t.ec_key[eb0 + lane];
            // already remembered?  Only an entry with this key AND this position counts (the lookup may have been skipped by
            // its per-launch budget); a same-key entry holding ANOTHER position — a key collision — does not keep this one out
            bool have = false;
            const uint32_t mine = lane < 12 ? t.pend_board[(size_t)g * 12 + lane] : 0u;
            for (unsigned long long m = __ballot(ek == key); m && !have; m &= m - 1ull) {
                const int hl = __ffsll((long long)m) - 1;
                const uint32_t lb = lane < 12 ? t.ec_board[(eb0 + hl) * 12 + lane] : 0u;
                have = __ballot(lb != mine) == 0ull;
            }
            if (!have) {
                const unsigned long long em = __ballot(ek == 0ull);
                int slot = em ? __ffsll((long long)em) - 1 : (int)((key >> 40) & 63);
                if (lane == slot) { t.ec_key[eb0 + lane] = key; t.ec_node[eb0 + lane] = leaf; t.ec_val[eb0 + lane] = val; }
                if (lane < 12) t.ec_board[(eb0 + slot) * 12 + lane] = t.pend_board[(size_t)g * 12 + lane];
            }
            // cross-tree level (cz_search_set_xcache): file the evaluation for the OTHER trees as well — a write-once entry in an
            // empty slot of the key's bucket, claimed by compare-and-swap on the key (two trees expanding the same position in this
            // launch: one of them wins the slot, the other finds the key taken or loses the swap and leaves).  Readers are the
            // select kernels of later launches, so the payload needs no flag: the kernel boundary publishes it.
            if (t.xc_base) {
                const size_t xb0 = (size_t)(xc_bucket(key) & t.xc_mask) * 64;
                const unsigned long long xk = czx_key(t)[xb0 + lane];
                const unsigned long long xm = __ballot(xk == 0ull);
                if (__ballot(xk == key) == 0ull) {
                    // this position's game ply: re-roots of the tree + levels below the root (the priority of the entry: low = shared)
                    const uint32_t myply = (uint32_t)min((int)t.root_ply[g] + (int)t.pend_depth[g], 0xFFFF);
                    int slot = 0, won = 0, dup = 0;
                    if (xm) {
                        // the first empty slot at or behind a key-dependent position: trees filing different positions into one
                        // bucket in the same launch do not all go for slot 0; a tree that loses the swap to ANOTHER position tries the
                        // next empty slot, up to three times; losing it to the SAME position means it is filed
                        const int rot = (int)((key >> 40) & 63);
                        unsigned long long xr = rot ? (xm >> rot) | (xm << (64 - rot)) : xm;
                        for (int attempt = 0; attempt < 3 && xr && !won && !dup; ++attempt) {
                            slot = (__ffsll((long long)xr) - 1 + rot) & 63;
                            xr &= xr - 1ull;
                            unsigned long long seen = 0ull;
                            if (lane == 0) seen = atomicCAS(&czx_key(t)[xb0 + slot], 0ull, key);
                            seen = __shfl(seen, 0, 64);
                            won = seen == 0ull ? 1 : 0;
                            dup = seen == key ? 1 : 0;
                        }
                    } else {
                        // full bucket: the entry deepest in its game makes room if this position is shallower.  One attempt: a lost
                        // swap (another tree of this launch took the victim) drops the filing.
                        const uint32_t ep = czx_cnt(t)[xb0 + lane] >> 16;
                        uint32_t best = (ep << 6) | (uint32_t)lane;
#pragma unroll
                        for (int o = 32; o >= 1; o >>= 1) best = max(best, (uint32_t)__shfl_xor((int)best, o, 64));
                        if ((best >> 6) > myply) {
                            slot = (int)(best & 63u);
                            const unsigned long long old = __shfl(xk, slot, 64);
                            unsigned long long seen = 0ull;
                            if (lane == 0) seen = atomicCAS(&czx_key(t)[xb0 + slot], old, key);
                            seen = __shfl(seen, 0, 64);
                            won = seen == old ? 1 : 0;
                            if (won && lane == 0) czx_tree_stats(t)[(size_t)g * 8 + 4] += 1u;
                        }
                    }
                    if (won) {
                        const size_t e = xb0 + slot;
                        const int n2 = t.pend_nmoves[g];
                        const int cb2 = begin;   // the children written above: lane l re-reads exactly the elements lane l wrote
                        if (lane < 12) czx_board(t)[e * 12 + lane] = mine;
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const int i = lane + 64 * r;
                            if (i < n2) {
                                czx_P(t)[e * CZD_MAXMOVES + i] = v.P[cb2 + i];
                                czx_moves(t)[e * CZD_MAXMOVES + i] = v.move[cb2 + i];
                                czx_sd(t)[e * CZD_MAXMOVES + i] = v.sd[cb2 + i];
                            }
                        }
                        if (lane == 0) { czx_val(t)[e] = val; czx_cnt(t)[e] = (uint32_t)n2 | (myply << 16); czx_tree_stats(t)[(size_t)g * 8 + 2] += 1u; }
                    } else if (!dup && lane == 0) {
                        czx_tree_stats(t)[(size_t)g * 8 + 3] += 1u;     // filings that found no room (deeper than everything in a full bucket, or lost every swap)
                    }
                }
            }
        }
        if (kind == 3) { if (lane == 0) t.pend_kind[g] = 0; return; }
    } else {
        val = t.pend_value[g];
    }
    // back_up_value on every selected node of the path, main.py:189-194,426-435; the root is never updated (quirk Q2).
    // (W + -3) + 3 reproduces the float32 rounding of the virtual loss.  The value alternates in sign level by level and
    // the levels are independent, so lane d updates the node k_select recorded for level d: one memory round trip for
    // the whole path instead of one per level along the parent pointers.
    const int depth = t.pend_depth[g];
    if (depth <= CZ_PATH_MAX) {
        if (lane < depth) {
            const int n = t.pend_path[(size_t)g * CZ_PATH_MAX + lane];
            const float x = ((depth - 1 - lane) & 1) ? val * -1.0f : val;
            float w = v.W[n];
            w = w + -3.0f;
            w = w + 3.0f;
            const int cnt = v.N[n] + 1;
            w = w + x;
            v.N[n] = cnt; v.W[n] = w; v.Q[n] = w / (float)cnt;
        }
        if (lane == 0) { t.sims[g] += 1; t.pend_kind[g] = 0; }
    } else if (lane == 0) {
        const int root = t.root_node[g];
        int n = leaf;
        float x = val;
        while (n != root) {
            float w = v.W[n];
            w = w + -3.0f;
            w = w + 3.0f;
            const int cnt = v.N[n] + 1;
            w = w + x;
            v.N[n] = cnt; v.W[n] = w; v.Q[n] = w / (float)cnt;
            x = x * -1.0f;
            n = v.parent[n];
        }
        t.sims[g] += 1;
        t.pend_kind[g] = 0;
    }
}

// ---- K4/K5/K6 with k simulations in flight per tree (the reference's `search_threads`, main.py:250,337-348) ------
// Each tree runs up to K descents back to back inside one launch; a descent leaves its virtual loss physically in
// the tree (N += 3, W += -3 on every selected node, main.py:403-404) so that the following descents of the same batch
// are steered away from it, exactly as concurrently running coroutines are in the reference.  A descent that ends on
// a terminal / drawn child completes immediately (main.py:409-435 has no await on that path).  A descent that runs
// into a node whose expansion is already pending in this batch (the reference parks such a coroutine in
// `while node in now_expanding: await asyncio.sleep`, main.py:354-355) is abandoned: its virtual loss is taken back
// and the tree issues no further descents in this step.  The net is then evaluated for all pending leaves at once
// and k_expand_backup_k expands them in descent order and unwinds each path (N -= 3, W += 3, back_up_value).
// With K = 1 this is arithmetically identical to k_select / k_expand_backup (tested).  The asyncio interleaving of
// the reference for K > 1 depends on wall-clock sleeps, so K > 1 is checked against the C oracle's restatement of
// THIS schedule and through invariants, not against reference golden trees.
template <typename T>
__global__ __launch_bounds__(64) void k_select_k(CzTrees t, CzTables tab, int G, int mode, int K, int sim_target,
                                                 const uint8_t *__restrict__ active, T *__restrict__ planes, int C,
                                                 T one, uint8_t *__restrict__ needs_eval) {
    __shared__ __attribute__((aligned(16))) uint8_t b[CZD_BOARD_LDS];
    __shared__ uint16_t stage[64 * CZD_STAGE_STRIDE];
    __shared__ uint16_t mv[CZD_MAXMOVES];
    __shared__ CzdGroupLds GL;
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G) return;
    const bool parked = (active && !active[g]) || (t.status[g] & ~CZ_ST_BAD_ADVANCE) != 0;
    const TreeView v = view_of(t, g);
    const int root = t.root_node[g];
    bool stop = parked;
    int done_now = 0;   // simulations completed inside this launch (terminal / draw)
    // cz_search_set_sim_target: a tree issues no descent beyond its budget of completed simulations (every descent
    // issued here completes: inside this launch or in the following expand_backup_k), so a search ends with EXACTLY the
    // requested number of playouts per tree, as MCTS_tree.main does (main.py:489-493)
    const int budget = sim_target > 0 ? sim_target - t.sims[g] : 0x7FFFFFFF;
    int issued = 0;
    for (int j = 0; j < K; ++j) {
        if (mode != 0 && issued >= budget) stop = true;
        const size_t slot = (size_t)g * K + j;
        T *pl = planes ? planes + slot * 90 * C : nullptr;
        int kind = 0, leaf = 0, depth = 0, side = t.root_side[g];
        if (!stop) {
            __syncthreads();
            for (int i = lane; i < CZD_BOARD_LDS / 4; i += 64)
                ((uint32_t *)b)[i] = ((const uint32_t *)(t.root_board + (size_t)g * CZD_BOARD_LDS))[i];
            __syncthreads();
            int rr = t.root_rr[g];
            int node = root;
            const int c0 = b[lane], c1 = (lane + 64 < CZ_NSQ) ? b[lane + 64] : 0;
            bool Kmiss = (__ballot(c0 == 1) | __ballot(c1 == 1)) == 0ull;
            bool kmiss = (__ballot(c0 == 8) | __ballot(c1 == 8)) == 0ull;
            if (v.child_begin[root] < 0) {
                kind = 3; leaf = root; stop = true;  // root expansion is a step of its own (main.py:475-487)
            } else if (mode == 0) {
                stop = true;
            } else {
                for (;;) {
                    const int cb = v.child_begin[node];
                    bool abandon = false;
                    if (cb == -1) {   // not expanded: this descent owns the expansion
                        kind = 1; leaf = node; ++issued;
                        if (lane == 0) v.child_begin[node] = -2;
                        break;
                    }
                    const int cc = cb >= 0 ? (int)v.child_count[node] : 0;
                    if (cb == -2) abandon = true;                                   // expansion pending in this batch
                    else if (cc == 0) { if (lane == 0) t.status[g] |= CZ_ST_NO_MOVES; abandon = true; }
                    if (abandon) {
                        if (lane == 0)   // take the virtual loss of this descent back (main.py:426-427 without a backup)
                            for (int n = node; n != root; n = v.parent[n]) { v.N[n] -= 3; v.W[n] = v.W[n] + 3.0f; }
                        stop = true;
                        break;
                    }
                    const double sq = sqrt((double)v.N[node]);   // includes the virtual losses currently in flight
                    Cand best; best.s = -INFINITY; best.i = 0x7FFFFFFF;
                    bool first_nan = false;
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const int i = lane + 64 * r;
                        if (i < cc) {
                            const float cp = 5.0f * v.P[cb + i];
                            const double u = (double)cp * sq / (double)(1 + v.N[cb + i]);
                            double s = (double)v.Q[cb + i] + u;   // Q is not recomputed under virtual loss (quirk Q6)
                            if (s != s) { if (i == 0) first_nan = true; s = -INFINITY; }
                            Cand c; c.s = s; c.i = i;
                            best = better(best, c);
                        }
                    }
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) {
                        Cand o; o.s = __shfl_xor(best.s, d, 64); o.i = __shfl_xor(best.i, d, 64);
                        best = better(best, o);
                    }
                    int bi = best.i;
                    if (__shfl((int)first_nan, 0, 64)) bi = 0;
                    const int c = cb + bi;
                    const int sdc = v.sd[c];
                    const int src = sdc & 0xFF, dst = sdc >> 8;
                    const int cap = b[dst];
                    __syncthreads();
                    if (lane == 0) {
                        b[dst] = b[src]; b[src] = 0;
                        v.N[c] += 3; v.W[c] = v.W[c] + -3.0f;   // virtual loss, main.py:403-404
                    }
                    __threadfence_block();
                    __syncthreads();
                    side ^= 1;
                    rr = cap ? 0 : rr + 1;
                    ++depth;
                    if (cap == 1) Kmiss = true;
                    if (cap == 8) kmiss = true;
                    const bool term = Kmiss || kmiss;
                    if (term || rr >= 60) {
                        float value = 0.f;
                        if (term) {
                            if (Kmiss) value = side ? 1.0f : -1.0f;
                            if (kmiss) value = side ? -1.0f : 1.0f;
                            value = value * -1.0f;
                        }
                        if (lane == 0) {   // unwind immediately: main.py:426-435
                            float x = value;
                            for (int n = c; n != root; n = v.parent[n]) {
                                const int cnt = v.N[n] - 3 + 1;
                                float w = v.W[n] + 3.0f;
                                w = w + x;
                                v.N[n] = cnt; v.W[n] = w; v.Q[n] = w / (float)cnt;
                                x = x * -1.0f;
                            }
                        }
                        __threadfence_block();
                        __syncthreads();
                        ++done_now;
                        ++issued;
                        kind = 0;
                        break;
                    }
                    node = c;
                }
            }
        }
        int nmoves = 0;
        if (kind == 1 || kind == 3) {
            nmoves = czd_wave_movegen(b, side, tab.lut, GL, stage, mv, lane);
            if (nmoves < 0) {   // > 128 moves / unlabeled move: report, give the node and the virtual loss back
                if (lane == 0) {
                    t.status[g] |= CZ_ST_MOVE_OVERFLOW;
                    if (kind == 1) {
                        v.child_begin[leaf] = -1;
                        for (int n = leaf; n != root; n = v.parent[n]) { v.N[n] -= 3; v.W[n] = v.W[n] + 3.0f; }
                    }
                }
                if (kind == 1) --issued;
                kind = 0; nmoves = 0; stop = true;
            }
        }
        if (kind == 1 || kind == 3) {
            for (int i = lane; i < nmoves; i += 64) t.pend_moves[slot * CZD_MAXMOVES + i] = mv[i];
            if (pl) czd_wave_encode_planes<T>(b, side, 1, pl, C, one, lane);
        } else if (pl) {
            for (int e = lane; e < 90 * C; e += 64) pl[e] = (T)0;
        }
        if (lane == 0) {
            t.pk_kind[slot] = kind; t.pk_leaf[slot] = leaf; t.pk_value[slot] = 0.f;
            t.pk_side[slot] = (uint8_t)side; t.pk_nmoves[slot] = (uint16_t)nmoves;
            if (kind) t.last_depth[g] = (int16_t)depth;
            if (needs_eval) needs_eval[slot] = kind ? 1 : 0;
        }
        __threadfence_block();
        __syncthreads();
    }
    if (lane == 0 && done_now) t.sims[g] += done_now;
}

template <typename T>
__global__ __launch_bounds__(64) void k_expand_backup_k(CzTrees t, CzTables tab, int G, int K, const T *__restrict__ logits,
                                                        const T *__restrict__ value) {
    __shared__ float pr[CZD_MAXMOVES];
    __shared__ float tot_s;
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G) return;
    const TreeView v = view_of(t, g);
    const int root = t.root_node[g];
    for (int j = 0; j < K; ++j) {
        const size_t slot = (size_t)g * K + j;
        const int kind = t.pk_kind[slot];
        if (kind == 0) continue;
        const int leaf = t.pk_leaf[slot];
        const int n = t.pk_nmoves[slot];
        const int sd = t.pk_side[slot];
        const int begin = t.n_nodes[g];
        const bool fits = begin + n <= t.cap;
        __syncthreads();
        if (fits) {
            const T *lg = logits + slot * CZ_NLABELS;
            uint16_t lab[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = lane + 64 * r;
                lab[r] = 0;
                if (i < n) {
                    lab[r] = t.pend_moves[slot * CZD_MAXMOVES + i];
                    pr[i] = to_f32<T>(lg[sd ? tab.unflip[lab[r]] : lab[r]]);
                }
            }
            __syncthreads();
            if (lane == 0) {
                float tot = (float)1e-8;
                for (int i = 0; i < n; ++i) tot = tot + pr[i];
                tot_s = tot;
            }
            __syncthreads();
            const float tot = tot_s;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int i = lane + 64 * r;
                if (i < n) {
                    const int c = begin + i;
                    v.P[c] = pr[i] / tot;
                    v.W[c] = 0.f; v.Q[c] = 0.f; v.N[c] = 0; v.parent[c] = leaf; v.child_begin[c] = -1;
                    v.child_count[c] = 0; v.move[c] = lab[r]; v.sd[c] = tab.srcdst[lab[r]];
                }
            }
            if (lane == 0) { v.child_begin[leaf] = begin; v.child_count[leaf] = (uint16_t)n; t.n_nodes[g] = begin + n; }
        } else if (lane == 0) {
            t.status[g] |= CZ_ST_POOL_EXHAUSTED;
            v.child_begin[leaf] = -1;   // no longer pending
        }
        if (kind == 1 && lane == 0) {
            float x = to_f32<T>(value[slot]) * -1.0f;
            for (int m = leaf; m != root; m = v.parent[m]) {
                const int cnt = v.N[m] - 3 + 1;   // virtual loss off, visit on (main.py:426-427,190)
                float w = v.W[m] + 3.0f;
                w = w + x;
                v.N[m] = cnt; v.W[m] = w; v.Q[m] = w / (float)cnt;
                x = x * -1.0f;
            }
            t.sims[g] += 1;
        }
        if (lane == 0) t.pk_kind[slot] = 0;
        __threadfence_block();
        __syncthreads();
    }
}

// ---- root children: root.child.items(), main.py:1339 ----------------------------------------------
__global__ __launch_bounds__(64) void k_root_stats(CzTrees t, int G, uint16_t *__restrict__ label, int32_t *__restrict__ N,
                                                   float *__restrict__ Q, float *__restrict__ P, float *__restrict__ W,
                                                   uint16_t *__restrict__ count) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G) return;
    const TreeView v = view_of(t, g);
    const int root = t.root_node[g];
    const int cb = v.child_begin[root];
    const int n = cb < 0 ? 0 : v.child_count[root];
    if (lane == 0 && count) count[g] = (uint16_t)n;
    for (int i = lane; i < CZD_MAXMOVES; i += 64) {
        const size_t o = (size_t)g * CZD_MAXMOVES + i;
        const bool ok = i < n;
        if (label) label[o] = ok ? v.move[cb + i] : (uint16_t)0xFFFF;
        if (N) N[o] = ok ? v.N[cb + i] : 0;
        if (Q) Q[o] = ok ? v.Q[cb + i] : 0.f;
        if (P) P[o] = ok ? v.P[cb + i] : 0.f;
        if (W) W[o] = ok ? v.W[cb + i] : 0.f;
    }
}

// ---- K7: update_tree (main.py:272-276) with in-place subtree compaction -------------------------------
// The played child's subtree is the only part of the tree update_tree keeps (the reference drops the rest with
// `self.root.parent = None`).  Nodes are allocated in expansion order, so a child always has a larger index than its
// parent and a sibling group is contiguous: a STABLE compaction (kept nodes keep their relative order) therefore
// (a) keeps every sibling group contiguous and in generation order — selection is unchanged —, (b) moves every node
// to an index <= its old one, so it can be done in place, ascending, with no spare pool (round 1 kept a second pool
// of the same size as the compaction target: 2 x 59 GB at 8192 trees x 256 320 nodes).
//   pass 1  kept[i] = (i == played child) || kept[parent[i]], ascending in chunks of 256 nodes; a parent inside the
//           same chunk is resolved by iterating over the chunk (parent < child bounds the iterations by the chunk's
//           chain depth); one 64-bit word of the bitmap per wave and chunk (ballot)
//   rank    exclusive prefix count of the bitmap words: new index of node i = rank[i / 64] + popc(bits below i)
//   pass 2  ascending: a chunk is read into registers, barrier, written to the new indices with parent /
//           child_begin remapped through the rank
// Board / side / restrict_round follow selfplay's bookkeeping, main.py:1522-1528.  The tree's POOL_EXHAUSTED status
// is cleared: the compaction has made room again (a tree that stays full is flagged again by the next expansion).
__device__ __forceinline__ bool mark_tst(const unsigned long long *bits, int i) { return (bits[i >> 6] >> (i & 63)) & 1ull; }
__device__ __forceinline__ int mark_rank_of(const unsigned long long *bits, const uint32_t *rank, int i) {
    const unsigned long long below = (i & 63) ? (bits[i >> 6] & ((1ull << (i & 63)) - 1ull)) : 0ull;
    return (int)rank[i >> 6] + __popcll(below);
}

__global__ __launch_bounds__(256) void k_advance_global(CzTrees t, CzTables tab, int G, const uint16_t *__restrict__ played) {
    __shared__ int s_found, s_total;
    __shared__ int s_flag[256];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (g >= G) return;
    const uint16_t l = played[g];
    if (l >= CZ_NLABELS) return;
    const TreeView v = view_of(t, g);
    unsigned long long *bits = t.mark_bits + (size_t)g * t.words;
    uint32_t *rank = t.mark_rank + (size_t)g * t.words;
    const int root = t.root_node[g];
    const int n = t.n_nodes[g];
    // find the played child (children are unique; at most 128 of them)
    const int cb = v.child_begin[root];
    const int cc = cb < 0 ? 0 : v.child_count[root];
    if (tid == 0) s_found = -1;
    __syncthreads();
    if (tid < cc && v.move[cb + tid] == l) s_found = cb + tid;
    // board bookkeeping
    uint8_t *rb = t.root_board + (size_t)g * CZD_BOARD_LDS;
    if (tid == 0) {
        const int src = tab.srcdst[l] & 0xFF, dst = tab.srcdst[l] >> 8;
        const int cap = rb[dst];
        rb[dst] = rb[src]; rb[src] = 0;
        t.root_side[g] ^= 1;
        t.root_rr[g] = cap ? 0 : t.root_rr[g] + 1;
        t.root_ply[g] = (uint16_t)min((int)t.root_ply[g] + 1, 65535);
        t.sims[g] = 0;
    }
    __syncthreads();
    const int found = s_found;
    if (found < 0) {
        if (tid == 0) {
            t.status[g] = (t.status[g] & ~CZ_ST_POOL_EXHAUSTED) | CZ_ST_BAD_ADVANCE;
            init_root(v, 0);
            t.root_node[g] = 0; t.n_nodes[g] = 1;
        }
        ec_clear_tree(t, g, tid, 256);
        return;
    }
    // ---- pass 1: the kept-node bitmap.  Nothing below `found` can be in its subtree.
    const int w0 = found >> 6, W = (n + 63) >> 6;
    for (int w = tid; w < w0; w += 256) bits[w] = 0ull;
    __syncthreads();
    for (int base = w0 << 6; base < n; base += 256) {
        const int i = base + tid;
        int st, p = -1;   // st: 0 unknown, 1 kept, 2 dropped
        if (i >= n || i < found) st = 2;
        else if (i == found) st = 1;
        else {
            p = v.parent[i];
            if (p < found) st = 2;
            else if (p == found) st = 1;
            else if (p < base) st = mark_tst(bits, p) ? 1 : 2;
            else st = 0;
        }
        s_flag[tid] = st;
        __syncthreads();
        for (;;) {
            int open = 0;
            if (st == 0) { const int ps = s_flag[p - base]; if (ps) st = ps; else open = 1; }
            open = __syncthreads_or(open);
            s_flag[tid] = st;
            __syncthreads();
            if (!open) break;
        }
        const unsigned long long m = __ballot(st == 1);
        if (lane == 0 && (base >> 6) + wave < W) bits[(base >> 6) + wave] = m;
        __syncthreads();
    }
    // ---- rank: exclusive prefix count over the bitmap words (W <= cap / 64)
    {
        const int per = (W + 255) / 256;
        const int lo = tid * per, hi = min(W, lo + per);
        int c = 0;
        for (int w = lo; w < hi; ++w) c += __popcll(bits[w]);
        s_flag[tid] = c;
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int k = 0; k < 256; ++k) { const int x = s_flag[k]; s_flag[k] = acc; acc += x; }
            s_total = acc;
        }
        __syncthreads();
        int acc = s_flag[tid];
        for (int w = lo; w < hi; ++w) { rank[w] = (uint32_t)acc; acc += __popcll(bits[w]); }
        __syncthreads();
    }
    // ---- evaluation cache: entries of kept nodes follow them to their new indices, the others are forgotten
    if (t.ec_key) {
        unsigned long long *ek = t.ec_key + (size_t)g * CZ_EC_ENTRIES;
        int32_t *en = t.ec_node + (size_t)g * CZ_EC_ENTRIES;
        for (int e = tid; e < CZ_EC_ENTRIES; e += 256) {
            if (ek[e] == 0ull) continue;
            const int nd = en[e];
            if (nd >= found && nd < n && mark_tst(bits, nd)) en[e] = (int)mark_rank_of(bits, rank, nd);
            else ek[e] = 0ull;
        }
    }
    // ---- pass 2: move the kept nodes down, ascending (new index <= old index)
    for (int base = w0 << 6; base < n; base += 256) {
        const int i = base + tid;
        const bool keep = i < n && mark_tst(bits, i);
        float nP = 0.f, nW = 0.f, nQ = 0.f;
        int nN = 0, np = -1, ncb = -1;
        uint16_t ncc = 0, nmv = 0, nsd = 0;
        if (keep) {
            nP = v.P[i]; nW = v.W[i]; nQ = v.Q[i]; nN = v.N[i]; np = v.parent[i]; ncb = v.child_begin[i];
            ncc = v.child_count[i]; nmv = v.move[i]; nsd = v.sd[i];
        }
        __syncthreads();   // the whole chunk is in registers before any of its slots is overwritten
        if (keep) {
            const int o = mark_rank_of(bits, rank, i);
            v.P[o] = nP; v.W[o] = nW; v.Q[o] = nQ; v.N[o] = nN;
            v.parent[o] = i == found ? -1 : mark_rank_of(bits, rank, np);
            v.child_begin[o] = ncb >= 0 ? mark_rank_of(bits, rank, ncb) : -1;
            v.child_count[o] = ncc; v.move[o] = nmv; v.sd[o] = nsd;
        }
    }
    if (tid == 0) {
        t.root_node[g] = 0; t.n_nodes[g] = s_total;
        t.status[g] &= ~CZ_ST_POOL_EXHAUSTED;
    }
}

// The same compaction with the bitmap and its ranks in LDS, 1024 nodes per chunk (16 waves) and the next chunk's loads in
// flight while the current one is resolved: a tree of a 1600-playout search has ~62 k nodes — 245 strictly sequential
// 256-node chunks of the kernel above, each paying two or three dependent global round trips (parent, bitmap word, rank),
// ~0.6 ms for the handful of trees that move at one check of an asynchronous loop, i.e. ~3 % of the run.  Here a chunk costs
// LDS round trips only (61 chunks, parent / node fields prefetched one chunk ahead).  Used when the tree's bitmap fits the
// workgroup's dynamic LDS (12 bytes per 64 nodes); the global-memory variant above remains for larger pools.
#define CZ_ADV_T 1024
__device__ __forceinline__ bool lmark_tst(const unsigned long long *bits, int i) { return (bits[i >> 6] >> (i & 63)) & 1ull; }
__device__ __forceinline__ int lmark_rank_of(const unsigned long long *bits, const uint32_t *rank, int i) {
    const unsigned long long below = (i & 63) ? (bits[i >> 6] & ((1ull << (i & 63)) - 1ull)) : 0ull;
    return (int)rank[i >> 6] + __popcll(below);
}

// the trees that move at this call (a handful of thousands in an asynchronous loop): compacted into a list so that the
// 1024-thread workgroups below are launched for them only
__global__ __launch_bounds__(256) void k_advance_list(CzTrees t, int G, const uint16_t *__restrict__ played) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < G && played[g] < CZ_NLABELS) t.adv_list[atomicAdd(t.adv_cnt, 1)] = g;
}

__device__ __forceinline__ void advance_tree_lds(const CzTrees &t, const CzTables &tab, int g, unsigned long long *adv_lds,
                                                 const uint16_t *__restrict__ played) {
    __shared__ int s_found, s_total;
    __shared__ int s_flag[CZ_ADV_T];
    __shared__ int s_wave[CZ_ADV_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint16_t l = played[g];
    const TreeView v = view_of(t, g);
    unsigned long long *bits = adv_lds;                          // [t.words]
    uint32_t *rank = (uint32_t *)(adv_lds + t.words);            // [t.words]
    const int root = t.root_node[g];
    const int n = t.n_nodes[g];
    const int cb = v.child_begin[root];
    const int cc = cb < 0 ? 0 : v.child_count[root];
    if (tid == 0) s_found = -1;
    __syncthreads();
    if (tid < cc && v.move[cb + tid] == l) s_found = cb + tid;
    uint8_t *rb = t.root_board + (size_t)g * CZD_BOARD_LDS;
    if (tid == 0) {   // board bookkeeping (selfplay, main.py:1522-1528)
        const int src = tab.srcdst[l] & 0xFF, dst = tab.srcdst[l] >> 8;
        const int cap = rb[dst];
        rb[dst] = rb[src]; rb[src] = 0;
        t.root_side[g] ^= 1;
        t.root_rr[g] = cap ? 0 : t.root_rr[g] + 1;
        t.root_ply[g] = (uint16_t)min((int)t.root_ply[g] + 1, 65535);
        t.sims[g] = 0;
    }
    __syncthreads();
    const int found = s_found;
    if (found < 0) {
        if (tid == 0) {
            t.status[g] = (t.status[g] & ~CZ_ST_POOL_EXHAUSTED) | CZ_ST_BAD_ADVANCE;
            init_root(v, 0);
            t.root_node[g] = 0; t.n_nodes[g] = 1;
        }
        ec_clear_tree(t, g, tid, CZ_ADV_T);
        return;
    }
    // ---- pass 1: the kept-node bitmap (nothing below `found` can be in its subtree)
    const int w0 = found >> 6, W = (n + 63) >> 6;
    for (int w = tid; w < w0; w += CZ_ADV_T) bits[w] = 0ull;
    const int first = w0 << 6;
    int pn = (first + tid < n && first + tid > found) ? v.parent[first + tid] : -1;
    __syncthreads();
    for (int base = first; base < n; base += CZ_ADV_T) {
        const int i = base + tid;
        const int p = pn;
        const int in = i + CZ_ADV_T;
        pn = (in < n) ? v.parent[in] : -1;      // the next chunk's parents are in flight while this chunk is resolved
        int st;   // 0 unknown, 1 kept, 2 dropped
        if (i >= n || i < found) st = 2;
        else if (i == found) st = 1;
        else if (p < found) st = 2;
        else if (p == found) st = 1;
        else if (p < base) st = lmark_tst(bits, p) ? 1 : 2;
        else st = 0;
        s_flag[tid] = st;
        __syncthreads();
        for (;;) {   // parents inside the chunk: iterate (parent < child bounds the rounds by the chunk's chain depth)
            int open = 0;
            if (st == 0) { const int ps = s_flag[p - base]; if (ps) st = ps; else open = 1; }
            open = __syncthreads_or(open);
            s_flag[tid] = st;
            __syncthreads();
            if (!open) break;
        }
        const unsigned long long m = __ballot(st == 1);
        if (lane == 0 && (base >> 6) + wave < W) bits[(base >> 6) + wave] = m;
        __syncthreads();
    }
    // ---- rank: exclusive prefix count over the bitmap words
    {
        const int per = (W + CZ_ADV_T - 1) / CZ_ADV_T;
        const int lo = tid * per, hi = min(W, lo + per);
        int c = 0;
        for (int w = lo; w < hi; ++w) c += __popcll(bits[w]);
        int x = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        if (lane == 63) s_wave[wave] = x;
        __syncthreads();
        if (wave == 0) {
            int tv = lane < CZ_ADV_T / 64 ? s_wave[lane] : 0, ti = tv;
#pragma unroll
            for (int d = 1; d < CZ_ADV_T / 64; d <<= 1) { const int y = __shfl_up(ti, d, 64); if (lane >= d) ti += y; }
            if (lane < CZ_ADV_T / 64) s_wave[lane] = ti - tv;
            if (lane == CZ_ADV_T / 64 - 1) s_total = ti;
        }
        __syncthreads();
        int acc = x - c + s_wave[wave];
        for (int w = lo; w < hi; ++w) { rank[w] = (uint32_t)acc; acc += __popcll(bits[w]); }
        __syncthreads();
    }
    // ---- evaluation cache: entries of kept nodes follow them to their new indices, the others are forgotten
    if (t.ec_key) {
        unsigned long long *ek = t.ec_key + (size_t)g * CZ_EC_ENTRIES;
        int32_t *en = t.ec_node + (size_t)g * CZ_EC_ENTRIES;
        for (int e = tid; e < CZ_EC_ENTRIES; e += CZ_ADV_T) {
            if (ek[e] == 0ull) continue;
            const int nd = en[e];
            if (nd >= found && nd < n && lmark_tst(bits, nd)) en[e] = lmark_rank_of(bits, rank, nd);
            else ek[e] = 0ull;
        }
    }
    // ---- pass 2: move the kept nodes down, ascending (new index <= old index: a chunk's stores never reach the next chunk,
    // whose loads are therefore issued before this chunk's stores)
    float nP = 0.f, nW = 0.f, nQ = 0.f;
    int nN = 0, np = -1, ncb = -1;
    uint16_t ncc = 0, nmv = 0, nsd = 0;
    bool nkeep = first + tid < n && lmark_tst(bits, first + tid);
    if (nkeep) {
        const int i = first + tid;
        nP = v.P[i]; nW = v.W[i]; nQ = v.Q[i]; nN = v.N[i]; np = v.parent[i]; ncb = v.child_begin[i]; ncc = v.child_count[i]; nmv = v.move[i]; nsd = v.sd[i];
    }
    for (int base = first; base < n; base += CZ_ADV_T) {
        const int i = base + tid;
        const bool keep = nkeep;
        const float cP = nP, cW = nW, cQ = nQ;
        const int cN = nN, cp = np, ccb = ncb;
        const uint16_t ccc = ncc, cmv = nmv, csd = nsd;
        const int in = i + CZ_ADV_T;
        nkeep = in < n && lmark_tst(bits, in);
        if (nkeep) {
            nP = v.P[in]; nW = v.W[in]; nQ = v.Q[in]; nN = v.N[in]; np = v.parent[in]; ncb = v.child_begin[in]; ncc = v.child_count[in]; nmv = v.move[in]; nsd = v.sd[in];
        }
        __syncthreads();   // every thread holds its node of this chunk (loaded one iteration ago) before any slot of it is overwritten
        if (keep) {
            const int o = lmark_rank_of(bits, rank, i);
            v.P[o] = cP; v.W[o] = cW; v.Q[o] = cQ; v.N[o] = cN;
            v.parent[o] = i == found ? -1 : lmark_rank_of(bits, rank, cp);
            v.child_begin[o] = ccb >= 0 ? lmark_rank_of(bits, rank, ccb) : -1;
            v.child_count[o] = ccc; v.move[o] = cmv; v.sd[o] = csd;
        }
    }
    if (tid == 0) {
        t.root_node[g] = 0; t.n_nodes[g] = s_total;
        t.status[g] &= ~CZ_ST_POOL_EXHAUSTED;
    }
}

__global__ __launch_bounds__(CZ_ADV_T) void k_advance_lds(CzTrees t, CzTables tab, const uint16_t *__restrict__ played) {
    extern __shared__ unsigned long long adv_lds[];
    const int cnt = *t.adv_cnt;
    for (int e = blockIdx.x; e < cnt; e += gridDim.x) {
        advance_tree_lds(t, tab, t.adv_list[e], adv_lds, played);
        __syncthreads();   // the LDS bitmap and flags are reused by the next tree
    }
}

__global__ void k_root_state(CzTrees t, int G, uint8_t *__restrict__ boards, uint8_t *__restrict__ side, int32_t *__restrict__ rr) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (g >= G) return;
    if (boards) for (int i = lane; i < CZ_NSQ; i += 64) boards[(size_t)g * CZ_NSQ + i] = t.root_board[(size_t)g * CZD_BOARD_LDS + i];
    if (lane == 0) { if (side) side[g] = t.root_side[g]; if (rr) rr[g] = t.root_rr[g]; }
}

__global__ void k_status(CzTrees t, int G, int32_t *__restrict__ status, int32_t *__restrict__ nodes, int32_t *__restrict__ sims,
                         int32_t *__restrict__ depth) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    if (status) status[g] = t.status[g];
    if (nodes) nodes[g] = t.n_nodes[g];
    if (sims) sims[g] = t.sims[g];
    if (depth) depth[g] = t.last_depth[g];
}

__global__ void k_clear_cache_stats(CzTrees t, int G) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G) { t.ec_hits[g] = 0u; t.ec_lookups[g] = 0u; t.ec_collisions[g] = 0u; }
}

}  // namespace

int czk_search_status(cz_ctx *c, int32_t *status, int32_t *nodes, int32_t *sims, int32_t *depth) {
    hipLaunchKernelGGL(k_status, dim3((c->G + 255) / 256), dim3(256), 0, c->stream, c->t, c->G, status, nodes, sims, depth);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_search_clear_cache_stats(cz_ctx *c) {
    hipLaunchKernelGGL(k_clear_cache_stats, dim3((c->max_games + 255) / 256), dim3(256), 0, c->stream, c->t, c->max_games);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_search_reset(cz_ctx *c, const uint8_t *boards, const uint8_t *side, const int32_t *rr, int G, const uint8_t *which) {
    hipLaunchKernelGGL(k_reset, dim3(G), dim3(64), 0, c->stream, c->t, boards, side, rr, G, which);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_search_select(cz_ctx *c, int mode, const uint8_t *active, void *planes, int dtype, int C, uint8_t *needs_eval, bool compact) {
    const uint16_t one16 = (uint16_t)(dtype == CZ_F16 ? 0x3C00 : 0x3F80);
    const int par = c->step_parity;
#define CZ_LAUNCH_SELECT(KERNEL, TT, ONE)                                                                               \
    hipLaunchKernelGGL((KERNEL<TT>), dim3(c->G), dim3(64), 0, c->stream, c->t, c->tab, c->G, mode, active, (TT *)planes, C, ONE, needs_eval, par, c->sim_target, c->terminal_extra)
    const bool cache = c->t.ec_key != nullptr, xc = cache && c->t.xc_base != nullptr;
    if (xc) {
        if (dtype == CZ_F32) { if (compact) CZ_LAUNCH_SELECT(k_select_compact_xcache, float, 1.0f); else CZ_LAUNCH_SELECT(k_select_xcache, float, 1.0f); }
        else { if (compact) CZ_LAUNCH_SELECT(k_select_compact_xcache, uint16_t, one16); else CZ_LAUNCH_SELECT(k_select_xcache, uint16_t, one16); }
    } else if (dtype == CZ_F32) {
        if (cache) { if (compact) CZ_LAUNCH_SELECT(k_select_compact_cache, float, 1.0f); else CZ_LAUNCH_SELECT(k_select_cache, float, 1.0f); }
        else { if (compact) CZ_LAUNCH_SELECT(k_select_compact, float, 1.0f); else CZ_LAUNCH_SELECT(k_select, float, 1.0f); }
    } else {
        if (cache) { if (compact) CZ_LAUNCH_SELECT(k_select_compact_cache, uint16_t, one16); else CZ_LAUNCH_SELECT(k_select_cache, uint16_t, one16); }
        else { if (compact) CZ_LAUNCH_SELECT(k_select_compact, uint16_t, one16); else CZ_LAUNCH_SELECT(k_select, uint16_t, one16); }
    }
#undef CZ_LAUNCH_SELECT
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_search_expand_backup(cz_ctx *c, const void *logits, const void *value, int dtype) {
    if (dtype == CZ_F32)
        hipLaunchKernelGGL((k_expand_backup<float, false>), dim3(c->G), dim3(64), 0, c->stream, c->t, c->tab, c->G, (const float *)logits,
                           (const float *)value, (const float *)nullptr, (const float *)nullptr, -1);
    else
        hipLaunchKernelGGL((k_expand_backup<uint16_t, false>), dim3(c->G), dim3(64), 0, c->stream, c->t, c->tab, c->G, (const uint16_t *)logits,
                           (const uint16_t *)value, (const float *)nullptr, (const float *)nullptr, -1);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_search_expand_backup_fc(cz_ctx *c, const float *z, const float *value, const float *fcw, const float *fcb, bool compact) {
    hipLaunchKernelGGL((k_expand_backup<float, true>), dim3(c->G), dim3(64), 0, c->stream, c->t, c->tab, c->G, z, value, fcw, fcb,
                       compact ? c->step_parity : -1);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_search_root_stats(cz_ctx *c, uint16_t *label, int32_t *N, float *Q, float *P, float *W, uint16_t *count) {
    hipLaunchKernelGGL(k_root_stats, dim3(c->G), dim3(64), 0, c->stream, c->t, c->G, label, N, Q, P, W, count);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_search_pick_ready(cz_ctx *c, int32_t *thr, int next_thr, uint16_t *played, uint8_t *ready, unsigned long long *banked) {
    hipLaunchKernelGGL(k_pick_ready, dim3(c->G), dim3(64), 0, c->stream, c->t, c->G, thr, next_thr, played, ready, banked);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_search_reload_finished(cz_ctx *c, const uint8_t *ready, const uint16_t *played, const uint8_t *boards, const uint8_t *side,
                               const int32_t *rr, unsigned long long *reloaded) {
    hipLaunchKernelGGL(k_reload_finished, dim3(c->G), dim3(64), 0, c->stream, c->t, c->G, ready, played, boards, side, rr, reloaded);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_search_advance(cz_ctx *c, const uint16_t *played) {
    // bitmap + ranks of one tree in LDS: 12 bytes per 64 nodes (38 KB for the bench's 205 056-node pools)
    const size_t lds = (size_t)c->t.words * 12;
    const size_t lds_max = 150 * 1024;   // 160 KB per CU minus the kernel's static LDS; the attribute is per function, so always the maximum
    if (lds <= lds_max && !c->adv_force_global) {
        if (!c->adv_attr_set) {
            CZ_HIP(hipFuncSetAttribute((const void *)k_advance_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
            c->adv_attr_set = true;
        }
        CZ_HIP(hipMemsetAsync(c->t.adv_cnt, 0, sizeof(int32_t), c->stream));
        hipLaunchKernelGGL(k_advance_list, dim3((c->G + 255) / 256), dim3(256), 0, c->stream, c->t, c->G, played);
        hipLaunchKernelGGL(k_advance_lds, dim3(c->G < 512 ? c->G : 512), dim3(CZ_ADV_T), lds, c->stream, c->t, c->tab, played);
    } else {
        hipLaunchKernelGGL(k_advance_global, dim3(c->G), dim3(256), 0, c->stream, c->t, c->tab, c->G, played);
    }
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int cz_search_root_state(cz_ctx *c, uint8_t *boards, uint8_t *side, int32_t *rr) {
    if (!c) { cz_set_error("null ctx"); return CZ_EINVAL; }
    if (c->G == 0) return CZ_OK;
    hipLaunchKernelGGL(k_root_state, dim3(c->G), dim3(64), 0, c->stream, c->t, c->G, boards, side, rr);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_search_select_k(cz_ctx *c, int mode, int K, const uint8_t *active, void *planes, int dtype, int C, uint8_t *needs_eval) {
    if (dtype == CZ_F32)
        hipLaunchKernelGGL(k_select_k<float>, dim3(c->G), dim3(64), 0, c->stream, c->t, c->tab, c->G, mode, K, c->sim_target, active, (float *)planes, C, 1.0f, needs_eval);
    else
        hipLaunchKernelGGL(k_select_k<uint16_t>, dim3(c->G), dim3(64), 0, c->stream, c->t, c->tab, c->G, mode, K, c->sim_target, active, (uint16_t *)planes, C, (uint16_t)(dtype == CZ_F16 ? 0x3C00 : 0x3F80), needs_eval);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}

int czk_search_expand_backup_k(cz_ctx *c, int K, const void *logits, const void *value, int dtype) {
    if (dtype == CZ_F32)
        hipLaunchKernelGGL(k_expand_backup_k<float>, dim3(c->G), dim3(64), 0, c->stream, c->t, c->tab, c->G, K, (const float *)logits, (const float *)value);
    else
        hipLaunchKernelGGL(k_expand_backup_k<uint16_t>, dim3(c->G), dim3(64), 0, c->stream, c->t, c->tab, c->G, K, (const uint16_t *)logits, (const uint16_t *)value);
    CZ_HIP(hipGetLastError());
    return CZ_OK;
}
