// cz_internal.h — context layout and helpers shared by the libcchess_hip translation units.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/cchess_hip.h"
#include "cz_device.h"

// ---- error plumbing -------------------------------------------------------------------------
void cz_set_error(const char *fmt, ...);
#define CZ_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            cz_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return CZ_EHIP;                                                                       \
        }                                                                                         \
    } while (0)
#define CZ_REQUIRE(cond, msg)                                \
    do {                                                     \
        if (!(cond)) { cz_set_error("%s", msg); return CZ_EINVAL; } \
    } while (0)

// ---- host tables (cz_tables.hip) --------------------------------------------------------------
struct CzHostTables {
    char labels[CZ_NLABELS * 5];
    int16_t lut[CZ_NSQ * CZ_NSQ];
    int16_t unflip[CZ_NLABELS];
    uint16_t srcdst[CZ_NLABELS];
    uint64_t zob[15 * CZ_NSQ + 1];  // [15*90] = side key
};
const CzHostTables &cz_host_tables();

#define CZ_PATH_MAX 32   // deeper paths fall back to the parent walk

// ---- tree storage: structure-of-arrays, one fixed-capacity pool per tree -----------------------
// A node is the edge into it plus its expansion record (the reference's leaf_node, main.py:93-103,
// minus the eagerly materialised state string, quirk Q8).  Children of a node are contiguous, so a
// wave scores up to 128 siblings with three coalesced loads (Q, P, N).  Nodes are allocated in expansion
// order, so every child has a larger index than its parent — cz_search_advance relies on that to compact
// the kept subtree IN PLACE (one pool per tree, no spare pool).
struct CzPool {
    float *P, *W, *Q;
    int32_t *N, *parent, *child_begin;
    uint16_t *child_count, *move;
    uint16_t *sd;   // src | dst << 8 of `move` (k_select applies the move without the label -> (src, dst) table round trip)
};

// cz_selfplay_*: per game slot, the (s, pi, z) records of the game in progress (cchess_main.selfplay keeps
// states / mcts_probs / current_players lists, main.py:1496-1518) and the counters of finished games
struct CzSelfplay {
    int max_plies;               // history capacity per game; a game reaching it is adjudicated a draw
    uint8_t *hist;               // [max_games][max_plies][CZ_REC_BYTES]
    int32_t *ply;                // [max_games] plies recorded for the game in progress
    uint8_t *stalled;            // [max_games] the last choose found no root child (node pool exhausted at the root)
    uint8_t *active;             // [max_games] 0 = parked (finished, not re-seeded)
    uint8_t *start_board;        // [max_games][96] position every new game of the slot starts from
    uint8_t *start_side;         // [max_games]
    int32_t *start_rr;           // [max_games]
    long long *stats;            // [CZ_SP_NSTATS]
};

#define CZ_EC_BUCKETS 128
#define CZ_EC_ENTRIES (CZ_EC_BUCKETS * 64)
#ifndef CZ_EC_BUDGET
#define CZ_EC_BUDGET 4   // evaluation-cache hits a tree may complete inside one select launch (its own budget, beside terminal_extra)
#endif

// Per-tree scalars: ONE 64-byte record per tree instead of sixteen arrays.  A wave owns a tree, so what it reads at entry
// (root, counters, status) and leaves behind (the pending leaf) is one cache line, and the kernels hold one base pointer
// instead of sixteen (the select kernels spilled 30-60 SGPRs keeping the array bases alive).  Kernel code keeps the array
// syntax — t.status[g] — through CzRecField: every field of the union below IS the record pointer plus a constant offset.
struct CzTreeRec {
    int32_t root_rr, root_node, n_nodes, status;      //  0
    int32_t sims;                                     // 16
    int16_t last_depth;                               // 20  depth of the last simulation (statistics)
    uint16_t root_ply;                                // 22  re-roots since the tree was (re)set: the root's ply in its game (cross-tree cache priority)
    int32_t pend_kind, pend_leaf;                     // 24  pending leaf between select and expand_backup (width 1)
    float pend_value;                                 // 32
    int32_t pend_depth;                               // 36  levels of the pending path (pend_path)
    uint32_t ec_hits, ec_lookups;                     // 40  evaluation cache statistics of the tree
    unsigned long long pend_key;                      // 48  Zobrist key of the pending leaf (evaluation cache)
    uint16_t pend_nmoves;                             // 56
    uint8_t root_side, pend_side;                     // 58
    uint32_t ec_collisions;                           // 60  key matches whose stored position differed (taken as misses)
};
static_assert(sizeof(CzTreeRec) == 64, "CzTreeRec is one 64-byte line per tree");

template <typename T, int OFF>
struct CzRecField {
    char *base;
    __host__ __device__ __forceinline__ T &operator[](int g) const { return *reinterpret_cast<T *>(base + (size_t)g * sizeof(CzTreeRec) + OFF); }
};
#define CZ_REC_FIELD(type, name) CzRecField<type, offsetof(CzTreeRec, name)> name

struct CzTrees {
    CzPool pool;         // [max_games * cap]
    int cap;
    int words;           // ceil(cap / 64): 64-node words of the advance bitmap
    unsigned long long *mark_bits;   // [max_games][words] k_advance: which nodes of the tree are kept
    uint32_t *mark_rank;             // [max_games][words] kept nodes before the word = new index of its first kept node
    int32_t *adv_list, *adv_cnt;     // [max_games], [1]  k_advance_list: the trees that move at this cz_search_advance
    uint8_t *root_board; // [max_games][96]
    union {              // [max_games] records; t.<field>[g] addresses rec[g].<field>
        CzTreeRec *rec;
        CZ_REC_FIELD(int32_t, root_rr); CZ_REC_FIELD(int32_t, root_node); CZ_REC_FIELD(int32_t, n_nodes);
        CZ_REC_FIELD(int32_t, status); CZ_REC_FIELD(int32_t, sims); CZ_REC_FIELD(int16_t, last_depth); CZ_REC_FIELD(uint16_t, root_ply);
        CZ_REC_FIELD(int32_t, pend_kind); CZ_REC_FIELD(int32_t, pend_leaf); CZ_REC_FIELD(float, pend_value);
        CZ_REC_FIELD(int32_t, pend_depth); CZ_REC_FIELD(uint32_t, ec_hits); CZ_REC_FIELD(uint32_t, ec_lookups); CZ_REC_FIELD(uint32_t, ec_collisions);
        CZ_REC_FIELD(unsigned long long, pend_key); CZ_REC_FIELD(uint16_t, pend_nmoves);
        CZ_REC_FIELD(uint8_t, root_side); CZ_REC_FIELD(uint8_t, pend_side);
    };
    uint16_t *pend_moves;             // [max_games * width][128] legal moves of the pending leaves
    // the selected path of the pending simulation (width 1): node index per level below the root, so that the backup
    // updates all levels in parallel instead of chasing parent pointers (one dependent round trip per level)
    int32_t *pend_path;               // [max_games][CZ_PATH_MAX]
    // pending leaves of the width > 1 kernels (k_select_k / k_expand_backup_k), slot = tree * width + j
    int32_t *pk_kind, *pk_leaf;       // [max_games * width]
    float *pk_value;
    uint8_t *pk_side;
    uint16_t *pk_nmoves;
    // evaluation cache (cz_search_set_eval_cache): per tree CZ_EC_BUCKETS buckets of 64 entries {Zobrist key of an expanded
    // node's position, its node index, the value its evaluation backed up}.  The priors are the node's children's P.
    unsigned long long *ec_key;       // [max_games][CZ_EC_ENTRIES]  0 = empty      (NULL: cache off)
    int32_t *ec_node;                 // [max_games][CZ_EC_ENTRIES]
    float *ec_val;                    // [max_games][CZ_EC_ENTRIES]
    uint32_t *ec_board;               // [max_games][CZ_EC_ENTRIES][12] the entry's position, packed (wave_pack_board): checked on every hit
    uint32_t *pend_board;             // [max_games][12] packed position of the pending leaf (select -> expand_backup)
    unsigned long long ec_key_mask;   // ~0; tests narrow it (cz_search_debug_eval_cache_key_bits) to force key collisions
    // cross-tree level of the evaluation cache (cz_search_set_xcache): ONE table per context, shared by all of its trees.  An
    // entry cannot lend node indices (the lender tree's nodes move at its next re-root), so it is self-contained: key, packed
    // position, the value the evaluation backed up, the move count, the <= 128 labels / (src, dst) pairs / priors.  Entries are
    // claimed by k_expand_backup with an atomic compare-and-swap on the key — an empty slot of the key's 64-entry bucket, or,
    // when the bucket is full (round 6), the entry whose position lies DEEPEST in its game (game ply = re-roots of the filing tree
    // + depth of the leaf, kept in the upper half of the move-count word) if the new position is shallower: the table converges
    // to the shallowest positions ever evaluated — the openings every restarted game walks through again — instead of whatever
    // arrived first.  k_select (a later launch: the kernel boundary publishes the payload) only reads, and verifies the stored
    // position on every hit.  Emptied by the host whenever the weights change.
    // ONE base pointer (the kernels are short of scalar registers): with n = (xc_mask + 1) * 64 entries the block holds
    //   keys u64 [n] | counters u64 [8] | value f32 [n] | move count u32 [n] | position u32 [n][12] | labels u16 [n][128] |
    //   (src, dst) u16 [n][128] | priors f32 [n][128]        (czx_* below)
    char *xc_base;                    // NULL: off
    uint32_t xc_mask;                 // buckets - 1 (a power of two); a bucket = 64 consecutive entries
    // compact evaluation batches (cz_search_select_compact): row of the step's leaf in planes / z / value, or -1
    int32_t *slot_of;                 // [max_games]
    int32_t *evcnt;                   // [2] rows handed out this step / next step (ping-pong, zeroed one step ahead)
    unsigned long long *evtotal;      // [2] rows evaluated, steps: running totals for the flop accounting
};

struct cz_ctx {
    int device;
    hipStream_t stream;
    int max_games, cap, G;
    CzTables tab;      // device tables
    void *tab_block;   // single allocation behind `tab`
    CzTrees t;
    void *tree_block;  // single allocation behind the per-tree arrays
    void *pool_block;
    bool adv_attr_set;   // dynamic-LDS opt-in of k_advance_lds done
    bool adv_force_global;   // cz_search_debug_advance_in_global_memory (tests): take the path of pools whose bitmap exceeds LDS
    bool conv_attr_set, tower_attr_set, split_attr_set, mx_attr_set;  // dynamic-LDS opt-in of the MFMA kernels done for this device
    bool mx2_attr_set;
    int mx_kernel;       // cz_net_trunk_mx: 0 = not chosen yet, 1 = k_trunk_mx_c128, 2 = k_trunk_mx2_c128 (default; CCHESS_MX_KERNEL=1 selects the former)
    void *mx_xbuf;       // k_trunk_mx2_c128's block-input scratch (98,304 B per workgroup), grown on demand
    size_t mx_xbuf_bytes;
    int width;         // simulations in flight per tree the pending arrays are sized for (cz_search_set_width)
    void *pend_block;  // separate allocation of the pending arrays when width > 1
    int terminal_extra;   // cz_search_set_terminal_extra: terminal simulations a tree may complete inside one select launch
    int sim_target;    // cz_search_set_sim_target: completed simulations per tree a k > 1 search stops at (0: no limit)
    int step_parity;   // which evcnt entry the current compact step uses
    const int32_t *batch_count;  // cz_set_batch_count: device row count bounding the net launches, or NULL
    void *xc_block;      // cz_search_set_xcache: the cross-tree table's allocation
    int xc_log2_entries;
    const struct CzmTables *mask_tab;  // cz_maskgen.h tables on the device (k_movegen_mask)
    unsigned long long *clock_probe;  // cz_set_clock_probe: [clock_probe_wgs][4] stamps written by the trunk kernels, or NULL
    int clock_probe_wgs, clock_probe_last_grid;
    CzSelfplay sp;     // cz_selfplay_begin
    void *sp_block;
    void *ec_block;    // cz_search_set_eval_cache
};

// cross-tree cache: the arrays inside CzTrees::xc_base
__host__ __device__ __forceinline__ size_t czx_n(const CzTrees &t) { return ((size_t)t.xc_mask + 1) * 64; }
__host__ __device__ __forceinline__ unsigned long long *czx_key(const CzTrees &t) { return reinterpret_cast<unsigned long long *>(t.xc_base); }
// statistics: PER TREE, [max_games][4] uint32 (hits, lookups, entries written, claims lost) behind the entries, updated by the
// tree's own wave without atomics — four counters of the context bumped with same-address atomics by every probe of every tree
// cost the select launch 95 us of its 167 (profiles/r04x_xcache_stats_atomics.txt); cz_search_xcache_stats sums them
__host__ __device__ __forceinline__ uint32_t *czx_tree_stats(const CzTrees &t) { return reinterpret_cast<uint32_t *>(t.xc_base + czx_n(t) * 1088 + 64); }
__host__ __device__ __forceinline__ float *czx_val(const CzTrees &t) { return reinterpret_cast<float *>(t.xc_base + czx_n(t) * 8 + 64); }
__host__ __device__ __forceinline__ uint32_t *czx_cnt(const CzTrees &t) { return reinterpret_cast<uint32_t *>(t.xc_base + czx_n(t) * 12 + 64); }
__host__ __device__ __forceinline__ uint32_t *czx_board(const CzTrees &t) { return reinterpret_cast<uint32_t *>(t.xc_base + czx_n(t) * 16 + 64); }
__host__ __device__ __forceinline__ uint16_t *czx_moves(const CzTrees &t) { return reinterpret_cast<uint16_t *>(t.xc_base + czx_n(t) * 64 + 64); }
__host__ __device__ __forceinline__ uint16_t *czx_sd(const CzTrees &t) { return reinterpret_cast<uint16_t *>(t.xc_base + czx_n(t) * 320 + 64); }
__host__ __device__ __forceinline__ float *czx_P(const CzTrees &t) { return reinterpret_cast<float *>(t.xc_base + czx_n(t) * 576 + 64); }

// ---- device helpers shared by cz_search.hip / cz_selfplay.hip -----------------------------------
struct TreeView {
    float *P, *W, *Q;
    int32_t *N, *parent, *child_begin;
    uint16_t *child_count, *move;
    uint16_t *sd;   // src | dst << 8 of `move` (k_select applies the move without the label -> (src, dst) table round trip)
};

__device__ __forceinline__ TreeView view_of(const CzTrees &t, int g) {
    const CzPool &p = t.pool;
    const size_t base = (size_t)g * (size_t)t.cap;
    TreeView v;
    v.P = p.P + base; v.W = p.W + base; v.Q = p.Q + base;
    v.N = p.N + base; v.parent = p.parent + base; v.child_begin = p.child_begin + base;
    v.child_count = p.child_count + base; v.move = p.move + base; v.sd = p.sd + base;
    return v;
}

__device__ __forceinline__ void init_root(TreeView v, int idx) {
    v.P[idx] = 1.0f;  // p_ = 0.75 + 0.25 * dirichlet([0.3]) == 1 (quirk Q4), main.py:238
    v.W[idx] = 0.f; v.Q[idx] = 0.f; v.N[idx] = 0; v.parent[idx] = -1; v.child_begin[idx] = -1;
    v.child_count[idx] = 0; v.move[idx] = 0xFFFF; v.sd[idx] = 0;
}

// evaluation cache: forget everything tree g knows (fresh root: reset / reload / re-seed / failed advance)
__device__ __forceinline__ void ec_clear_tree(const CzTrees &t, int g, int tid, int nthreads) {
    if (!t.ec_key) return;
    unsigned long long *k = t.ec_key + (size_t)g * CZ_EC_ENTRIES;
    for (int i = tid; i < CZ_EC_ENTRIES; i += nthreads) k[i] = 0ull;
}

// kernels' launch wrappers (cz_rules.hip / cz_search.hip)
int czk_movegen(cz_ctx *, const uint8_t *, const uint8_t *, int, uint16_t *, uint16_t *, uint32_t *, int flags);
int czk_apply_move(cz_ctx *, uint8_t *, uint8_t *, const uint16_t *, int, uint64_t *, uint8_t *, int8_t *);
int czk_hash(cz_ctx *, const uint8_t *, const uint8_t *, int, uint64_t *);
int czk_encode_planes(cz_ctx *, const uint8_t *, const uint8_t *, int, void *, int, int, int);
int czk_search_reset(cz_ctx *, const uint8_t *, const uint8_t *, const int32_t *, int, const uint8_t *which);
int czk_search_status(cz_ctx *, int32_t *, int32_t *, int32_t *, int32_t *);
int czk_search_clear_cache_stats(cz_ctx *);
int czk_search_select(cz_ctx *, int, const uint8_t *, void *, int, int, uint8_t *, bool compact = false);
int czk_search_expand_backup(cz_ctx *, const void *, const void *, int);
int czk_search_expand_backup_fc(cz_ctx *, const float *, const float *, const float *, const float *, bool compact);
int czk_search_root_stats(cz_ctx *, uint16_t *, int32_t *, float *, float *, float *, uint16_t *);
int czk_search_advance(cz_ctx *, const uint16_t *);
int czk_search_pick_ready(cz_ctx *, int32_t *, int, uint16_t *, uint8_t *, unsigned long long *);
int czk_search_reload_finished(cz_ctx *, const uint8_t *, const uint16_t *, const uint8_t *, const uint8_t *, const int32_t *, unsigned long long *);
int czk_search_select_k(cz_ctx *, int, int, const uint8_t *, void *, int, int, uint8_t *);
int czk_search_expand_backup_k(cz_ctx *, int, const void *, const void *, int);
int czk_selfplay_seed(cz_ctx *, const uint8_t *, const uint8_t *, const int32_t *);
int czk_selfplay_choose(cz_ctx *, const float *, const float *, const uint16_t *, double, float, int, uint16_t *);
int czk_selfplay_adjudicate(cz_ctx *, int, const uint16_t *, int32_t *);
int czk_selfplay_flush(cz_ctx *, const int32_t *, const long long *, uint8_t *, long long, const long long *);
