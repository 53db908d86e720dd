// cz_internal.h — context layout and helpers shared by the libcchess_hip translation units.
#pragma once
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

// ---- tree storage: structure-of-arrays, one fixed-capacity pool per tree -----------------------
// A node is the edge into it plus its expansion record (the reference's leaf_node, main.py:93-103,
// minus the eagerly materialised state string, quirk Q8).  Children of a node are contiguous, so a
// wave scores up to 128 siblings with three coalesced loads (Q, P, N).
struct CzPool {
    float *P, *W, *Q;
    int32_t *N, *parent, *child_begin;
    uint16_t *child_count, *move;
};

struct CzTrees {
    CzPool pool[2];      // [max_games * cap] each; pool[cur] is live, the other is the compaction target
    int cap;
    // per tree [max_games]
    int32_t *cur;        // which pool holds tree g
    uint8_t *root_board; // [max_games][96]
    uint8_t *root_side;
    int32_t *root_rr, *root_node, *n_nodes, *status, *sims, *last_depth;
    // pending leaf between select and expand_backup
    int32_t *pend_kind, *pend_leaf;
    float *pend_value;
    uint8_t *pend_side;
    uint16_t *pend_nmoves, *pend_moves;  // [max_games][128]
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
    void *pool_block[2];
    bool conv_attr_set, tower_attr_set;  // dynamic-LDS opt-in of the MFMA kernels done for this device
    int width;         // simulations in flight per tree the pending arrays are sized for (cz_search_set_width)
    void *pend_block;  // separate allocation of the pending arrays when width > 1
    int step_parity;   // which evcnt entry the current compact step uses
    const int32_t *batch_count;  // cz_set_batch_count: device row count bounding the net launches, or NULL
};

// kernels' launch wrappers (cz_rules.hip / cz_search.hip)
int czk_movegen(cz_ctx *, const uint8_t *, const uint8_t *, int, uint16_t *, uint16_t *, uint32_t *);
int czk_apply_move(cz_ctx *, uint8_t *, uint8_t *, const uint16_t *, int, uint64_t *, uint8_t *, int8_t *);
int czk_hash(cz_ctx *, const uint8_t *, const uint8_t *, int, uint64_t *);
int czk_encode_planes(cz_ctx *, const uint8_t *, const uint8_t *, int, void *, int, int, int);
int czk_search_reset(cz_ctx *, const uint8_t *, const uint8_t *, const int32_t *, int);
int czk_search_select(cz_ctx *, int, const uint8_t *, void *, int, int, uint8_t *, bool compact = false);
int czk_search_expand_backup(cz_ctx *, const void *, const void *, int);
int czk_search_expand_backup_fc(cz_ctx *, const float *, const float *, const float *, const float *, bool compact);
int czk_search_root_stats(cz_ctx *, uint16_t *, int32_t *, float *, float *, float *, uint16_t *);
int czk_search_advance(cz_ctx *, const uint16_t *);
int czk_search_select_k(cz_ctx *, int, int, const uint8_t *, void *, int, int, uint8_t *);
int czk_search_expand_backup_k(cz_ctx *, int, const void *, const void *, int);
