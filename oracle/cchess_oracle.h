/*
 * cchess_oracle.h — CPU restatement of the cchess-zero hot path (TEST INFRASTRUCTURE).
 *
 * This library is the parity checker for the HIP path.  It is NOT product code:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  Every function cites the reference file:line it restates
 * (paths relative to chengstone/cchess-zero).
 *
 * Pinning: the restatement is checked against golden vectors generated from the
 * UNMODIFIED reference Python (tests/golden/gen_golden.py, run in the build
 * container where /root/reference exists; fixtures committed under tests/golden/).
 *
 * Board encoding shared with the HIP library (include/cchess_hip.h):
 *   board[90], index sq = y*9 + x, y = rank 0..9 (rank 0 = first FEN row = red/
 *   upper-case/'w' home, main.py:585), x = file 0..8 ('a'..'i').
 *   piece code = 1 + index in pieces_order 'KARBNPCkarbnpc' (main.py:208); 0 = empty.
 *   side: 0 = 'w' (red, upper case), 1 = 'b' (black, lower case).
 */
#ifndef CCHESS_ORACLE_H
#define CCHESS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CZO_NLABELS 2086
#define CZO_MAXMOVES 128
#define CZO_NSQ 90
#define CZO_PLANE_ELEMS (9 * 10 * 14)

/* ---- tables: main.py:23-27 (flipped_uci_labels), :30-65 (create_uci_labels), :211-217 ---- */
/* labels: CZO_NLABELS * 5 bytes, NUL-terminated 4-char strings. */
const char *czo_labels(void);
/* lut[src_sq*90+dst_sq] = label index or -1 (label2i, main.py:217). */
const int16_t *czo_lut(void);
/* unflipped_index (main.py:214). */
const int16_t *czo_unflip(void);
/* label -> src_sq | dst_sq<<8 */
const uint16_t *czo_label_srcdst(void);

/* ---- state string <-> board (main.py:585, :705-714, :691-699) ---- */
int czo_fen_to_board(const char *fen, uint8_t *board90);
/* out must hold >= 100 bytes. Returns length. */
int czo_board_to_fen(const uint8_t *board90, char *out);

/* ---- rules ---- */
/* GameBoard.get_legal_moves (main.py:743-1109): ordered label list. Returns count,
 * or -1 if a generated move has no label / more than CZO_MAXMOVES moves. */
int czo_legal_moves(const uint8_t *board90, int side, uint16_t *labels_out);
/* GameBoard.sim_do_action (main.py:647-702) + is_kill_move (:226) + king test (:409-413).
 * Moves in place. *captured = code of captured piece (0 none).
 * returns terminal flags: bit0 = 'K' missing, bit1 = 'k' missing (after the move). */
int czo_apply_move(uint8_t *board90, uint16_t label, uint8_t *captured);
/* MCTS_tree.generate_inputs (main.py:531-533) = try_flip (:560-574) + state_to_positions
 * (:547-557) with quirk Q1 (9-stride read into a 10-wide plane) when quirk_q1 != 0.
 * planes: float[9][10][14]. */
void czo_encode_planes(const uint8_t *board90, int side, int quirk_q1, float *planes);
/* 64-bit Zobrist (defined by this project; the reference has no hash — SURVEY §8c). */
uint64_t czo_zobrist_key(int code, int sq); /* code 1..14 */
uint64_t czo_zobrist_side(void);
uint64_t czo_hash(const uint8_t *board90, int side);

/* ---- search: leaf_node (main.py:93-206), MCTS_tree (:234-529), search_threads = 1 semantics ---- */
typedef struct czo_search czo_search;

czo_search *czo_search_create(int max_games, int max_nodes_per_tree);
void czo_search_destroy(czo_search *);
/* MCTS_tree.__init__/reload (main.py:235-259): fresh unexpanded roots. rr = restrict_round. */
int czo_search_reset(czo_search *, const uint8_t *boards, const uint8_t *side,
                     const int32_t *restrict_round, int G);
/* One lock-step selection. mode 0 = root expansion only (main.py:475-487): trees whose
 * root is unexpanded present the root as the leaf, others idle.  mode 1 = one simulation
 * (start_tree_search, main.py:350-435) per tree: walk by select_new (:158) to an
 * unexpanded node or a terminal/draw child.  planes [G][9][10][14] (generate_inputs of the
 * leaf, zeros if no eval needed); needs_eval[G]. */
int czo_search_select(czo_search *, int mode, float *planes, uint8_t *needs_eval);
/* expand (main.py:175-187, flip_policy :1153) + back_up_value (:189-194) along the unwind
 * (:426-435).  logits [G][2086], value [G]. */
int czo_search_expand_backup(czo_search *, const float *logits, const float *value);
/* root.child.items() (main.py:1339): per tree up to 128 (label, N, Q, P, W). Arrays are
 * [G][128]; count [G]. Any output may be NULL. */
int czo_search_root_stats(const czo_search *, uint16_t *label, int32_t *N, float *Q,
                          float *P, float *W, uint16_t *count);
/* update_tree (main.py:272-276) + the game-side bookkeeping of selfplay (:1522-1528):
 * re-root onto the played child, keep the subtree, advance board/side/restrict_round. */
int czo_search_advance(czo_search *, const uint16_t *played_label);
/* status[G]: 0 ok, bit0 node pool exhausted, bit1 node with zero legal moves selected
 * (reference raises, Q7), bit2 >128 moves / unlabeled move. nodes_used[G], sims[G]. */
int czo_search_status(const czo_search *, int32_t *status, int32_t *nodes_used, int32_t *sims);
/* current root position of every tree */
int czo_search_root_state(const czo_search *, uint8_t *boards, uint8_t *side, int32_t *rr);
/* debugging/parity: depth (edges from root) of the last selected leaf per tree */
int czo_search_last_depth(const czo_search *, int32_t *depth);

/* k simulations in flight per tree with physical virtual loss: restates the schedule of the HIP kernels
 * k_select_k / k_expand_backup_k (see cchess_oracle.c).  planes [G*K][9][10][14], needs_eval [G*K],
 * logits [G*K][2086], value [G*K]. */
int czo_search_select_k(czo_search *, int mode, int K, float *planes, uint8_t *needs_eval);
/* budget of completed + in-flight simulations per tree for the k > 1 schedule (0 = none): a tree issues no descent
 * beyond it, so a search ends with exactly `playouts` simulations like MCTS_tree.main (main.py:489-493) */
int czo_search_set_sim_target(czo_search *, int target);
int czo_search_expand_backup_k(czo_search *, int K, const float *logits, const float *value);

/* pre-order dump of tree g: records of 7 int32 {depth, label, N, bits(W), bits(Q), bits(P),
 * child_count or -1}; returns the record count (writes at most max_records). */
int czo_search_tree_dump(const czo_search *, int g, int32_t *out, int max_records);

#ifdef __cplusplus
}
#endif
#endif
