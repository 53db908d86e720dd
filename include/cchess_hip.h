/*
 * cchess_hip.h — C-ABI of libcchess_hip.so: the MI355X (gfx950) hot path of cchess-zero.
 *
 * The reference (chengstone/cchess-zero) is pure Python and has no FFI layer; its boundary
 * for this path is the class surface of main.py.  Each entry point below names the
 * reference interface it replaces (file:line, relative to the reference repo).  The Python
 * façade that keeps the reference's class/CLI names (main.py, policy_value_network.py at the
 * repo root of this project) binds exactly these symbols through ctypes — see INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, a negative CZ_E* code on failure;
 *     cz_last_error() returns a thread-local message for the last failure.
 *   - cz_ctx owns one HIP device, one stream and all device buffers of G game trees.  A ctx is
 *     not thread-safe; distinct ctxs are independent (one per GPU / process).
 *   - all array arguments are DEVICE pointers (e.g. torch tensors' data_ptr()); they are
 *     borrowed for the duration of the enqueued work, never freed by the library.  Work is
 *     enqueued on the ctx stream (cz_set_stream) and is asynchronous; cz_synchronize waits.
 *     cz_malloc/cz_upload/cz_download exist for callers without a tensor library.
 *
 * Data formats (shared with the CPU oracle, oracle/cchess_oracle.h)
 *   board   uint8[90], sq = y*9 + x; y = rank 0..9 (rank 0 = first row of the reference's
 *           state string = red / upper-case / 'w' home, main.py:585), x = file 'a'..'i'.
 *           code 0 = empty, 1..14 = 1 + index in pieces_order "KARBNPCkarbnpc" (main.py:208).
 *   side    uint8: 0 = 'w' (red, upper case) to move, 1 = 'b' (black).
 *   label   uint16 index into the 2086-entry move vocabulary labels_array (main.py:30-65,211).
 *   planes  [G][9][10][C] (C >= 14), element (h,w,c) = 1 iff the side-to-move-canonical board
 *           has piece code c+1 on cell h*9+w — the reference's 9-stride quirk (SURVEY Q1,
 *           main.py:547-557) is reproduced; channels >= 14 are zero padding.
 */
#ifndef CCHESS_HIP_H
#define CCHESS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CZ_NLABELS 2086
#define CZ_MAXMOVES 128
#define CZ_NSQ 90
#define CZ_MASK_WORDS 66 /* ceil(2086/32) */

#define CZ_OK 0
#define CZ_EINVAL (-1)
#define CZ_EHIP (-2)
#define CZ_ENOMEM (-3)

#define CZ_F32 0
#define CZ_BF16 1
#define CZ_F16 2 /* IEEE half: accepted where planes are written (cz_encode_planes, cz_search_select[_k]) */

/* status bits reported by cz_search_status */
#define CZ_ST_POOL_EXHAUSTED 1 /* node pool of the tree is full: expansion skipped */
#define CZ_ST_NO_MOVES 2       /* an expanded node with zero legal moves was selected (reference raises, quirk Q7) */
#define CZ_ST_MOVE_OVERFLOW 4  /* > 128 moves or a move without a label */
#define CZ_ST_BAD_ADVANCE 8    /* played move is not a child of the root (reference: KeyError) */

/* One self-play training record (cz_selfplay_*): the tuple cchess_main.selfplay appends per ply (main.py:1504-1518) in
 * packed form.  board = root position BEFORE the move in real coordinates (the canonical, side-to-move-first state of
 * the reference is try_flip of it for black, main.py:1505), side = mover, count = number of root children,
 * z = game result from the mover's point of view (main.py:1533-1544), labels/visits = the root's children in generation
 * order (root.child.items(), main.py:1339) with their visit counts N — pi = softmax(1/T * log(visits)) (main.py:1341) is
 * recomputed from them on the host in float64, bit for bit.  Unused label slots hold 0xFFFF, unused visits 0. */
#define CZ_REC_BYTES 608
#define CZ_REC_SIDE 90    /* uint8 */
#define CZ_REC_COUNT 91   /* uint8 */
#define CZ_REC_Z 92       /* int8: +1 / -1 / 0 */
#define CZ_REC_FLAGS 93   /* uint8: bit 0 = a visit count saturated at 65535 */
#define CZ_REC_PLY 94     /* uint16: index of the ply in its game */
#define CZ_REC_LABELS 96  /* uint16[128] */
#define CZ_REC_VISITS 352 /* uint16[128] */
/* cz_selfplay_stats slots */
#define CZ_SP_NSTATS 8
#define CZ_SP_GAMES 0      /* games finished (including stalled ones) */
#define CZ_SP_RED_WINS 1   /* 'k' captured: "w" wins */
#define CZ_SP_BLACK_WINS 2 /* 'K' captured: "b" wins */
#define CZ_SP_DRAWS 3      /* restrict_round >= 60 (main.py:1542) or the history capacity reached */
#define CZ_SP_PLIES 4      /* records handed to the ring */
#define CZ_SP_STALLED 5    /* games dropped because the root had no child to play (node pool exhausted) */
#define CZ_SP_DROPPED 6    /* records dropped because the ring was full */
#define CZ_SP_SIMS 7       /* simulations of the searches behind the moves played so far */

typedef struct cz_ctx cz_ctx;

const char *cz_last_error(void);
/* host utility: CRC-32C (Castagnoli) of a host buffer — the checksum of the reference's tf.train.Saver checkpoint files
 * (policy_value_network.py:148,176-184), read and written by cchess_zero_amd/tf_checkpoint.py */
unsigned int cz_crc32c(const void *data, size_t n);
int cz_version(void);

/* ---- static tables (host pointers, valid for the process lifetime) --------------------------
 * replaces: create_uci_labels / labels_array / label2i / unflipped_index, main.py:23-65,211-217.
 *   lut90x90[src*90+dst] -> label or -1;  unflip2086[i] = unflipped_index[i];
 *   labels = 2086 x 5 chars ("a0a1\0");   srcdst[i] = src | dst << 8. */
int cz_tables(const int16_t **lut90x90, const int16_t **unflip2086, const char **labels,
              const uint16_t **srcdst);
/* Zobrist keys used by cz_hash / cz_apply_move: keys[15][90] (row 0 unused), side key.
 * The reference has no state hash (identity = the state string); this format is ours. */
int cz_zobrist(const uint64_t **keys15x90, uint64_t *side_key);

/* ---- context ------------------------------------------------------------------------------
 * replaces: cchess_main.__init__ building GameBoard + MCTS_tree, main.py:1140-1144. */
int cz_create(int device, int max_games, int max_nodes_per_tree, cz_ctx **out);
void cz_destroy(cz_ctx *);
int cz_set_stream(cz_ctx *, void *hip_stream /* hipStream_t, NULL = default stream */);
int cz_synchronize(cz_ctx *);
int cz_malloc(cz_ctx *, size_t bytes, void **dptr);
int cz_free(cz_ctx *, void *dptr);
int cz_upload(cz_ctx *, void *dst_device, const void *src_host, size_t bytes);
int cz_download(cz_ctx *, void *dst_host, const void *src_device, size_t bytes);

/* ---- rules kernels (stand-alone, G positions per launch) -----------------------------------
 * K1  replaces GameBoard.get_legal_moves, main.py:743-1109.
 *     moves [G][128] labels in the reference's generation order (0xFFFF padding), count [G];
 *     mask [G][66] 2086-bit legality mask (bit i of word i/32).  moves or mask may be NULL.
 *     Alignment: `moves` must be 16-byte aligned (the list rows leave as 16-byte stores; CZ_EINVAL otherwise); boards, side,
 *     count and mask may sit at any byte / element address (unaligned inputs take a slower staging path).
 *     A board this library cannot be asked about answers count 0xFFFF (its list / mask row is undefined): more than 16 pieces of
 *     the side to move, more of a kind than a Xiangqi set holds (3 rooks / cannons / knights / advisors / bishops, 6 pawns, 2
 *     kings of the side to move — the one-lane-per-position generators take a kind's squares as the lowest and highest of its
 *     set), or an advisor / bishop step the 2086-label vocabulary has no label for. */
int cz_movegen(cz_ctx *, const uint8_t *boards, const uint8_t *side, int G, uint16_t *moves,
               uint16_t *count, uint32_t *mask);
/*     cz_movegen_ex: the same with flags.  CZ_MOVES_NO_PAD: row g of `moves` is written up to count[g] only (in 16-byte pieces; the
 *     labels behind count[g] are UNDEFINED, not 0xFFFF) — the padding is two thirds of a row (~40 moves per position, main.py:743)
 *     and was 1.97x the ordered kernel's algorithmic HBM traffic; a caller that reads rows up to `count`, as every consumer of
 *     get_legal_moves' list does, loses nothing.  flags = 0 is cz_movegen.
 *     The count-0xFFFF restriction above is the stand-alone kernels' only: the generator INSIDE the search (k_select, lane =
 *     piece, csrc/cz_device.h) generates moves for any board the reference's get_legal_moves does. */
#define CZ_MOVES_NO_PAD 1
int cz_movegen_ex(cz_ctx *, const uint8_t *boards, const uint8_t *side, int G, uint16_t *moves,
                  uint16_t *count, uint32_t *mask, int flags);
/* K2  replaces GameBoard.sim_do_action (main.py:647-702), is_kill_move (:226) and the king test
 *     (:409-413).  Updates boards/side in place.  hash: in/out incremental Zobrist (may be NULL);
 *     captured [G] = captured piece code or 0; terminal [G]: bit0 'K' missing, bit1 'k' missing.
 *     move_label 0xFFFF leaves the game untouched. */
int cz_apply_move(cz_ctx *, uint8_t *boards, uint8_t *side, const uint16_t *move_label, int G,
                  uint64_t *hash, uint8_t *captured, int8_t *terminal);
int cz_hash(cz_ctx *, const uint8_t *boards, const uint8_t *side, int G, uint64_t *hash);
/* K3  replaces MCTS_tree.generate_inputs = try_flip + state_to_positions, main.py:531-574.
 *     planes [G][9][10][channels] of dtype CZ_F32 / CZ_BF16.  quirk_q1 = 1 reproduces the
 *     reference bit for bit; 0 gives the transposed encoding its shapes suggest. */
int cz_encode_planes(cz_ctx *, const uint8_t *boards, const uint8_t *side, int G, void *planes,
                     int dtype, int channels, int quirk_q1);

/* ---- lock-step search over G trees, one wavefront per tree ---------------------------------
 * replaces: MCTS_tree.__init__/reload (main.py:235-259): fresh, unexpanded roots. */
int cz_search_reset(cz_ctx *, const uint8_t *root_boards, const uint8_t *root_side,
                    const int32_t *restrict_round /* may be NULL = 0 */, int G);
/* replaces: MCTS_tree.reload / GameBoard.reload when a game is over (main.py:255-258,604-608,1494,1552): the trees g with
 *   which[g] != 0 start afresh (unexpanded root, no simulations) from root_boards[g] / root_side[g] / restrict_round[g]
 *   ([G] arrays like cz_search_reset's; restrict_round may be NULL = 0); every other tree is left untouched. */
int cz_search_reload(cz_ctx *, const uint8_t *which, const uint8_t *root_boards, const uint8_t *root_side,
                     const int32_t *restrict_round);
/* K4 (+K1,K2,K3 on the leaf)  replaces one start_tree_search descent, main.py:350-418
 *   (select_new :158, get_Q_plus_U_new :108-116, kill-move/restrict_round :393-396, terminal
 *   tests :409-416, generate_inputs :362, get_legal_moves :374) with search_threads = 1.
 *   mode 0: root expansion only (MCTS_tree.main, main.py:475-487); mode 1: one simulation.
 *   active [G] (may be NULL): trees with 0 idle this step.
 *   leaf_planes [G][9][10][channels]; needs_eval [G] = 1 where the net must be evaluated. */
int cz_search_select(cz_ctx *, int mode, const uint8_t *active, void *leaf_planes, int dtype,
                     int channels, uint8_t *needs_eval);
/* K5+K6  replaces leaf_node.expand (main.py:175-187) with flip_policy (:1153-1155) and
 *   back_up_value (:189-194) along the unwind (:426-435), including the float32 effect of the
 *   virtual-loss add/remove (:403-404,426-427).  logits [G][2086], value [G] of `dtype`. */
int cz_search_expand_backup(cz_ctx *, const void *logits, const void *value, int dtype);

/* cz_search_expand_backup with the policy FC (policy_value_network.py:62-63) folded in: instead of a full logits row
 * per tree it takes the head-conv outputs z [G][90][3] f32 (cz_net_trunk_*), the FC weight pfc_w [2086][180] f32
 * (row = label, torch / "out,in" layout; the reference stores [in,out]) and bias pfc_b [2086], and evaluates the FC
 * only for the <= 128 labels each expansion reads (leaf_node.expand gathers action_probs[label2i[action]], main.py:
 * 179-181).  The float32 evaluation order of the 180-term dot product is fixed and documented at k_expand_backup
 * (cz_search.hip) so that a CPU restatement reproduces the priors bit for bit (tests/test_hip_search.py).
 * value [G] f32.  Pairs with cz_search_select (one simulation per tree). */
int cz_search_expand_backup_fc(cz_ctx *, const float *z, const float *value, const float *pfc_w, const float *pfc_b,
                               int compact /* 1: z / value rows are the ones cz_search_select_compact handed out */);

/* Compact evaluation batches.  Terminal and drawn leaves (main.py:409-416) and parked trees need no net evaluation — the
 * reference never calls forward() for them — so cz_search_select_compact writes the leaf planes of the trees that DO
 * need one to consecutive rows 0 .. n-1 of leaf_planes (rows are handed out by an atomic counter; which tree gets
 * which row is irrelevant because every row of the net is computed independently).  *slot_of -> device int32 [G]: the
 * row of tree g's leaf or -1; *n_rows -> device int32: n for this step (valid until the step after the next).  Pass
 * n_rows to cz_set_batch_count so that the following cz_net_trunk_bf16 / cz_net_trunk_f16 / cz_fc_heads_f32 launches
 * stop after n rows (whole workgroups beyond them return at once; no host synchronisation), then call
 * cz_search_expand_backup_fc(..., compact = 1).  cz_search_eval_totals returns the running totals of rows evaluated
 * and of compact steps (synchronises the stream): the flop accounting of a benchmark. */
int cz_search_select_compact(cz_ctx *, int mode, const uint8_t *active, void *leaf_planes, int dtype, int channels,
                             const int32_t **slot_of, const int32_t **n_rows);
int cz_set_batch_count(cz_ctx *, const int32_t *n_rows_dev /* or NULL: every row of B */);
int cz_search_eval_totals(cz_ctx *, unsigned long long *rows, unsigned long long *steps);
/* k simulations in flight per tree and step, with the reference's virtual loss (N += 3, W -= 3 while in flight):
 * replaces the `search_threads` coroutines of MCTS_tree (asyncio.Semaphore(search_threads), main.py:250,337-348,
 * virtual loss :231,403-404,426-427, now_expanding :354-360).  cz_search_set_width sizes the pending-leaf arrays
 * (1 <= width <= 64, default 1; call before the search loop).  leaf_planes / needs_eval / logits / value then have
 * G*k rows, slot g*k + j = j-th descent of tree g; slots a tree does not use have needs_eval = 0.  k = 1 is
 * arithmetically identical to cz_search_select / cz_search_expand_backup. */
int cz_search_set_width(cz_ctx *, int width);
/* cz_search_set_sim_target: with k > 1 a tree stops issuing descents once `target` simulations (counted since the last
 * cz_search_reset / cz_search_advance, cz_search_status `sims`) are completed or in flight, so that a search of
 * ceil(playouts / k) + a few steps ends with EXACTLY `playouts` simulations per tree like MCTS_tree.main (main.py:489-493:
 * `playouts` coroutines, at most search_threads of them in flight).  0 = no limit (default). */
int cz_search_set_sim_target(cz_ctx *, int target);
/* cz_search_set_terminal_extra: a simulation that ends on a king capture or on the 60-ply rule needs no net evaluation
 * (main.py:409-416 returns before push_queue).  With n > 0, cz_search_select[_compact] completes up to n such simulations
 * per tree inside the launch (value backed up along the path, simulation counted) and goes on to the next descent, so that
 * every tree presents a leaf that DOES need the net: a lock-step then completes 1 / (1 - f) simulations per net row.
 * Simulations of one tree stay strictly sequential: the tree after N simulations is bit-identical to the n = 0 schedule.
 * Use cz_search_set_sim_target (also honoured by cz_search_select) to stop every tree at exactly N.  Default 0. */
int cz_search_set_terminal_extra(cz_ctx *, int n);
/* Evaluation cache (off by default; width 1 only).  The net is a pure function of (board, side to move) and this library
 *   computes every row of a batch independently of the others, so a position a tree has evaluated before would get the
 *   same priors and value again, bit for bit (the reference re-evaluates it: transpositions are separate nodes,
 *   main.py:357-384).  With the cache on, every expanded node is remembered under the 64-bit Zobrist key of its position
 *   (cz_zobrist; per tree 128 buckets x 64 entries of {key, node, value, the position itself packed into 48 bytes} =
 *   512 KB); a leaf whose key is found AND whose position equals the stored one is expanded inside the select launch from
 *   the remembered node's children (labels, priors) and backed up with the remembered value — no net row.  The key only
 *   finds the candidate, the position decides: a key collision is a miss (counted), never a wrong node.  Up to 4 hits
 *   complete per tree and launch, a budget of their own beside cz_search_set_terminal_extra (the cache works with
 *   terminal_extra = 0).  Trees are identical with the cache on or off; entries follow their nodes through
 *   cz_search_advance and are dropped with them.  Turning it on empties it.
 *   cz_search_eval_cache_stats: hits / lookups summed over the trees since the cache was turned on (synchronises);
 *   cz_search_eval_cache_collisions: key matches refused because the stored position differed;
 *   cz_search_debug_eval_cache_key_bits (tests): keys narrowed to 8..24 bits so that collisions DO occur (64 = full). */
int cz_search_set_eval_cache(cz_ctx *, int on);
int cz_search_eval_cache_stats(cz_ctx *, unsigned long long *hits, unsigned long long *lookups);
int cz_search_eval_cache_collisions(cz_ctx *, unsigned long long *collisions);
int cz_search_debug_eval_cache_key_bits(cz_ctx *, int bits);
/* Cross-tree level of the evaluation cache (needs cz_search_set_eval_cache(ctx, 1)): one table of 2^log2_entries self-contained,
 * entries (key, packed position, value, the <= 128 labels and priors: 1088 bytes each) shared by all trees of the
 * context — in self-play from the start position the games share their openings, and the reference evaluates the same position
 * once per game (main.py:357-384).  A leaf whose position ANY tree has evaluated is expanded inside the select launch from the
 * remembered row, verified against the stored position; trees stay bit-identical (the net's row for a position does not depend
 * on the tree it is asked for).  log2_entries 0 frees the table; calling it again (or cz_search_set_eval_cache(ctx, 1)) empties
 * it — do so whenever the weights change.  Round 6: a FULL bucket is no longer closed — the entry whose position lies deepest in
 * its game (re-roots of the filing tree + depth of the leaf) is replaced when the new position is shallower, so the table converges
 * to the openings every restarted game walks through again.  stats4: hits, lookups, entries written, filings that found no room
 * (a full bucket of shallower positions, or every swap lost to another tree of the launch); stats5 adds: entries that replaced
 * a deeper one (included in `written`). */
int cz_search_set_xcache(cz_ctx *, int log2_entries);
int cz_search_xcache_stats(cz_ctx *, unsigned long long *stats4);
int cz_search_xcache_stats5(cz_ctx *, unsigned long long *stats5);
/* tests: cz_search_advance keeps the kept-node bitmap of a tree in LDS when it fits (12 bytes per 64 nodes) and in global memory
 * otherwise; on != 0 forces the global-memory kernel so that it is exercised at test sizes. */
int cz_search_debug_advance_in_global_memory(cz_ctx *, int on);
int cz_search_select_k(cz_ctx *, int mode, int k, const uint8_t *active, void *leaf_planes, int dtype,
                       int channels, uint8_t *needs_eval);
int cz_search_expand_backup_k(cz_ctx *, int k, const void *logits, const void *value, int dtype);
/* replaces: reading root.child.items() in get_action, main.py:1339 (+ MCTS_tree.Q :261).
 *   arrays [G][128] (any may be NULL), count [G]. */
int cz_search_root_stats(cz_ctx *, uint16_t *move_label, int32_t *N, float *Q, float *P, float *W,
                         uint16_t *count);
/* K7  replaces MCTS_tree.update_tree (main.py:272-276) + the board bookkeeping of selfplay
 *   (:1522-1528): re-root on the played child keeping its subtree (compacted in place: one node
 *   pool per tree); clears CZ_ST_POOL_EXHAUSTED.  played_label 0xFFFF leaves the tree untouched. */
int cz_search_advance(cz_ctx *, const uint16_t *played_label);
/* greedy re-rooting driver (device arrays, no host round trip; bench.py's search loop, "play the strongest move" loops)
 * replaces: get_action in its temperature -> 0 limit (main.py:1332-1341, default temperature 1e-3: softmax(log(visits)/T)
 *   puts all mass on the most visited child; first maximum in generation order like Python's max()) for the trees whose
 *   search is complete.  A tree is READY when it has completed sim_threshold[g] simulations since its last reset /
 *   advance or its node pool is full: played[g] = label of its most visited root child (0xFFFF: not ready, or no child),
 *   ready[g] = 1 / 0 (may be NULL), sim_threshold[g] = next_threshold, *banked_sims += its simulation count (device
 *   scalar, may be NULL).  Follow with cz_search_advance(played) and cz_search_reload_finished. */
int cz_search_pick_ready(cz_ctx *, int32_t *sim_threshold /*[G] in/out*/, int next_threshold, uint16_t *played /*[G]*/,
                         uint8_t *ready /*[G]*/, unsigned long long *banked_sims);
/* replaces: check_end + reload after the move (main.py:1380-1392 king gone / restrict_round >= 60; :255-258,604-608):
 *   the READY trees whose game is now over — or that had no move to play — start afresh from root_boards[g] /
 *   root_side[g] / restrict_round[g] (may be NULL = 0) like cz_search_reload; *reloaded (device scalar, may be NULL)
 *   counts them. */
int cz_search_reload_finished(cz_ctx *, const uint8_t *ready, const uint16_t *played, const uint8_t *root_boards,
                              const uint8_t *root_side, const int32_t *restrict_round, unsigned long long *reloaded);
int cz_search_status(cz_ctx *, int32_t *status, int32_t *nodes_used, int32_t *sims,
                     int32_t *last_depth); /* device [G] arrays, any may be NULL */
int cz_search_root_state(cz_ctx *, uint8_t *boards, uint8_t *side, int32_t *restrict_round);
/* parity/debug: pre-order dump of tree g into HOST memory; record = 7 int32
 *   {depth, label, N, bits(W), bits(Q), bits(P), child_count or -1}.  Returns the record count
 *   (>= 0; at most max_records are written) or a negative error. */
int cz_search_tree_dump(cz_ctx *, int g, int32_t *host_out, int max_records);

/* ---- device-resident self-play bookkeeping (one wave per game; no host round trip per ply) ------------------------
 * replaces: cchess_main.get_action (main.py:1332-1358) and the per-ply / end-of-game bookkeeping of cchess_main.selfplay
 * (:1493-1554) for the G games of the ctx at once.  One ply of every game =
 *     cz_search_select/expand_backup... (the playouts; pass cz_selfplay_active as the select `active` mask)
 *     cz_selfplay_choose      pi from the root visits, the sampled move, the (s, pi) record of the ply
 *     cz_search_advance       update_tree + board bookkeeping with the chosen moves
 *     cz_selfplay_adjudicate  game end tests on the new positions, z for the finished games' records, re-seed
 *     cz_selfplay_flush       finished games' records -> the caller's record ring
 * cz_selfplay_begin (after cz_search_reset): allocates the per-slot histories [G][max_plies] records and makes the
 *   given positions (NULL: the trees' current roots) the ones every new game of a slot starts from; zeroes the statistics.
 *   A game that reaches max_plies plies is adjudicated a draw (the reference has no such limit; its games end by the
 *   60-ply no-capture rule).
 * cz_selfplay_choose: gamma [G][128] f32 Gamma(0.3, 1) variates (normalised per game = Dirichlet(0.3), main.py:1346) or
 *   NULL, u [G] f32 uniforms in [0, 1), forced [G] labels overriding the sampled move (0xFFFF = none; NULL) for replaying
 *   recorded games; temperature as in get_action; noise_eps = 0.25 when exploring, 0 otherwise.  min_sims = 0: every
 *   active game moves (lock-step plies); min_sims > 0: only the games whose current search has completed that many
 *   simulations (or whose tree cannot go on) move — asynchronous plies, every game at its own pace.  played [G] out
 *   (0xFFFF for parked games, for games that do not move now and for games whose root has no child).
 * cz_selfplay_adjudicate: reseed != 0: a finished slot starts a new game at once; 0: it is parked (cz_selfplay_active
 *   turns 0).  played: the array cz_selfplay_choose filled (slots that did not move are skipped) or NULL (all slots).
 *   fin_n [G] out: records the finished game contributes (0 otherwise).
 * cz_selfplay_flush: offset [G] int64 = first ring index (before the modulo) of each finished game, i.e. an exclusive
 *   prefix sum of fin_n on top of the running cursor, computed by the caller; ring [ring_records][CZ_REC_BYTES];
 *   read_cursor: device int64 the caller advances after draining (records that would overwrite undrained ones are
 *   dropped and counted), or NULL.
 * cz_selfplay_stats: device int64 [CZ_SP_NSTATS] <- running totals. */
int cz_selfplay_begin(cz_ctx *, int max_plies, const uint8_t *start_boards, const uint8_t *start_side,
                      const int32_t *start_rr);
int cz_selfplay_active(cz_ctx *, const uint8_t **active_dev /* [G] */);
int cz_selfplay_choose(cz_ctx *, const float *gamma, const float *u, const uint16_t *forced, double temperature,
                       float noise_eps, int min_sims, uint16_t *played);
int cz_selfplay_adjudicate(cz_ctx *, int reseed, const uint16_t *played, int32_t *fin_n);
int cz_selfplay_flush(cz_ctx *, const int32_t *fin_n, const long long *offset, uint8_t *ring, long long ring_records,
                      const long long *read_cursor);
int cz_selfplay_stats(cz_ctx *, long long *stats_dev);

/* ---- N1: policy/value network kernels ---------------------------------------------------------
 * One residual-tower layer: 3x3 SAME convolution 128->128 over [B][9][10] boards, NHWC bf16, with the
 * (BN-folded) bias, optional residual add and optional ReLU fused in.
 * replaces: tf.layers.conv2d(128,3,'SAME') + tf.contrib.layers.batch_norm(center=False) [+ tf.add(orig,.)]
 *           + tf.nn.relu, policy_value_network.py:45-47 and residual_block :151-162.
 *   in, residual, out : [B][90][128] bf16 (residual may be NULL; out must not alias in)
 *   wpk  : [9 taps = dy*3+dx][16 = ci/8][128 co][8 = ci%8] bf16, BN scale folded in (net.py packs it)
 *   bias : [128] float32, BN folded */
int cz_conv3x3_c128_bf16(cz_ctx *, const void *in, const void *wpk, const float *bias,
                         const void *residual, void *out, int B, int relu);

/* The whole residual tower (nblocks x [conv3x3+BN+ReLU, conv3x3+BN, add, ReLU]) in ONE launch:
 * activations stay in LDS across all 2*nblocks layers, only `in` and `out` touch HBM.
 * replaces: the `for _ in range(res_block_nums): residual_block(...)` loop, policy_value_network.py:50-53,151-162.
 *   in, out : [B][90][128] bf16 (may alias)
 *   wpk  : [2*nblocks][9][16][128][8] bf16 (layer-major, same per-layer packing as cz_conv3x3_c128_bf16)
 *   bias : [2*nblocks][128] float32 */
int cz_tower_c128_bf16(cz_ctx *, const void *in, const void *wpk, const float *bias, void *out, int B,
                       int nblocks);
/* Same launch with the two head 1x1 convolutions fused in (conv1x1 128->2 policy, 128->1 value, BN folded,
 * ReLU; policy_value_network.py:57-59,68-70), computed from the LDS-resident trunk:
 *   head_w [3][128] f32 (rows: policy ch0, policy ch1, value), head_b [3] f32,
 *   head_out [B][90][3] f32 (post-ReLU; [:, :, 0:2] flattens to the 180 policy-FC inputs in the reference's
 *   (h, w, c) order, [:, :, 2] to the 90 value-FC inputs).  trunk_out may be NULL (trunk never leaves the CU). */
int cz_tower_heads_c128_bf16(cz_ctx *, const void *in, const void *wpk, const float *bias, void *trunk_out,
                             const float *head_w, const float *head_b, float *head_out, int B, int nblocks);

/* Input planes to head-conv outputs in ONE launch: conv3x3(14->128)+BN+ReLU (policy_value_network.py:45-47),
 * the residual tower and (optionally) the head 1x1 convolutions.
 *   planes16 : [B][90][16] bf16 — the K3 encoding with channels = 16 (14 planes + 2 zero channels), e.g. written by
 *              cz_search_select(..., CZ_BF16, 16, ...)
 *   w0 : [9 taps][2 = ci/8][128 co][8 = ci%8] bf16 (input channels 14,15 zero), b0 : [128] f32, BN folded
 *   other arguments as cz_tower_heads_c128_bf16; trunk_out and head_out may each be NULL (not both). */
int cz_net_trunk_bf16(cz_ctx *, const void *planes16, const void *w0, const float *b0, const void *wpk,
                      const float *bias, void *trunk_out, const float *head_w, const float *head_b,
                      float *head_out, int B, int nblocks);

/* cz_net_trunk_bf16 with IEEE fp16 activations and weights (fp32 accumulate): the reference's "19-block net fp16"
 * configuration (BASELINE.json configs[4]).  Same argument layout; planes16, w0, wpk and trunk_out hold fp16. */
int cz_net_trunk_f16(cz_ctx *, const void *planes16, const void *w0, const float *b0, const void *wpk,
                     const float *bias, void *trunk_out, const float *head_w, const float *head_b,
                     float *head_out, int B, int nblocks);

/* The STRICT-precision form of cz_net_trunk_*: every weight and every stored activation is carried as hi + lo, two values
 * of `halves_dtype` (CZ_F16: 22 significant bits, CZ_BF16: 16), and every product is three MFMAs (hi*hi + hi*lo + lo*hi,
 * fp32 accumulate) — the engine that meets north_star's "policy/value outputs within 1e-3 of fp32" for the reference's
 * fp32 sess.run (policy_value_network.py:202-214) also on trained (peaked) weights and at 19 blocks, at a third of the
 * 16-bit kernels' rate.
 *   planes16 : [B][90][16] of halves_dtype (0/1 planes are exact: the same buffer cz_search_select writes)
 *   w0   : [9 taps][hi, lo][2 = ci/8][128 co][8 = ci%8], b0 [128] f32
 *   wpk  : [2*nblocks][9 taps][4 = 32-channel quarter of the tap][hi, lo][4 = ci/8][128 co][8] (16 KB per quarter: one
 *          LDS-DMA slab), bias [2*nblocks][128] f32; BN folded; hi = rn(w), lo = rn(w - hi) (net.py packs it)
 *   trunk_out : [B][90][128] FLOAT32 (hi + lo) or NULL; head_w / head_b / head_out as cz_net_trunk_bf16. */
int cz_net_trunk_split(cz_ctx *, const void *planes16, const void *w0, const float *b0, const void *wpk,
                       const float *bias, float *trunk_out, const float *head_w, const float *head_b,
                       float *head_out, int B, int nblocks, int halves_dtype);

/* The strict-precision trunk at half the matrix work (round 5; k_trunk_mx_c128, csrc/cz_trunk_mx.h): a*w = a_hi*w_hi on fp16
 * MFMAs + BOTH cross terms (a_hi*w_lo + a_lo*w_hi) of 32 input channels on ONE block-scaled fp6 MFMA
 * (v_mfma_scale_f32_32x32x64_f8f6f4, E2M3 operands, one E8M0 scale per cell / output channel and 16 input channels):
 * 1.5 instead of 3 fp16-MFMA-equivalents per product, the cross terms to 4 significant bits per operand (~2^-16 of a product).
 * Replaces the same fp32 sess.run (policy_value_network.py:202-214); |dlogit|, |dvalue| <= 1e-3 against it on trained-like
 * weights up to 8 blocks with a factor of two to spare, at the edge of 1e-3 at 19 (tests/test_net.py, tests/mxemu.py is its CPU
 * emulation).  Which of this and cz_net_trunk_split a net runs is MEASURED on its live weights by the host side (precision
 * "strict", cchess_zero_amd/net.py: strict_check), not decided by depth.
 *   planes16 : [B][90][16] fp16; w0 / b0 : as cz_net_trunk_split with CZ_F16 (the first layer is two fp16 MFMAs per tap)
 *   wpk  : [2*nblocks][9 taps][4 quarters][16384 bytes]: per 32-input-channel quarter of a tap (one LDS-DMA slab)
 *          [fp16 w_hi: 4 = ci/8][128 co][8] (8192 B) [fp6 blocks, first 16 bytes: 2 halves][128 co][16] (4096 B)
 *          [last 8 bytes: 2][128 co][8] (2048 B) [E8M0 scale in byte 0 of a dword: 2][128 co] (1024 B) [1024 B unused];
 *          block (co, half h) = 32 six-bit slots, slot 2j = q6(2^11 w_lo[c_j]), slot 2j+1 = q6(w_hi[c_j]),
 *          c_j = 32 quarter + 8 (j / 4) + 4 h + j % 4, under 2^(exponent(max |w_hi|) - 2); the stored byte is that exponent
 *          - 11 (cchess_zero_amd/net.py: mx_pack_layer); 16-byte aligned.  bias [2*nblocks][128] f32, BN folded.
 *   trunk_out : [B][90][128] FLOAT32 or NULL; head_w / head_b / head_out as cz_net_trunk_bf16. */
int cz_net_trunk_mx(cz_ctx *, const void *planes16, const void *w0, const float *b0, const void *wpk,
                    const float *bias, float *trunk_out, const float *head_w, const float *head_b,
                    float *head_out, int B, int nblocks);

/* Measurement hook for bench.py (roofline.effective_clock_GHz): while buf_dev is set, every workgroup w of the following
 * cz_net_trunk_* launches (grids up to max_workgroups) writes buf_dev[4w .. 4w+3] = {shader-clock cycle counter at its start,
 * at the end of its last layer, 100 MHz reference clock at the same two points}: cycles / (ticks * 10 ns) is the clock the
 * kernel really ran at under the chip's power governor.  NULL switches it off (the default; one uniform branch per
 * workgroup).  cz_clock_probe_last_grid: workgroups of the last probed launch. */
int cz_set_clock_probe(cz_ctx *, unsigned long long *buf_dev, int max_workgroups);
int cz_clock_probe_last_grid(cz_ctx *);

/* Measurement: back-to-back v_mfma_f32_32x32x16 (dtype CZ_BF16 | CZ_F16) on register operands, two waves per SIMD, every CU:
 * the practical MFMA ceiling of this chip under its power governor, measured in the bench run itself (tools/mfma_peak.hip
 * is the stand-alone form).  data: 0 dense random operands, 1 zeros, 2 random with half the elements zero; iters x 48
 * MFMAs per wave.  *tflops, *ms: dense-equivalent TFLOP/s and duration of the (second) launch.  Synchronises the stream. */
int cz_probe_mfma_peak(cz_ctx *, int dtype, int data, int iters, double *tflops, double *ms);

/* The three fully connected layers behind the head convolutions (policy_value_network.py:62-63,72-74): policy FC
 * 180 -> 2086 (raw logits) and value FC 90 -> 256, ReLU, FC 256 -> 1, tanh, from the head conv outputs
 * z [B][90][3] f32 as cz_net_trunk_bf16 / cz_tower_heads_c128_bf16 leave them (flatten order (h,w,c)).
 *   pfc_w_hi, pfc_w_lo : policy FC weight split into bf16 hi + lo parts (w ~= hi + lo), each packed in MFMA
 *                        fragment order [66 = label/32][12 = k/16][64 lanes][8] bf16 with
 *                        element = W[label = tile*32 + (lane & 31)][k = kblock*16 + (lane >> 5)*8 + j], zero padded
 *   pfc_b [2086] f32;  v1_wt [90][256] f32 (value FC1 weight, input-major);  v1_b [256];  v2_w [256];  v2_b [1]
 *   logits [B][2086] f32, value [B] f32 (either may be NULL to skip that head). */
int cz_fc_heads_f32(cz_ctx *, const float *z, const void *pfc_w_hi, const void *pfc_w_lo, const float *pfc_b,
                    const float *v1_wt, const float *v1_b, const float *v2_w, const float *v2_b,
                    float *logits, float *value, int B);

#ifdef __cplusplus
}
#endif
#endif
