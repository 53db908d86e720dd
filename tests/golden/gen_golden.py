#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by running the UNMODIFIED reference.

Run in the build container (where /root/reference exists):
    python tests/golden/gen_golden.py
The reference is imported through tests/golden/ref_harness.py (stub tensorflow/uvloop,
awaitable Semaphore; no reference source is modified or copied).  The outputs are small
and committed, so the parity tests can run where the reference is absent (GPU box).

Fixtures
  tables.json  label-table pins (main.py:30-65, 211-217)
  rules.npz    positions from seeded random playouts + crafted edge cases:
               ordered get_legal_moves (main.py:743), sim_do_action (:647), is_kill_move
               (:226), generate_inputs planes (:531); every position is also cross-checked
               against the GUI rules (ChessBoard.py / chessman/*.py can_move) as move SETS.
  mcts.json    MCTS_tree.main (main.py:473) runs with search_threads=1 and the exact-integer
               fake forward of tests/fakenet.py: root children (label, N, W, Q, P bit
               patterns), whole-tree digests, the ordered list of evaluated positions.
  mcts_deep.json  the same at the metric's depth — playout 1600, one black-to-move root, one near the 60-ply rule, one
               across update_tree — and with the 'deep' fake forward whose searches select paths of 40-60 levels.
  policy_update.npz  cchess_main.policy_update (main.py:1157-1204) through a stand-in self with a scripted net: train steps
               taken, learning rates passed, lr_multiplier after, the logged KL / explained variances.
  selfplay.npz cchess_main.selfplay (main.py:1493) games with search_threads=1, the fake forward and
               a seeded np.random: per ply (canonical state, pi[2086] float64, z), the root
               children's visit counts and the sampled move (get_action, main.py:1332-1358).
"""
import hashlib
import json
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import ref_harness as rh  # noqa: E402
import fakenet  # noqa: E402

PIECES = ".KARBNPCkarbnpc"
START = "RNBAKABNR/9/1C5C1/P1P1P1P1P/9/9/p1p1p1p1p/1c5c1/9/rnbakabnr"


def fen_to_board(fen):
    b = np.zeros(90, np.uint8)
    y = 0
    for row in fen.split("/"):
        x = 0
        for ch in row:
            if ch.isdigit():
                x += int(ch)
            else:
                b[y * 9 + x] = PIECES.index(ch)
                x += 1
        assert x == 9, fen
        y += 1
    assert y == 10, fen
    return b


def f32bits(x):
    return int(np.asarray(x, dtype=np.float32).reshape(-1)[0].view(np.uint32)) if np.ndim(x) else int(np.float32(x).view(np.uint32))


# ----------------------------------------------------------------------------- chessman oracle
def chessman_moves(cb_mod, fen, side):
    """Move set according to ChessBoard.py / chessman/*.py (GUI rules)."""
    from chessman.Bing import Bing
    from chessman.Che import Che
    from chessman.Ma import Ma
    from chessman.Pao import Pao
    from chessman.Shi import Shi
    from chessman.Shuai import Shuai
    from chessman.Xiang import Xiang
    cls = {"k": Shuai, "a": Shi, "r": Che, "b": Xiang, "n": Ma, "p": Bing, "c": Pao}
    cb_mod.ChessBoard.pieces.clear()  # class-level dict (quirk Q10)
    board = cb_mod.ChessBoard.__new__(cb_mod.ChessBoard)
    b = fen_to_board(fen)
    for sq in range(90):
        c = b[sq]
        if not c:
            continue
        ch = PIECES[c]
        x, y = sq % 9, sq // 9
        red = ch.isupper()
        # north_is_red=True layout: red pieces are the "north" army (ChessBoard.py:19-40)
        cb_mod.ChessBoard.pieces[x, y] = cls[ch.lower()](x, y, red, "north" if red else "south")
    out = set()
    with rh.quiet():
        for (x, y), p in list(cb_mod.ChessBoard.pieces.items()):
            if p.is_red != (side == "w"):
                continue
            for (nx, ny) in p.get_move_locs(board):
                out.add("abcdefghi"[x] + str(y) + "abcdefghi"[nx] + str(ny))
    return out


CRAFTED = [
    # kings facing on an open file (flying general, main.py:1097-1107)
    ("4K4/9/9/9/9/9/9/9/9/4k4", "w"), ("4K4/9/9/9/9/9/9/9/9/4k4", "b"),
    ("3K5/9/9/9/9/9/9/9/9/3k5", "w"), ("5K3/9/9/9/4P4/9/9/9/9/5k3", "b"),
    # one blocker between the kings
    ("4K4/9/9/4P4/9/9/9/9/9/4k4", "w"), ("4K4/9/9/9/9/9/4p4/9/9/4k4", "b"),
    # cannons: 0 / 1 / 2 screens on rank and file
    ("4K4/9/9/9/C3p3r/9/9/9/9/4k4", "w"), ("4K4/9/9/9/C1P1p3r/9/9/9/9/4k4", "w"),
    ("4K4/9/9/9/C1P1p1n1r/9/9/9/9/4k4", "w"), ("c3K4/9/P8/9/R8/9/9/9/9/4k4", "b"),
    ("c3K4/P8/P8/9/R8/9/9/9/9/4k4", "b"),
    # pawns either side of the river, on the edges
    ("4K4/9/9/9/p7p/P7P/9/9/9/4k4", "w"), ("4K4/9/9/9/p7p/P7P/9/9/9/4k4", "b"),
    ("4K4/9/9/9/9/9/9/9/9/P3k3P", "w"), ("p3K3p/9/9/9/9/9/9/9/9/4k4", "b"),
    # knights with blocked legs, bishops with blocked eyes, advisors/kings at palace edges
    ("1N2K2N1/1P5P1/9/9/9/9/9/9/9/4k4", "w"), ("N3K3N/9/9/9/9/9/9/9/1p5p1/1n2k2n1", "b"),
    ("2B1K1B2/3P1P3/9/9/9/9/9/9/9/4k4", "w"), ("4K4/9/9/9/2B3B2/9/9/9/9/4k4", "w"),
    ("4K4/9/9/9/9/2b3b2/9/9/9/4k4", "b"), ("3AKA3/9/9/9/9/9/9/9/4a4/3k1a3", "b"),
    ("3K5/4A4/5A3/9/9/9/9/3a5/4a4/5k3", "w"), ("5K3/9/9/9/9/9/9/9/9/3k5", "b"),
    # rooks on all four edges / corners
    ("R3K3R/9/9/9/9/9/9/9/9/r3k3r", "w"), ("R3K3R/9/9/9/9/9/9/9/9/r3k3r", "b"),
    ("4K4/9/9/9/R7r/9/9/9/9/4k4", "w"), ("4K4/9/9/9/R7r/9/9/9/9/4k4", "b"),
    # dense middle game
    ("R1BAKAB1R/9/1CN3NC1/P1P1P1P1P/9/9/p1p1p1p1p/1cn3nc1/9/r1bakab1r", "w"),
    ("R1BAKAB1R/9/1CN3NC1/P1P1P1P1P/9/9/p1p1p1p1p/1cn3nc1/9/r1bakab1r", "b"),
]


def gen_tables(m, out):
    la = m.labels_array
    d = {
        "n_labels": len(la),
        "labels_sha256": hashlib.sha256("\n".join(la).encode()).hexdigest(),
        "unflip_sha256": hashlib.sha256(np.asarray(m.unflipped_index, dtype=np.int16).tobytes()).hexdigest(),
        "unflip_involution": all(m.unflipped_index[m.unflipped_index[i]] == i for i in range(len(la))),
        "unflip_fixed_points": sum(1 for i in range(len(la)) if m.unflipped_index[i] == i),
        "label2i": {k: m.label2i[k] for k in ["a0a1", "d7e8", "a2c4", "e0e9", "i9i0", "h7g9", "i7g9"]},
        "first_labels": la[:12],
        "last_labels": la[-4:],
        "pieces_order": m.pieces_order,
        "start_moves": m.GameBoard.get_legal_moves(START, "w"),
        "perft": {},
    }
    # pseudo-legal perft: 44 / 1926 / 80288
    def perft(state, player, depth):
        if depth == 0:
            return 1
        n = 0
        nxt = "b" if player == "w" else "w"
        for mv in m.GameBoard.get_legal_moves(state, player):
            n += perft(m.GameBoard.sim_do_action(mv, state), nxt, depth - 1) if depth > 1 else 1
        return n
    for dpt in (1, 2, 3):
        d["perft"][str(dpt)] = perft(START, "w", dpt)
    json.dump(d, open(os.path.join(out, "tables.json"), "w"), indent=1, sort_keys=True)
    print("tables:", d["labels_sha256"][:12], d["perft"])


def gen_rules(m, out, n_games=36, max_plies=110, seed=20260925):
    rng = random.Random(seed)
    cb_mod = rh.load_chessboard()
    positions = []  # (fen, player)
    king_capture_roots = []
    for g in range(n_games):
        state, player = START, "w"
        for ply in range(max_plies):
            positions.append((state, player))
            moves = m.GameBoard.get_legal_moves(state, player)
            if not moves:
                break
            # remember positions from which a king can be captured (search fixtures use them)
            for mv in moves:
                nxt = m.GameBoard.sim_do_action(mv, state)
                if "K" not in nxt or "k" not in nxt:
                    king_capture_roots.append((state, player))
                    break
            mv = rng.choice(moves)
            state = m.GameBoard.sim_do_action(mv, state)
            player = "b" if player == "w" else "w"
            if "K" not in state or "k" not in state:
                positions.append((state, player))  # king-less position: movegen must still agree
                break
    positions += CRAFTED
    # both sides to move for a subset (movegen is defined for either player on any state)
    extra = [(s, "b" if p == "w" else "w") for (s, p) in positions[::7]]
    positions += extra

    M = len(positions)
    boards = np.zeros((M, 90), np.uint8)
    side = np.zeros(M, np.uint8)
    counts = np.zeros(M, np.uint16)
    moves_arr = np.full((M, 128), 0xFFFF, np.uint16)
    chosen = np.full(M, 0xFFFF, np.uint16)
    next_boards = np.zeros((M, 90), np.uint8)
    kill = np.zeros(M, np.int8)
    planes_bits = np.zeros((M, 158), np.uint8)
    mcts = rh.new_mcts(START, None, 1)
    mismatches = 0
    checked = 0
    for i, (fen, pl) in enumerate(positions):
        boards[i] = fen_to_board(fen)
        side[i] = 1 if pl == "b" else 0
        mv = m.GameBoard.get_legal_moves(fen, pl)
        counts[i] = len(mv)
        assert len(mv) <= 128
        for j, a in enumerate(mv):
            moves_arr[i, j] = m.label2i[a]
        # second oracle: GUI rules as sets (positions with both kings only; Shuai logic needs pieces)
        cm = chessman_moves(cb_mod, fen, pl)
        checked += 1
        if cm != set(mv):
            mismatches += 1
            print("chessman mismatch:", fen, pl, sorted(cm ^ set(mv)))
        if mv:
            a = mv[rng.randrange(len(mv))]
            chosen[i] = m.label2i[a]
            nxt = m.GameBoard.sim_do_action(a, fen)
            next_boards[i] = fen_to_board(nxt)
            kill[i] = m.is_kill_move(fen, nxt)
        pl_planes = mcts.generate_inputs(fen, pl)
        assert pl_planes.shape == (9, 10, 14)
        planes_bits[i] = np.packbits(pl_planes.reshape(-1) > 0.5)
    np.savez_compressed(os.path.join(out, "rules.npz"), boards=boards, side=side, counts=counts, moves=moves_arr,
                        chosen=chosen, next_boards=next_boards, kill=kill, planes_bits=planes_bits,
                        chessman_checked=np.int64(checked), chessman_mismatches=np.int64(mismatches))
    print("rules: %d positions, avg %.1f moves, max %d; chessman set-equal: %d checked, %d mismatches" %
          (M, counts.mean(), counts.max(), checked, mismatches))
    return positions, king_capture_roots


def dump_tree(node):
    """pre-order, children in dict (= generation) order; same record as czo_search_tree_dump."""
    rec = []

    def walk(n, depth):
        for a, c in n.child.items():
            rec.append((depth, a, int(c.N), f32bits(c.W), f32bits(c.Q), f32bits(c.P), len(c.child) if c.child else -1))
            walk(c, depth + 1)
    walk(node, 0)
    return rec


def run_spec(m, sp, deep=False):
    """One golden search case: MCTS_tree.main per ply with the fake forward, most visited move played (update_tree).
    deep=True (mcts_deep.json): additionally the deepest level of the tree per ply, and the evaluated positions as a
    digest instead of a list (1600-playout cases evaluate ~1600 positions per ply)."""
    log = []
    fwd = fakenet.make_forward(sp["mode"], sp["salt"], log)
    state, player, rr = sp["fen"], sp["player"], sp["rr"]
    t = rh.new_mcts(state, fwd, 1)
    plies = []
    for po in sp["playouts"]:
        with rh.quiet(), np.errstate(all="ignore"):
            t.main(state, player, rr, po)
        root_children = [(m.label2i[a], int(c.N), f32bits(c.W), f32bits(c.Q), f32bits(c.P)) for a, c in t.root.child.items()]
        tree = dump_tree(t.root)
        digest = hashlib.sha256(np.asarray([(d, m.label2i[a], n, w, q, p, k) for (d, a, n, w, q, p, k) in tree],
                                           dtype=np.int64).astype(np.int32).tobytes()).hexdigest()
        # play the most visited move (first max), like update_tree after get_action
        best = max(t.root.child.items(), key=lambda kv: kv[1].N)[0]
        ply = dict(state=state, player=player, rr=rr, playouts=po, root=root_children,
                   tree_records=len(tree), tree_sha256=digest, played=m.label2i[best],
                   root_N=int(t.root.N), evals=len(log))
        if deep:
            ply["max_level"] = max(d for (d, a, n, w, q, p, k) in tree if n > 0)   # deepest VISITED node, root children = level 0
        plies.append(ply)
        nxt = m.GameBoard.sim_do_action(best, state)
        rr = rr + 1 if m.is_kill_move(state, nxt) == 0 else 0
        state = nxt
        player = "b" if player == "w" else "w"
        t.update_tree(best)
        if "K" not in state or "k" not in state:
            break
    case = dict(name=sp["name"], mode=sp["mode"], salt=sp["salt"], plies=plies)
    if deep:
        case["eval_keys_sha256"] = hashlib.sha256(",".join("%016x" % k for k in log).encode()).hexdigest()
    else:
        case["eval_keys"] = ["%016x" % k for k in log]
    print("mcts case %-18s plies=%d evals=%d tree=%s%s" % (sp["name"], len(plies), len(log), [p["tree_records"] for p in plies],
                                                         (" max level %s" % [p["max_level"] for p in plies]) if deep else ""))
    return case


def gen_mcts(m, out, positions, king_roots, seed=7):
    rng = random.Random(seed)
    cases = []
    mid = [p for p in positions[200:2500:173]]
    specs = [
        dict(name="start_pos64", fen=START, player="w", rr=0, playouts=[64], mode="pos", salt=0),
        dict(name="start_signed200", fen=START, player="w", rr=0, playouts=[200], mode="signed", salt=1),
        dict(name="start_3plies", fen=START, player="w", rr=0, playouts=[60, 60, 60], mode="pos", salt=2),
    ]
    for i, (fen, pl) in enumerate(mid[:8]):
        specs.append(dict(name="mid%d" % i, fen=fen, player=pl, rr=rng.randrange(0, 20), playouts=[100], mode="pos", salt=10 + i))
    for i, (fen, pl) in enumerate(mid[8:12]):
        specs.append(dict(name="mid_rr%d" % i, fen=fen, player=pl, rr=56 + i, playouts=[120, 40], mode="pos", salt=30 + i))
    for i, (fen, pl) in enumerate(king_roots[:6]):
        specs.append(dict(name="kingcap%d" % i, fen=fen, player=pl, rr=3, playouts=[150], mode="pos", salt=50 + i))
    for i, (fen, pl) in enumerate(king_roots[6:9]):
        specs.append(dict(name="kingcap_signed%d" % i, fen=fen, player=pl, rr=0, playouts=[150, 80], mode="signed", salt=70 + i))

    for sp in specs:
        cases.append(run_spec(m, sp))
    json.dump(dict(numpy=np.__version__, cases=cases), open(os.path.join(out, "mcts.json"), "w"), separators=(",", ":"))


def gen_mcts_deep(m, out):
    """mcts_deep.json: searches at the METRIC's depth (playout 1600, main.py:473-493 / README: train_playout) and searches
    whose selected paths run far below 32 levels (fakenet mode 'deep'; the 60-ply rule, main.py:415-416, ends the longest
    lines at level 60) — the long-path backups of the device search.  Positions come from the committed rules.npz."""
    g = np.load(os.path.join(out, "rules.npz"))

    def fen_of(i):
        b = g["boards"][i].reshape(10, 9)
        rows = []
        for y in range(10):
            row, run = "", 0
            for x in range(9):
                if b[y, x] == 0:
                    run += 1
                else:
                    row += (str(run) if run else "") + PIECES[b[y, x]]
                    run = 0
            rows.append(row + (str(run) if run else ""))
        return "/".join(rows), ("b" if g["side"][i] else "w")
    ok = [i for i in range(len(g["boards"])) if (g["boards"][i] == 1).any() and (g["boards"][i] == 8).any() and g["counts"][i] > 20]
    blacks = [i for i in ok if g["side"][i] == 1]
    whites = [i for i in ok if g["side"][i] == 0]
    fb, pb = fen_of(blacks[len(blacks) // 3])
    fw, pw = fen_of(whites[len(whites) // 2])
    fb2, pb2 = fen_of(blacks[2 * len(blacks) // 3])
    specs = [
        dict(name="p1600_start", fen=START, player="w", rr=0, playouts=[1600], mode="pos", salt=1600),
        dict(name="p1600_black", fen=fb, player=pb, rr=7, playouts=[1600], mode="pos", salt=1601),
        dict(name="p1600_rr52", fen=fw, player=pw, rr=52, playouts=[1600], mode="signed", salt=1602),
        dict(name="p1600_2plies", fen=fb2, player=pb2, rr=0, playouts=[1600, 800], mode="pos", salt=1603),
        dict(name="deep_start800", fen=START, player="w", rr=0, playouts=[800], mode="deep", salt=913),
        dict(name="deep_2plies", fen=START, player="w", rr=0, playouts=[400, 200], mode="deep", salt=915),
        dict(name="deep_black", fen=fb, player=pb, rr=0, playouts=[600], mode="deep", salt=904),
        dict(name="deep_rr40", fen=fw, player=pw, rr=40, playouts=[500], mode="deep", salt=912),
    ]
    cases = [run_spec(m, sp, deep=True) for sp in specs]
    json.dump(dict(numpy=np.__version__, cases=cases), open(os.path.join(out, "mcts_deep.json"), "w"), separators=(",", ":"))


SELFPLAY_CASES = [
    # (name, train_playout, fakenet mode, fakenet salt, np.random seed)
    dict(name="sp_pos30", playout=30, mode="pos", salt=101, seed=11),
    dict(name="sp_signed24", playout=24, mode="signed", salt=202, seed=12),
    dict(name="sp_pos12", playout=12, mode="pos", salt=303, seed=13),
    # GameBoard.reload patched to start at restrict_round 59: the 60-ply no-capture tie (main.py:1542-1545) is reached
    dict(name="sp_tie", playout=10, mode="pos", salt=404, seed=14, rr0=59),
    # a game at a few hundred playouts per move: visit counts in the hundreds, re-rooted subtrees that already hold most
    # of the next search's visits (the root starts each search with N well above 0), deeper trees carried across plies
    dict(name="sp_pos200", playout=200, mode="pos", salt=505, seed=15),
    dict(name="sp_signed160", playout=160, mode="signed", salt=606, seed=16),
]


def gen_selfplay(m, out):
    """cchess_main.selfplay() of the UNMODIFIED reference (main.py:1493-1554) with search_threads = 1, the exact-integer
    fake forward and a seeded np.random: per ply the canonical state, the root children with their visit counts, pi
    (float64, as returned: scattered over the 2086 canonical labels), the move np.random.choice picked, and z."""
    import tempfile

    class FakePV(object):   # stands in for policy_value_network(res_block_nums): only forward() is used by selfplay()
        def __init__(self, *a, **k):
            self.forward = None

        def train_step(self, *a, **k):
            raise RuntimeError("not used")

        def save(self, *a, **k):
            pass

    m.policy_value_network = FakePV
    cols = dict(case=[], state=[], side=[], played=[], count=[], labels=[], visits=[], z=[], pi_ptr=[0], pi_idx=[], pi_val=[])
    meta = []
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp()
    os.chdir(tmp)   # cchess_main.__init__ opens ./log_file.txt
    try:
        for ci, c in enumerate(SELFPLAY_CASES):
            log = []
            fwd = fakenet.make_forward(c["mode"], c["salt"], log)
            with rh.quiet():
                cm = m.cchess_main(c["playout"], 512, True, 1, "cpu", 1, 2)
            cm.policy_value_netowrk.forward = fwd
            cm.mcts.forward = fwd
            trace = []   # (player, [(label, N)], played label) captured where get_action hands the move to update_tree
            orig_update = cm.mcts.update_tree

            def update_tree(act, cm=cm, trace=trace, orig_update=orig_update):
                kids = [(m.label2i[a], int(n.N)) for a, n in cm.mcts.root.child.items()]
                trace.append((cm.game_borad.current_player, kids, m.label2i[act]))
                return orig_update(act)
            cm.mcts.update_tree = update_tree
            if c.get("rr0"):
                orig_reload = cm.game_borad.reload

                def reload(cm=cm, orig_reload=orig_reload, rr0=c["rr0"]):
                    orig_reload()
                    cm.game_borad.restrict_round = rr0
                cm.game_borad.reload = reload
            np.random.seed(c["seed"])
            with rh.quiet(), np.errstate(all="ignore"):
                play_data, n = cm.selfplay()
            play_data = list(play_data)
            assert n == len(play_data) == len(trace)
            for (state, prob, z), (player, kids, played) in zip(play_data, trace):
                cols["case"].append(ci)
                cols["state"].append(fen_to_board(state))   # canonical: try_flip'ed for black (main.py:1505)
                cols["side"].append(1 if player == "b" else 0)
                cols["played"].append(played)
                cols["count"].append(len(kids))
                lab = np.full(128, 0xFFFF, np.uint16)
                vis = np.zeros(128, np.int32)
                lab[:len(kids)] = [k for k, _ in kids]
                vis[:len(kids)] = [v for _, v in kids]
                cols["labels"].append(lab)
                cols["visits"].append(vis)
                cols["z"].append(float(z))
                nz = np.nonzero(prob)[0]
                cols["pi_idx"].extend(int(i) for i in nz)
                cols["pi_val"].extend(float(prob[i]) for i in nz)
                cols["pi_ptr"].append(len(cols["pi_idx"]))
            meta.append(dict(c, plies=n, evals=len(log), z_first=float(play_data[0][2]), z_values=sorted(set(float(d[2]) for d in play_data))))
            print("selfplay case %-12s plies=%d evals=%d z=%s" % (c["name"], n, len(log), meta[-1]["z_values"]))
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(out, "selfplay.npz"),
                        case=np.asarray(cols["case"], np.int32), state=np.stack(cols["state"]).astype(np.uint8),
                        side=np.asarray(cols["side"], np.uint8), played=np.asarray(cols["played"], np.uint16),
                        count=np.asarray(cols["count"], np.uint8), labels=np.stack(cols["labels"]), visits=np.stack(cols["visits"]),
                        z=np.asarray(cols["z"], np.float64), pi_ptr=np.asarray(cols["pi_ptr"], np.int64),
                        pi_idx=np.asarray(cols["pi_idx"], np.int32), pi_val=np.asarray(cols["pi_val"], np.float64),
                        meta=np.asarray(json.dumps(meta)))


def scripted_outputs(states, k, scale):
    """The scripted net of the policy_update fixture: (logits, value) as exact functions of (state content, number of
    train steps taken so far k, drift scale) — shared by this generator and tests/test_train.py."""
    s = np.asarray(states, np.float64).reshape(len(states), -1)
    key = s @ (np.arange(s.shape[1], dtype=np.float64) % 97 + 1.0)
    j = np.arange(2086, dtype=np.float64)
    base = ((key[:, None] * 0.37 + j[None, :] * 1.61) % 1.0) - 0.2          # raw logits, some negative (log -> nan, filtered)
    drift = ((key[:, None] * 0.11 + j[None, :] * 0.73) % 1.0) - 0.5
    logits = base + scale * k * drift
    v = np.tanh(((key * 0.013) % 2.0) - 1.0 + 0.05 * k)
    return logits.astype(np.float32), v.astype(np.float32).reshape(-1, 1)


POLICY_UPDATE_CASES = [
    # (name, drift scale of the scripted net per train step, lr_multiplier before)
    dict(name="kl_small_raises_lr", scale=1e-4, lrm=1.0),
    dict(name="kl_small_at_cap", scale=1e-4, lrm=10.5),
    dict(name="kl_mid_keeps_lr", scale=6e-4, lrm=1.0),
    dict(name="kl_large_lowers_lr", scale=1.6e-3, lrm=1.0),
    dict(name="kl_stops_at_epoch3", scale=4e-3, lrm=3.0),
    dict(name="kl_huge_stops_early", scale=0.6, lrm=2.25),
    dict(name="kl_huge_at_floor", scale=0.6, lrm=0.09),
]


def gen_policy_update(m, out):
    """cchess_main.policy_update of the UNMODIFIED reference (main.py:1157-1204) driven through a stand-in `self` with a
    scripted net: the KL estimate on raw logits, the early stop at 4 x kl_targ, the x / 1.5 learning-rate adaptation with
    its 0.1 / 10 bounds, and the explained-variance figures it logs."""
    import re

    class Log(object):
        def __init__(self):
            self.lines = []

        def write(self, x):
            self.lines.append(x)

        def flush(self):
            pass

    rng = np.random.RandomState(5)
    B = 24
    states = [(rng.rand(9, 10, 14) < 0.07).astype(np.float32) for _ in range(B)]
    pis = [rng.dirichlet(np.ones(2086) * 0.05) for _ in range(B)]
    zs = [float(rng.choice([-1.0, 0.0, 1.0])) for _ in range(B)]
    cases = []
    for c in POLICY_UPDATE_CASES:
        class Net(object):
            k = 0
            lrs = []

            def train_step(self, sb, pb, wb, lr):
                Net.k += 1
                Net.lrs.append(float(lr))
                return 0.5, 1.25, 100 + Net.k

            def save(self, step):
                Net.saved = int(step)
        Net.k, Net.lrs = 0, []

        class Mcts(object):
            def forward(self, sb):
                return scripted_outputs(sb, Net.k, c["scale"])
        fake = types.SimpleNamespace(data_buffer=list(zip(states, pis, zs)), batch_size=B, mcts=Mcts(), epochs=5,
                                     policy_value_netowrk=Net(), learning_rate=0.001, lr_multiplier=c["lrm"], kl_targ=0.025,
                                     log_file=Log(), global_step=0)
        random.seed(99)
        with rh.quiet(), np.errstate(all="ignore"):
            m.cchess_main.policy_update(fake)
        msg = fake.log_file.lines[-1]
        f = dict(re.findall(r"(kl|lr_multiplier|explained_var_old|explained_var_new):([-+0-9.einfa]+)", msg))
        cases.append(dict(c, steps=Net.k, lrs=Net.lrs, lr_multiplier_after=float(fake.lr_multiplier), saved_step=Net.saved,
                          global_step=int(fake.global_step), kl_logged=f["kl"], explained_var_old_logged=f["explained_var_old"],
                          explained_var_new_logged=f["explained_var_new"]))
        print("policy_update case %-22s steps=%d lrm %.4f -> %.4f  log: %s" % (c["name"], Net.k, c["lrm"], fake.lr_multiplier, msg.strip()))
    np.savez_compressed(os.path.join(out, "policy_update.npz"), states=np.stack(states).astype(np.uint8), pi=np.stack(pis), z=np.asarray(zs),
                        meta=np.asarray(json.dumps(cases)))


def main():
    m = rh.load_main()
    out = HERE
    if "--policy-update-only" in sys.argv:
        return gen_policy_update(m, out)
    if "--deep-only" in sys.argv:
        return gen_mcts_deep(m, out)
    if "--selfplay-only" not in sys.argv:
        gen_tables(m, out)
        positions, king_roots = gen_rules(m, out)
        gen_mcts(m, out, positions, king_roots)
    gen_selfplay(m, out)
    gen_mcts_deep(m, out)
    gen_policy_update(m, out)


if __name__ == "__main__":
    main()
