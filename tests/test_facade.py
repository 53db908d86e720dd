"""The reference-named façade (main.py / policy_value_network.py at the repo root): CLI surface on CPU,
behaviour on the GPU against golden vectors of the unmodified reference."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FLAGS = ["--mode", "--ai_count", "--ai_function", "--train_playout", "--batch_size", "--play_playout", "--delay",
             "--end_delay", "--search_threads", "--processor", "--num_gpus", "--res_block_nums", "--human_color"]


def test_cli_keeps_every_reference_flag():
    """main.py:1557-1575 of the reference."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "main.py"), "--help"], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stderr
    for f in REF_FLAGS:
        assert f in out.stdout, f
    assert "{train,play}" in out.stdout and "{mcts,net}" in out.stdout and "{cpu,gpu}" in out.stdout


def test_module_surface_cpu(tables_golden):
    sys.path.insert(0, ROOT)
    import importlib
    m = importlib.import_module("main")
    assert m.labels_len == 2086 and m.label2i["d7e8"] == tables_golden["label2i"]["d7e8"]
    assert m.flipped_uci_labels(["a0a1", "h7g9"]) == ["a9a8", "h2g0"]
    assert m.is_kill_move("RNBAKABNR/9/9/9/9/9/9/9/9/rnbakabnr", "RNBAKABNR/9/9/9/9/9/9/9/9/Rnbakabn1") == 1
    p = np.arange(2086, dtype=np.float32)
    assert np.array_equal(m.cchess_main.flip_policy(p), p[np.asarray(m.unflipped_index)])
    for name in ("GameBoard", "MCTS_tree", "leaf_node", "cchess_main", "softmax", "get_pieces_count", "create_uci_labels"):
        assert hasattr(m, name)
    import policy_value_network as pv
    import policy_value_network_gpus as pvg
    assert hasattr(pv.policy_value_network, "forward") and hasattr(pv.policy_value_network, "train_step") and hasattr(pv.policy_value_network, "save")
    assert issubclass(pvg.policy_value_network_gpus, pv.policy_value_network)


@pytest.mark.gpu
def test_gameboard_and_tree_match_reference(tables_golden, mcts_golden, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    sys.path.insert(0, ROOT)
    import main as M
    import fakenet
    start = M.GameBoard().state
    assert M.GameBoard.get_legal_moves(start, "w") == tables_golden["start_moves"]
    assert M.GameBoard.sim_do_action("a0a1", start) == "1NBAKABNR/R8/1C5C1/P1P1P1P1P/9/9/p1p1p1p1p/1c5c1/9/rnbakabnr"
    case = [c for c in mcts_golden["cases"] if c["name"] == "start_3plies"][0]
    t = M.MCTS_tree(start, fakenet.make_forward(case["mode"], case["salt"]), 1)
    assert t.generate_inputs(start, "w").sum() == 26.0
    state, player, rr = start, "w", 0
    for ply in case["plies"]:
        assert (state, player, rr) == (ply["state"], ply["player"], ply["rr"])
        t.main(state, player, rr, ply["playouts"])
        got = [(M.label2i[a], n.N, int(np.float32(n.W).view(np.uint32)), int(np.float32(n.Q).view(np.uint32)), int(np.float32(n.P).view(np.uint32)))
               for a, n in t.root.child.items()]
        assert got == [tuple(x) for x in ply["root"]]
        best = M.labels_array[ply["played"]]
        nxt = M.GameBoard.sim_do_action(best, state)
        rr = rr + 1 if M.is_kill_move(state, nxt) == 0 else 0
        assert abs(t.Q(best) - t.root.child[best].Q) == 0
        t.update_tree(best)
        state, player = nxt, ("b" if player == "w" else "w")


@pytest.mark.gpu
def test_tree_with_search_threads_16(tmp_path, monkeypatch):
    """MCTS_tree(search_threads=16): exactly `playouts` simulations, visit counts add up, best move plausible."""
    monkeypatch.chdir(tmp_path)
    sys.path.insert(0, ROOT)
    import main as M
    import fakenet
    start = M.GameBoard().state
    t = M.MCTS_tree(start, fakenet.make_forward("pos", 0), 16)
    t.main(start, "w", 0, 200)
    ch = t.root.child
    assert len(ch) == 44 and sum(n.N for n in ch.values()) == 200
    assert all(n.N >= 0 and abs(n.Q) <= 1 for n in ch.values())
    best = max(ch.items(), key=lambda kv: kv[1].N)[0]
    t.update_tree(best)
    nxt = M.GameBoard.sim_do_action(best, start)
    t.main(nxt, "b", 1, 64)
    assert sum(n.N for n in t.root.child.values()) >= 64


@pytest.mark.gpu
def test_cchess_main_selfplay_and_update(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    sys.path.insert(0, ROOT)
    import main as M
    np.random.seed(0)
    cm = M.cchess_main(playout=6, in_batch_size=8, exploration=True, in_search_threads=16, processor="gpu", num_gpus=1,
                       res_block_nums=2, human_color="b", games=16)
    # one move through the single-tree surface
    act, move_probs, win_rate = cm.get_action(cm.game_borad.state, cm.temperature)
    assert act in M.GameBoard.get_legal_moves(cm.game_borad.state, "w")
    assert abs(float(np.sum(move_probs[0][1])) - 1.0) < 1e-6 and len(move_probs[0][0]) == 44
    cm.mcts.reload()
    # forward signature (policy_value_network.py:202-214)
    lg, v = cm.policy_value_netowrk.forward(np.zeros((3, 9, 10, 14), np.float32))
    assert lg.shape == (3, 2086) and v.shape == (3, 1) and lg.dtype == np.float32
    # batched self-play -> packed records -> dense tuples -> one policy update
    rec = cm.selfplay_batch(games=16, max_plies=4)
    from cchess_zero_amd.selfplay import to_dense, unpack_records
    # unfinished games are not reported; force-finish bookkeeping by playing until some end is not needed here:
    sp_rec = rec
    assert sp_rec.shape[1] > 0
    # build a small buffer from a longer run of 4 games with few playouts
    cm.playout_counts = 2
    rec = cm.selfplay_batch(games=4, max_plies=None)
    planes, pi, z = to_dense(rec)
    assert len(z) > 0 and planes.shape[1:] == (9, 10, 14) and pi.shape[1] == 2086
    assert np.allclose(pi.sum(axis=1), 1.0, atol=2e-2) and set(np.unique(z)) <= {-1.0, 0.0, 1.0}
    cm.data_buffer.extend(zip(planes, pi, z))
    step0 = cm.global_step
    cm.policy_update()
    assert cm.global_step > step0
    assert os.path.exists(os.path.join("gpu_models", "best_model.ckpt-%d.pt" % cm.global_step))
    # a fresh network auto-restores the newest checkpoint (policy_value_network.py:164-174)
    from policy_value_network_gpus import policy_value_network_gpus
    n2 = policy_value_network_gpus(1, 2)
    assert n2.global_step == cm.global_step
    # one iteration of the training loop itself (main.py:1206-1248): continuous self-play on 8 slots until 8 games have
    # finished, records -> buffer (resized to hold two batches, shuffled), policy updates proportional to the new samples
    cm.games = 8
    step1 = cm.global_step
    cm.data_buffer.clear()          # run() keeps PACKED records in the buffer (the dense tuples above are the reference's form)
    cm.run(max_batches=1)
    assert all(isinstance(r, np.ndarray) and r.dtype == np.uint8 and r.shape == (608,) for r in list(cm.data_buffer)[:5])
    st = cm.last_selfplay_stats
    assert st["games"] >= 8 and st["stalled"] == 0 and st["dropped"] == 0
    assert len(cm.data_buffer) >= st["plies"] > 0 and cm.data_buffer.maxlen >= 2 * st["plies"]
    # asynchronous plies: every recorded ply had its full search; at most (terminal_extra + 1) simulations per slot and step
    assert cm.global_step > step1 and st["plies"] * cm.playout_counts <= st["sims"] <= st["lock_steps"] * 8 * 5
    # the slots live across batches: the games in progress at the end of a batch go on in the next one, so long
    # games reach the buffer too; the second batch's statistics are deltas and the totals add up
    sp1, tot1 = cm._sp, cm._sp.stats()
    in_progress = int(cm._sp.active().sum())
    assert in_progress == 8
    cm.run(max_batches=1)
    st2, tot2 = cm.last_selfplay_stats, cm._sp.stats()
    assert cm._sp is sp1 and st2["games"] >= 8 and tot2["games"] == tot1["games"] + st2["games"]
    assert tot2["plies"] == tot1["plies"] + st2["plies"] and st2["dropped"] == 0
