"""SURVEY §8 rows f2 / f3 on the GPU: the `--mode play` engine surface (main.py:1278-1329,1380-1491 of the reference:
select_move / human_move / get_hint / check_end with --ai_function {mcts,net} and --human_color coordinate flipping,
ChessGame.py:183-195 feeds them board coordinates) and checkpoint I/O (policy_value_network.py:164-184): torch
checkpoints with momentum slots, and the import of the reference's TF1 variables by name."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = "abcdefghi"


def _main(tmp_path, monkeypatch, **kw):
    monkeypatch.chdir(tmp_path)
    sys.path.insert(0, ROOT)
    import main as M
    args = dict(playout=12, in_batch_size=8, exploration=False, in_search_threads=1, processor="cpu", num_gpus=1,
                res_block_nums=1, human_color="b")
    args.update(kw)
    return M, M.cchess_main(**args)


def _expected_net_move(cm, state, player):
    """select_move's `net` branch restated with the oracle's rules (main.py:1437-1461)."""
    from oracle import oracle as O
    board, side = O.fen_to_board(state), (1 if player == "b" else 0)
    logits, value = cm.policy_value_netowrk.forward(O.encode_planes(board, side)[None])
    p = logits.flatten()
    if side:
        p = p[O.unflip().astype(np.int64)]
    moves = O.legal_moves(board, side)
    tot = 1e-8
    pr = []
    for l in moves:
        pr.append(p[l])
        tot += p[l]
    pr = [x / tot for x in pr]
    best = int(np.argmax(pr))   # max() keeps the first maximum
    return O.labels()[int(moves[best])], float(value[0, 0]), {O.labels()[int(l)]: x for l, x in zip(moves, pr)}


def _flip(a):
    return "".join(str(9 - int(c)) if c.isdigit() else c for c in a)


@pytest.mark.parametrize("human_color", ["b", "w"])
def test_select_move_net_branch_and_coordinate_flipping(tmp_path, monkeypatch, human_color):
    M, cm = _main(tmp_path, monkeypatch, human_color=human_color)
    for ply in range(4):   # two moves of each colour: the black ones go through flip_policy
        state, player, rnd, rr = cm.game_borad.state, cm.game_borad.current_player, cm.game_borad.round, cm.game_borad.restrict_round
        want, want_v, _ = _expected_net_move(cm, state, player)
        (sx, sy, dx, dy), win_rate = cm.select_move("net")
        shown = _flip(want) if human_color == "w" else want       # main.py:1475-1476
        assert (sx, sy, dx, dy) == (FILES.index(shown[0]), int(shown[1]), FILES.index(shown[2]) - FILES.index(shown[0]), int(shown[3]) - int(shown[1]))
        assert abs(win_rate - want_v) < 1e-6
        nxt = M.GameBoard.sim_do_action(want, state)
        assert cm.game_borad.state == nxt and cm.game_borad.round == rnd + 1
        assert cm.game_borad.current_player == ("b" if player == "w" else "w")
        assert cm.game_borad.restrict_round == (rr + 1 if M.is_kill_move(state, nxt) == 0 else 0)


def test_select_move_mcts_branch_and_human_move(tmp_path, monkeypatch):
    from oracle import oracle as O
    M, cm = _main(tmp_path, monkeypatch, human_color="w")
    start = cm.game_borad.state
    (sx, sy, dx, dy), win_rate = cm.select_move("mcts")              # AI (red) moves first
    shown = FILES[sx] + str(sy) + FILES[sx + dx] + str(sy + dy)
    action = _flip(shown)                                            # human_color 'w': coordinates are shown rank-flipped
    assert action in [O.labels()[int(l)] for l in O.legal_moves(O.fen_to_board(start), 0)]
    assert cm.game_borad.state == M.GameBoard.sim_do_action(action, start) and cm.game_borad.current_player == "b"
    assert -1.0 <= win_rate <= 1.0
    # the tree was re-rooted on the played move (update_tree inside get_action)
    assert cm.mcts._state == cm.game_borad.state
    # human reply given in (flipped) board coordinates, mcts branch: the tree follows (main.py:1403-1412)
    state = cm.game_borad.state
    reply = O.labels()[int(O.legal_moves(O.fen_to_board(state), 1)[3])]
    shown = _flip(reply)
    coord = (FILES.index(shown[0]), int(shown[1]), FILES.index(shown[2]), int(shown[3]))
    wr = cm.human_move(coord, "mcts")
    assert cm.game_borad.state == M.GameBoard.sim_do_action(reply, state) and cm.game_borad.current_player == "w"
    assert cm.mcts._state == cm.game_borad.state and -1.0 <= wr <= 1.0
    # and the next AI move searches from the position after the human move
    cm.select_move("mcts")
    assert cm.game_borad.round == 4


@pytest.mark.parametrize("human_color", ["b", "w"])
def test_get_hint_both_branches(tmp_path, monkeypatch, human_color):
    M, cm = _main(tmp_path, monkeypatch, human_color=human_color)
    called = []
    hint = cm.get_hint("mcts", True, lambda: called.append(1))
    assert called == [1]                                             # the root had no children yet: message handler fired
    moves = M.GameBoard.get_legal_moves(cm.game_borad.state, "w")
    keys = [k for k, _ in hint]
    assert sorted(keys) == sorted(_flip(a) if human_color == "w" else a for a in moves)
    probs = [p for _, p in hint]
    assert probs == sorted(probs, reverse=True) and abs(sum(probs) - 1.0) < 1e-9
    # visits -> softmax(log N / T): proportional to the visit counts at temperature 1
    ch = cm.mcts.root.child
    tot = sum(n.N for n in ch.values())
    for k, p in hint:
        a = _flip(k) if human_color == "w" else k
        assert abs(p - ch[a].N / tot) < 1e-12
    # net branch: normalised raw-logit priors of the legal moves (main.py:1300-1324), ascending when reverse=False
    _, _, want = _expected_net_move(cm, cm.game_borad.state, "w")
    hint = cm.get_hint("net", False, lambda: None)
    assert [p for _, p in hint] == sorted(p for _, p in hint)
    for k, p in hint:
        a = _flip(k) if human_color == "w" else k
        assert abs(p - want[a]) <= 1e-6 * max(1.0, abs(want[a]))


def test_check_end(tmp_path, monkeypatch, capsys):
    M, cm = _main(tmp_path, monkeypatch)
    assert cm.check_end() == (False, "")
    cm.game_borad.state = "RNBA1ABNR/9/1C5C1/P1P1P1P1P/9/9/p1p1p1p1p/1c5c1/9/rnbakabnr"
    assert cm.check_end() == (True, "b")
    cm.game_borad.state = "RNBAKABNR/9/1C5C1/P1P1P1P1P/9/9/p1p1p1p1p/1c5c1/9/rnba1abnr"
    assert cm.check_end() == (True, "w")
    cm.game_borad.reload()
    cm.game_borad.restrict_round = 60
    assert cm.check_end() == (True, "t")


@pytest.mark.parametrize("ai_function", ["mcts", "net"])
def test_play_headless_ai_vs_ai_terminates(tmp_path, monkeypatch, ai_function):
    """`python main.py --mode play --ai_count 2` without the GUI: a scripted AI-vs-AI game to its end."""
    import types
    monkeypatch.chdir(tmp_path)
    sys.path.insert(0, ROOT)
    import main as M
    args = types.SimpleNamespace(play_playout=3, batch_size=8, search_threads=1, processor="cpu", num_gpus=1, res_block_nums=1,
                                 human_color="b", ai_count=2, ai_function=ai_function, delay=0)
    if ai_function == "net":
        # the arg-max net player repeats moves: the 60-ply no-capture rule ends the game
        who = M._play_headless(args)
        assert who in ("w", "b", "t")
    else:
        np.random.seed(3)
        who = M._play_headless(args)
        assert who in ("w", "b", "t")


def test_checkpoint_roundtrip_with_momentum_and_tf_variable_import(tmp_path, monkeypatch):
    import torch
    monkeypatch.chdir(tmp_path)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import nethelpers as H
    from policy_value_network import policy_value_network
    rng = np.random.default_rng(0)
    a = policy_value_network(2, save_dir=str(tmp_path / "m1"), seed=1)
    x = H.positions(16, 5)
    pi = rng.random((16, 2086)).astype(np.float32)
    pi /= pi.sum(axis=1, keepdims=True)
    z = rng.choice([-1.0, 1.0], size=(16, 1)).astype(np.float32)
    a.train_step(x, pi, z, 0.05)
    a.train_step(x, pi, z, 0.05)
    path = a.save(a.global_step)
    la, va = a.forward(x)
    # a fresh network restores weights, step AND the momentum slots: one more identical step gives identical weights
    b = policy_value_network(2, save_dir=str(tmp_path / "m1"), seed=9)
    assert b.global_step == 2
    lb, vb = b.forward(x)
    assert np.array_equal(la, lb) and np.array_equal(va, vb)
    a.train_step(x, pi, z, 0.05)
    b.train_step(x, pi, z, 0.05)
    for p, q in zip(a.module.parameters(), b.module.parameters()):   # without the slots the step would differ by ~lr * 0.9 * |accum| ~ 1e-3
        assert torch.allclose(p, q, rtol=0, atol=1e-5)
    # the reference's TF1 variables by name -> a fresh network (different seed): same function
    d = a.export_tf_variables(str(tmp_path / "tf_vars.npz"))
    assert "conv2d/kernel" in d and "BatchNorm_6/moving_variance" in d and "fully_connected_2/biases" in d and d["conv2d_1/kernel"].shape == (3, 3, 128, 128)
    c = policy_value_network(2, save_dir=str(tmp_path / "m2"), seed=5)
    c.restore(str(tmp_path / "tf_vars.npz"))
    lc, vc = c.forward(x)
    la, va = a.forward(x)
    assert np.array_equal(la, lc) and np.array_equal(va, vc) and c.global_step == a.global_step
