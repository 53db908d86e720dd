"""Rows a17 / a18 of SURVEY §8 against the UNMODIFIED reference: cchess_main.get_action (main.py:1332-1358) and
cchess_main.selfplay (main.py:1493-1554).

tests/golden/selfplay.npz holds whole self-play games the reference played here (search_threads = 1, the exact-integer
fake forward of tests/fakenet.py, seeded np.random): per ply the canonical state, the root children and their visit
counts, pi[2086] as float64, the move np.random.choice picked, z.

  CPU  the C oracle, driven through the same get_action / selfplay logic with the same np.random seed, replays every game
       move for move (this pins the oracle's re-rooting across a whole game and the sampling arithmetic);
  GPU  (a) the façade's single-game cchess_main.selfplay() reproduces the reference's return value exactly — states, pi
       bit for bit, z — from the same seed;
       (b) the batched device-resident loop (SelfPlay: cz_selfplay_choose / advance / adjudicate / flush) with the sampled
       moves forced to the golden ones writes records whose dense expansion (to_dense) equals the reference's tuples
       exactly, for all games at once, including the game-end tests and z.
"""
import json
import os
import sys

import numpy as np
import pytest

import fakenet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
START_FEN = "RNBAKABNR/9/1C5C1/P1P1P1P1P/9/9/p1p1p1p1p/1c5c1/9/rnbakabnr"


def _cases():
    g = np.load(os.path.join(ROOT, "tests", "golden", "selfplay.npz"))
    meta = json.loads(str(g["meta"]))
    out = []
    for ci, m in enumerate(meta):
        rows = np.nonzero(g["case"] == ci)[0]
        pi = np.zeros((len(rows), 2086), np.float64)
        for j, r in enumerate(rows):
            lo, hi = int(g["pi_ptr"][r]), int(g["pi_ptr"][r + 1])
            pi[j, g["pi_idx"][lo:hi]] = g["pi_val"][lo:hi]
        out.append(dict(meta=m, state=g["state"][rows], side=g["side"][rows], played=g["played"][rows], count=g["count"][rows],
                        labels=g["labels"][rows], visits=g["visits"][rows], z=g["z"][rows], pi=pi))
    return out


def _softmax(x):   # main.py:1111-1116
    probs = np.exp(x - np.max(x))
    probs /= np.sum(probs)
    return probs


def _canon(board, side):
    from cchess_zero_amd.selfplay import canonical_boards
    return canonical_boards(board[None], np.array([side]))[0]


def test_oracle_replays_reference_selfplay_games():
    from oracle import oracle as O
    unflip = O.unflip().astype(np.int64)
    for c in _cases():
        m = c["meta"]
        fwd = fakenet.make_forward(m["mode"], m["salt"])
        s = O.Search(1, 50 * (m["playout"] + 1) * (m["plies"] + 2))   # the oracle never compacts: room for every expansion of the game
        s.reset(O.fen_to_board(START_FEN)[None], np.zeros(1, np.uint8), np.array([m.get("rr0", 0)], np.int32))
        np.random.seed(m["seed"])
        players, ply = [], 0
        while True:
            for step in range(m["playout"] + 1):   # MCTS_tree.main, main.py:473-493
                planes, need = s.select(0 if step == 0 else 1)
                lg, v = fwd(planes)
                s.expand_backup(lg, v)
            st = s.root_stats()
            k = int(st["count"][0])
            labels, visits = st["label"][0, :k], tuple(int(x) for x in st["N"][0, :k])
            assert k == c["count"][ply] and np.array_equal(labels, c["labels"][ply, :k]), (m["name"], ply)
            assert np.array_equal(np.asarray(visits), c["visits"][ply, :k]), (m["name"], ply)
            with np.errstate(divide="ignore"):
                probs = _softmax(1.0 / 1 * np.log(visits))            # main.py:1341, temperature = 1
            pick = np.random.choice(k, p=0.75 * probs + 0.25 * np.random.dirichlet(0.3 * np.ones(k)))   # main.py:1346
            assert int(labels[pick]) == int(c["played"][ply]), (m["name"], ply)
            b, sd, rr = s.root_state()
            assert np.array_equal(_canon(b[0], int(sd[0])), c["state"][ply]) and int(sd[0]) == int(c["side"][ply])
            lab = labels.astype(np.int64)
            if sd[0]:
                lab = unflip[lab]                                         # main.py:1507-1512
            dense = np.zeros(2086)
            dense[lab] = probs
            assert np.array_equal(dense, c["pi"][ply]), (m["name"], ply)
            players.append(int(sd[0]))
            s.advance(np.array([labels[pick]], np.uint16))
            ply += 1
            b, sd, rr = s.root_state()
            K, kk = (b[0] == 1).any(), (b[0] == 8).any()
            if not K or not kk:                                           # main.py:1532-1541
                winner = 1 if not K else 0
                z = np.where(np.array(players) == winner, 1.0, -1.0)
                break
            if rr[0] >= 60:                                               # main.py:1542-1545
                z = np.zeros(len(players))
                break
        assert ply == m["plies"] and np.array_equal(z, c["z"]), m["name"]


@pytest.mark.gpu
def test_facade_selfplay_returns_the_reference_tuples(tmp_path, monkeypatch):
    """cchess_main.selfplay() (single-game surface, every operation a HIP kernel with G = 1) from the same np.random seed:
    identical (state, pi, z) tuples to the reference's, pi bit for bit."""
    import types
    monkeypatch.chdir(tmp_path)

    class FakePV(object):
        def __init__(self, *a, **k):
            self.forward = None

        def save(self, *a, **k):
            pass
    for name in ("policy_value_network", "policy_value_network_gpus"):
        mod = types.ModuleType(name)
        setattr(mod, name, FakePV)
        monkeypatch.setitem(sys.modules, name, mod)
    import main as M
    from oracle import oracle as O
    for c in _cases():
        m = c["meta"]
        fwd = fakenet.make_forward(m["mode"], m["salt"])
        cm = M.cchess_main(m["playout"], 512, True, 1, "cpu", 1, 2)
        cm.policy_value_netowrk.forward = fwd
        cm.mcts.forward = fwd
        if m.get("rr0"):
            orig = cm.game_borad.reload

            def reload(cm=cm, orig=orig, rr0=m["rr0"]):
                orig()
                cm.game_borad.restrict_round = rr0
            cm.game_borad.reload = reload
        np.random.seed(m["seed"])
        data, n = cm.selfplay()
        data = list(data)
        assert n == m["plies"] == len(data), m["name"]
        for ply, (state, prob, z) in enumerate(data):
            assert np.array_equal(O.fen_to_board(state), c["state"][ply]), (m["name"], ply)
            assert prob.dtype == np.float64 and np.array_equal(prob, c["pi"][ply]), (m["name"], ply)
            assert float(z) == c["z"][ply], (m["name"], ply)


@pytest.mark.gpu
def test_batched_device_selfplay_records_equal_the_reference_tuples():
    """All golden games at once through the device-resident loop, sampled moves forced to the golden ones: the packed
    records (visit counts, labels, boards, z) and their dense expansion equal the reference's tuples exactly; finished
    games are adjudicated (king capture / 60-ply tie) on the device and parked."""
    import torch
    from cchess_zero_amd.engine import SearchEngine
    from cchess_zero_amd.selfplay import SelfPlay, canonical_boards, to_dense, unpack_records
    from oracle import oracle as O
    cases = _cases()
    G = len(cases)
    playouts = [c["meta"]["playout"] for c in cases]
    fwds = [fakenet.make_forward(c["meta"]["mode"], c["meta"]["salt"]) for c in cases]

    def forward(planes):   # per-game fake nets (each golden game has its own salt)
        p = planes.float().cpu().numpy()
        lg = np.zeros((G, 2086), np.float32)
        v = np.zeros((G, 1), np.float32)
        for g in range(G):
            lg[g:g + 1], v[g:g + 1] = fwds[g](p[g:g + 1])
        return torch.from_numpy(lg).cuda(), torch.from_numpy(v).cuda()

    # the games use different playout counts: search in lock-step up to the largest, masking trees that are done
    eng = SearchEngine(G, 60000, plane_dtype=torch.float32, channels=14)
    sp = SelfPlay(eng, None, max(playouts), exploration=True, temperature=1.0, seed=1, max_plies=512, continuous=False)
    b0 = np.tile(O.fen_to_board(START_FEN), (G, 1))
    sp.start(b0, np.zeros(G, np.uint8), np.array([c["meta"].get("rr0", 0) for c in cases], np.int32))

    def search(fwd, n, active=None):   # SearchEngine.search with a per-tree playout budget
        alive = sp.active().numpy().astype(bool)
        eng.step(fwd, mode=0, active=alive.astype(np.uint8))
        for i in range(n):
            eng.step(fwd, mode=1, active=(alive & (np.array(playouts) > i)).astype(np.uint8))
    eng.search = search
    max_plies = max(c["meta"]["plies"] for c in cases)
    for ply in range(max_plies):
        forced = np.array([c["played"][ply] if ply < c["meta"]["plies"] else 0xFFFF for c in cases], np.uint16)
        sp.step_ply(forward, forced=forced)
    rec = sp.drain()
    st = sp.stats()
    assert not bool(sp.active().any()) and st["games"] == G and st["stalled"] == 0 and st["dropped"] == 0
    u = unpack_records(rec)
    assert len(rec) == sum(c["meta"]["plies"] for c in cases) == st["plies"]
    planes, pi, z = to_dense(rec, 1.0, exact=True)
    starts = list(np.nonzero(u["ply"] == 0)[0]) + [len(rec)]
    seen = set()
    for a, b in zip(starts[:-1], starts[1:]):
        match = [i for i, c in enumerate(cases) if c["meta"]["plies"] == b - a and i not in seen
                 and np.array_equal(u["visits"][a:b].astype(np.int32), c["visits"])]
        assert match, "a finished game matches no golden game"
        c = cases[match[0]]
        seen.add(match[0])
        assert np.array_equal(u["ply"][a:b], np.arange(b - a))
        assert np.array_equal(u["labels"][a:b], c["labels"]) and np.array_equal(u["counts"][a:b], c["count"])
        assert np.array_equal(u["side"][a:b], c["side"])
        assert np.array_equal(canonical_boards(u["boards"][a:b], u["side"][a:b]), c["state"])
        assert np.array_equal(pi[a:b], c["pi"])                      # float64, bit for bit
        assert np.array_equal(z[a:b].astype(np.float64), c["z"])
        for j in range(a, b, 7):                                     # planes = state_to_positions(canonical state)
            assert np.array_equal(planes[j], O.encode_planes(u["boards"][j], int(u["side"][j])))
    assert len(seen) == G
    wins = sum(1 for c in cases if c["z"][0] != 0)
    assert st["draws"] == G - wins and st["red_wins"] + st["black_wins"] == wins
