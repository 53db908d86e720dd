"""Pins the C oracle (oracle/cchess_oracle.c) against golden vectors produced by the
UNMODIFIED reference (tests/golden/gen_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest

from oracle import oracle as O
import searchdrive


def test_tables(tables_golden):
    L = O.labels()
    assert len(L) == tables_golden["n_labels"] == 2086
    assert hashlib.sha256("\n".join(L).encode()).hexdigest() == tables_golden["labels_sha256"]
    uf = O.unflip()
    assert hashlib.sha256(uf.astype(np.int16).tobytes()).hexdigest() == tables_golden["unflip_sha256"]
    assert all(uf[uf[i]] == i for i in range(2086)) and not np.any(uf == np.arange(2086))
    for k, v in tables_golden["label2i"].items():
        assert L.index(k) == v
    lut = O.lut()
    sd = O.label_srcdst()
    for i in range(2086):
        assert lut[sd[i] & 0xFF, sd[i] >> 8] == i
    assert (lut >= 0).sum() == 2086


def test_start_position_and_perft(tables_golden):
    b = O.fen_to_board(O.START_FEN)
    assert O.board_to_fen(b) == O.START_FEN
    L = O.labels()
    assert [L[i] for i in O.legal_moves(b, 0)] == tables_golden["start_moves"]

    def perft(board, side, depth):
        mv = O.legal_moves(board, side)
        if depth == 1:
            return len(mv)
        return sum(perft(O.apply_move(board, m)[0], side ^ 1, depth - 1) for m in mv)
    for d in ("1", "2", "3"):
        assert perft(b, 0, int(d)) == tables_golden["perft"][d]


def test_movegen_ordered(rules_golden):
    g = rules_golden
    assert int(g["chessman_mismatches"]) == 0 and int(g["chessman_checked"]) == len(g["boards"])
    for i in range(len(g["boards"])):
        mv = O.legal_moves(g["boards"][i], int(g["side"][i]))
        n = int(g["counts"][i])
        assert len(mv) == n, i
        assert np.array_equal(mv, g["moves"][i, :n]), i


def test_apply_move_and_kill(rules_golden):
    g = rules_golden
    for i in range(len(g["boards"])):
        if g["chosen"][i] == 0xFFFF:
            continue
        nb, cap, term = O.apply_move(g["boards"][i], int(g["chosen"][i]))
        assert np.array_equal(nb, g["next_boards"][i]), i
        assert (1 if cap else 0) == int(g["kill"][i]), i
        fen = O.board_to_fen(nb)
        assert np.array_equal(O.fen_to_board(fen), nb)
        assert bool(term & 1) == ("K" not in fen) and bool(term & 2) == ("k" not in fen)


def test_planes_quirk_q1(rules_golden):
    g = rules_golden
    for i in range(len(g["boards"])):
        p = O.encode_planes(g["boards"][i], int(g["side"][i]), True)
        assert np.array_equal(np.packbits(p.reshape(-1) > 0.5), g["planes_bits"][i]), i
    assert O.encode_planes(O.fen_to_board(O.START_FEN), 0).sum() == 26.0  # not 32: quirk Q1


def test_zobrist_consistency(rules_golden):
    g = rules_golden
    seen = {}
    for i in range(0, len(g["boards"]), 3):
        b, s = g["boards"][i], int(g["side"][i])
        h = O.zhash(b, s)
        key = (b.tobytes(), s)
        if h in seen:
            assert seen[h] == key, "zobrist collision in corpus"
        seen[h] = key
        if g["chosen"][i] != 0xFFFF:
            # incremental update == from-scratch
            sd = O.label_srcdst()[int(g["chosen"][i])]
            src, dst = int(sd & 0xFF), int(sd >> 8)
            tab, sk = O.zobrist_table()
            h2 = h ^ int(tab[b[src], src]) ^ int(tab[b[src], dst]) ^ sk
            if b[dst]:
                h2 ^= int(tab[b[dst], dst])
            assert h2 == O.zhash(g["next_boards"][i], s ^ 1)


class _OracleEngine:
    def __init__(self, G, cap=200000):
        self.s = O.Search(G, cap)

    def reset(self, boards, side, rr):
        self.s.reset(boards, side, rr)
        self._parked = np.zeros(len(side), bool)

    def select(self, mode, mask=None):
        # trees outside `mask` must idle this step: the oracle has no mask argument, so run
        # all and revert is impossible -> emulate by per-tree engines instead (see below).
        raise NotImplementedError


class _PerTreeOracle:
    """One czo_search per tree so that trees can idle independently (different playout counts)."""

    def __init__(self, G, cap=200000):
        self.G = G
        self.e = [O.Search(1, cap) for _ in range(G)]

    def reset(self, boards, side, rr):
        for g in range(self.G):
            self.e[g].reset(boards[g:g + 1], side[g:g + 1], rr[g:g + 1])

    def select(self, mode, mask=None):
        planes = np.zeros((self.G, 9, 10, 14), np.float32)
        need = np.zeros(self.G, np.uint8)
        self._ran = np.zeros(self.G, bool)
        for g in range(self.G):
            if mask is not None and not mask[g]:
                continue
            p, n = self.e[g].select(mode)
            planes[g], need[g] = p[0], n[0]
            self._ran[g] = True
        return planes, need

    def expand_backup(self, logits, value):
        for g in range(self.G):
            if self._ran[g]:
                self.e[g].expand_backup(logits[g:g + 1], value[g:g + 1])

    def root_stats(self):
        st = [e.root_stats() for e in self.e]
        return {k: np.concatenate([s[k] for s in st]) for k in st[0]}

    def advance(self, played):
        for g in range(self.G):
            self.e[g].advance(played[g:g + 1])

    def tree_dump(self, g):
        return self.e[g].tree_dump(0)

    def status(self):
        return np.concatenate([e.status()[0] for e in self.e])


def test_search_matches_reference_at_depth(mcts_deep_golden):
    """The oracle against the unmodified reference at the METRIC's depth (playout 1600, main.py:473-493; one black-to-move
    root, one at restrict_round 52, one across update_tree) and on searches whose selected paths are 33-62 levels long
    (fakenet mode 'deep'): root children, whole-tree digests, evaluated positions, deepest visited level."""
    cases = mcts_deep_golden["cases"]
    assert max(p["playouts"] for c in cases for p in c["plies"]) == 1600
    assert max(p["max_level"] for c in cases for p in c["plies"]) > 48 and sum(p["max_level"] >= 32 for c in cases for p in c["plies"]) >= 4
    eng = _PerTreeOracle(len(cases))
    results, logs = searchdrive.run_cases(eng, cases)
    assert not np.any(eng.status() & ~8)
    searchdrive.check_against_golden(results, logs, cases)


def test_search_matches_reference(mcts_golden):
    cases = mcts_golden["cases"]
    eng = _PerTreeOracle(len(cases))
    results, logs = searchdrive.run_cases(eng, cases)
    assert not np.any(eng.status() & ~8)
    searchdrive.check_against_golden(results, logs, cases)
