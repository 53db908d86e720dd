"""cz_maskgen.h (the mask-only move generator cz_movegen runs when moves == NULL: one lane = one position, bit sets in
registers) compiled for the HOST and held, on the CPU, to the golden move lists of the unmodified reference (4 381 positions,
tests/golden/rules.npz) and to the C oracle on seeded random playouts — the same function the GPU kernel k_movegen_mask
inlines, so its rules are pinned without a GPU; tests/test_hip_rules.py then pins the kernel around it."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("maskgen") / "libmaskgen_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "maskgen_host.cpp")])
    lib = C.CDLL(so)
    from oracle import oracle as O
    lut = np.ascontiguousarray(O.lut(), np.int16)
    tab = (C.c_uint8 * lib.czm_host_sizeof_tables())()
    lib.czm_host_tables(lut.ctypes.data_as(C.c_void_p), tab)

    def masks(boards, side):
        boards = np.ascontiguousarray(boards, np.uint8).reshape(-1, 90)
        side = np.ascontiguousarray(side, np.uint8)
        n = len(boards)
        m = np.zeros((n, 66), np.uint32)
        c = np.zeros(n, np.int32)
        lib.czm_host_masks(tab, boards.ctypes.data_as(C.c_void_p), side.ctypes.data_as(C.c_void_p), n, m.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p))
        return m, c

    def lists(boards, side):
        boards = np.ascontiguousarray(boards, np.uint8).reshape(-1, 90)
        side = np.ascontiguousarray(side, np.uint8)
        n = len(boards)
        mv = np.zeros((n, 128), np.uint16)
        c = np.zeros(n, np.int32)
        m = np.zeros((n, 66), np.uint32)
        lib.czm_host_lists(tab, boards.ctypes.data_as(C.c_void_p), side.ctypes.data_as(C.c_void_p), n, mv.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p),
                           m.ctypes.data_as(C.c_void_p))
        lists.last_mask = m
        return mv, c
    masks.lists = lists
    return masks


def _mask_of(moves):
    m = np.zeros(66, np.uint32)
    for l in moves:
        m[int(l) >> 5] |= np.uint32(1) << np.uint32(int(l) & 31)
    return m


def test_maskgen_matches_reference_golden_lists(host, rules_golden):
    g = rules_golden
    m, c = host(g["boards"], g["side"])
    assert np.array_equal(c, g["counts"].astype(np.int32))
    exp = np.stack([_mask_of(g["moves"][i, :g["counts"][i]]) for i in range(len(c))])
    bad = np.nonzero((m != exp).any(axis=1))[0]
    assert len(bad) == 0, (len(bad), bad[:5])
    assert int(np.unpackbits(m.view(np.uint8), axis=1).sum()) == int(g["counts"].sum())


def test_maskgen_matches_oracle_on_random_playouts(host):
    """Seeded uniform-random playouts with the C oracle (both colours, captures, kings facing, pieces on every edge): the mask
    and the count at every ply, ~12 000 positions."""
    from oracle import oracle as O
    rng = np.random.default_rng(99)
    boards, sides, want = [], [], []
    for game in range(160):
        b, s = O.fen_to_board(O.START_FEN), 0
        for ply in range(110):
            mv = O.legal_moves(b, s)
            boards.append(b.copy()); sides.append(s); want.append(mv)
            if len(mv) == 0 or not (b == 1).any() or not (b == 8).any():
                break
            b = O.apply_move(b, int(mv[rng.integers(len(mv))]))[0]
            s ^= 1
    m, c = host(np.stack(boards), np.array(sides, np.uint8))
    assert len(boards) > 8000
    for i in range(len(boards)):
        assert c[i] == len(want[i]), i
        assert np.array_equal(m[i], _mask_of(want[i])), i


def test_listgen_matches_reference_golden_lists(host, rules_golden):
    """czm_list (the ordered list, one lane = one position) against the reference's 4 381 ordered lists, 0xFFFF padding included."""
    g = rules_golden
    mv, c = host.lists(g["boards"], g["side"])
    assert np.array_equal(c, g["counts"].astype(np.int32))
    bad = np.nonzero((mv != g["moves"]).any(axis=1))[0]
    assert len(bad) == 0, (len(bad), bad[:5], mv[bad[0]][:int(c[bad[0]])], g["moves"][bad[0]][:int(c[bad[0]])])
    m2, c2 = host(g["boards"], g["side"])     # the set czm_list emits beside the list == czm_position's
    assert np.array_equal(host.lists.last_mask, m2) and np.array_equal(c, c2)


def test_listgen_matches_oracle_on_random_playouts(host):
    from oracle import oracle as O
    rng = np.random.default_rng(77)
    boards, sides, want = [], [], []
    for game in range(120):
        b, s = O.fen_to_board(O.START_FEN), 0
        for ply in range(110):
            mv = O.legal_moves(b, s)
            boards.append(b.copy()); sides.append(s); want.append(mv)
            if len(mv) == 0 or not (b == 1).any() or not (b == 8).any():
                break
            b = O.apply_move(b, int(mv[rng.integers(len(mv))]))[0]
            s ^= 1
    mv, c = host.lists(np.stack(boards), np.array(sides, np.uint8))
    assert len(boards) > 6000
    for i in range(len(boards)):
        assert c[i] == len(want[i]), i
        assert np.array_equal(mv[i, :c[i]], np.asarray(want[i], np.uint16)), (i, mv[i, :c[i]], want[i])
        assert (mv[i, c[i]:] == 0xFFFF).all()
        assert np.array_equal(host.lists.last_mask[i], _mask_of(want[i])), i


def test_list_and_mask_on_open_boards_with_up_to_100_moves(host):
    """Sparse synthetic boards (tests/conftest.py::open_boards): lists of 33-100+ moves, twice what playouts from the start position
    reach — the long end of the 128-slot rows, where the list's slots are addressed from the ignore slot at 128."""
    from conftest import open_boards
    from oracle import oracle as O
    boards, side = open_boards(600, 11)
    want = [O.legal_moves(boards[i], int(side[i])) for i in range(len(boards))]
    assert max(len(w) for w in want) >= 95 and np.mean([len(w) for w in want]) > 55
    mv, c = host.lists(boards, side)
    m, c2 = host(boards, side)
    for i in range(len(boards)):
        assert c[i] == len(want[i]) == c2[i], i
        assert np.array_equal(mv[i, :c[i]], np.asarray(want[i], np.uint16)), (i, mv[i, :c[i]], want[i])
        assert (mv[i, c[i]:] == 0xFFFF).all()
        assert np.array_equal(m[i], _mask_of(want[i])) and np.array_equal(host.lists.last_mask[i], m[i]), i


def test_boards_with_more_of_a_kind_than_a_set_holds_are_errors(host):
    """the one-lane-per-position generators take a kind's squares as the lowest and highest of its set (pawns: five
    iterations) — a third rook / cannon / knight / advisor / bishop, a sixth pawn or a second king of the side to move would
    silently lose its moves.  Such a board answers -1 (count 0xFFFF through the C-ABI) from both forms; the same extra piece of
    the side NOT to move is none of the generator's business."""
    from oracle import oracle as O
    start = O.fen_to_board(O.START_FEN)
    own = {"R": 3, "C": 7, "N": 5, "A": 2, "B": 4, "P": 6, "K": 1}          # red's codes ("KARBNPCkarbnpc", code = index + 1)
    boards, sides, bad = [], [], []
    for kind, code in own.items():
        for colour in (0, 1):
            b = start.copy()
            b[4 * 9 + 4] = code + (7 if colour else 0)                      # an extra piece of that kind in the middle of the board
            for s in (0, 1):
                boards.append(b.copy()); sides.append(s); bad.append(s == colour)
    m, c = host(np.stack(boards), np.array(sides, np.uint8))
    mv, c2 = host.lists(np.stack(boards), np.array(sides, np.uint8))
    for i, is_bad in enumerate(bad):
        assert (c[i] == -1) == is_bad and (c2[i] == -1) == is_bad, (i, c[i], c2[i], is_bad)
        if not is_bad:
            want = O.legal_moves(boards[i], sides[i])
            assert c[i] == len(want) and np.array_equal(mv[i, :c2[i]], np.asarray(want, np.uint16))
