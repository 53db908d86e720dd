import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def rules_golden():
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "rules.npz")))


@pytest.fixture(scope="session")
def mcts_golden():
    import json
    return json.load(open(os.path.join(GOLDEN, "mcts.json")))


@pytest.fixture(scope="session")
def mcts_deep_golden():
    """Reference searches at the metric's depth (playout 1600) and with selected paths of 40-60 levels (gen_mcts_deep)."""
    import json
    return json.load(open(os.path.join(GOLDEN, "mcts_deep.json")))


@pytest.fixture(scope="session")
def tables_golden():
    import json
    return json.load(open(os.path.join(GOLDEN, "tables.json")))


def open_boards(n, seed):
    """n synthetic Xiangqi positions with FEW pieces on an open board (every kind on squares its rules allow, both kings present,
    the side to move rich in rooks / cannons / knights): lists of 60-110 moves, which random playouts from the start position
    never reach — the long end of the ordered list's 128 slots.  Returns (boards uint8 [n, 90], side uint8 [n])."""
    import numpy as np
    rng = np.random.default_rng(seed)
    palace = [(y, x) for y in (0, 1, 2) for x in (3, 4, 5)]
    adv = [(0, 3), (0, 5), (1, 4), (2, 3), (2, 5)]
    bis = [(0, 2), (0, 6), (2, 0), (2, 4), (2, 8), (4, 2), (4, 6)]
    kinds = {"K": (1, 1, palace), "A": (2, 2, adv), "B": (4, 2, bis), "R": (3, 2, None), "N": (5, 2, None), "C": (7, 2, None), "P": (6, 5, None)}
    boards = np.zeros((n, 90), np.uint8)
    side = rng.integers(0, 2, n).astype(np.uint8)
    for i in range(n):
        b = boards[i]
        for colour in (int(side[i]), 1 - int(side[i])):       # the mover's pieces first: they get the squares they want
            mover = colour == side[i]
            for k, (code, cnt, where) in kinds.items():
                if i % 3 == 2:     # the longest lists: every rook, cannon, knight and pawn (pawns across the river), no advisors / bishops, a bare enemy king
                    keep = (1.0 if k in "KRCNP" else 0.0) if mover else 0.0
                else:
                    keep = (1.0 if k in "KRCN" else 0.5 if i % 2 else 0.9) if mover else (0.1 if i % 2 else 0.3)
                for _ in range(cnt):
                    if k != "K" and rng.random() > keep:
                        continue
                    for _try in range(20):
                        if where is None:
                            y, x = int(rng.integers(0, 10)), int(rng.integers(0, 9))
                            if k == "P" and (y < (5 if i % 3 == 2 else 3) if colour == 0 else y > (4 if i % 3 == 2 else 6)):
                                continue                        # a pawn is never behind its own starting rank
                            yy = y
                        else:
                            y, x = where[int(rng.integers(len(where)))]
                            yy = y if colour == 0 else 9 - y
                        if b[yy * 9 + x] == 0:
                            b[yy * 9 + x] = code + (7 if colour else 0)
                            break
    # every third board climbs: 300 single-piece relocations of the mover's pieces, kept when the C oracle's list does not get
    # shorter — ~100 moves per position, the long end of what a Xiangqi set can have
    from oracle import oracle as O
    for i in range(2, n, 3):
        b, sd = boards[i], int(side[i])
        c = len(O.legal_moves(b, sd))
        for _ in range(300):
            src = [q for q in range(90) if b[q] and ((b[q] <= 7) == (sd == 0)) and b[q] not in (1, 8)]
            q = src[int(rng.integers(len(src)))]
            code, y, x = int(b[q]), int(rng.integers(0, 10)), int(rng.integers(0, 9))
            if b[y * 9 + x] or (code == 6 and y < 5) or (code == 13 and y > 4):
                continue
            nb = b.copy()
            nb[q], nb[y * 9 + x] = 0, code
            c2 = len(O.legal_moves(nb, sd))
            if c2 >= c:
                b[:], c = nb, c2
    return boards, side
