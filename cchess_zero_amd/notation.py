"""State-string <-> board-array conversion (host side, tiny).

The reference's state is a FEN-like string: 10 ranks joined by '/', rank 0 first (red / upper-case /
'w' home), digits = runs of empty squares (main.py:585, :705-714, :691-699).  The device format is
uint8[90], sq = y*9 + x, code = 1 + index in 'KARBNPCkarbnpc' (main.py:208), 0 = empty.
"""
import numpy as np

PIECES = ".KARBNPCkarbnpc"
START_STATE = "RNBAKABNR/9/1C5C1/P1P1P1P1P/9/9/p1p1p1p1p/1c5c1/9/rnbakabnr"
_CODE = {c: i for i, c in enumerate(PIECES) if i}
# the reference also accepts 'h'/'H' for knights and 'e'/'E' for bishops in get_legal_moves (main.py:835,857)
_CODE.update({"H": _CODE["N"], "h": _CODE["n"], "E": _CODE["B"], "e": _CODE["b"]})


def state_to_board(state):
    b = np.zeros(90, np.uint8)
    rows = state.split("/")
    if len(rows) != 10:
        raise ValueError("state must have 10 ranks: %r" % state)
    for y, row in enumerate(rows):
        x = 0
        for ch in row:
            if ch.isdigit():
                x += int(ch)
            else:
                if x >= 9 or ch not in _CODE:
                    raise ValueError("bad state string: %r" % state)
                b[y * 9 + x] = _CODE[ch]
                x += 1
        if x != 9:
            raise ValueError("rank %d of %r does not have 9 files" % (y, state))
    return b


def board_to_state(board):
    board = np.asarray(board, np.uint8).reshape(10, 9)
    rows = []
    for y in range(10):
        s, run = "", 0
        for x in range(9):
            c = int(board[y, x])
            if c == 0:
                run += 1
            else:
                if run:
                    s += str(run)
                    run = 0
                s += PIECES[c]
        if run:
            s += str(run)
        rows.append(s)
    return "/".join(rows)


def player_to_side(player):
    return 1 if player == "b" else 0


def side_to_player(side):
    return "b" if side else "w"
