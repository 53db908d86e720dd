"""ctypes binding of oracle/libcchess_oracle.so (TEST INFRASTRUCTURE).

The C file restates the reference (chengstone/cchess-zero main.py) function by
function; see cchess_oracle.h for the file:line map.  Build: `make -C oracle`.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcchess_oracle.so")

NLABELS = 2086
MAXMOVES = 128
NSQ = 90
START_FEN = "RNBAKABNR/9/1C5C1/P1P1P1P1P/9/9/p1p1p1p1p/1c5c1/9/rnbakabnr"  # main.py:585


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "cchess_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.czo_labels.restype = C.c_void_p
        L.czo_lut.restype = C.c_void_p
        L.czo_unflip.restype = C.c_void_p
        L.czo_label_srcdst.restype = C.c_void_p
        L.czo_hash.restype = C.c_uint64
        L.czo_zobrist_key.restype = C.c_uint64
        L.czo_zobrist_side.restype = C.c_uint64
        L.czo_search_create.restype = C.c_void_p
        L.czo_search_create.argtypes = [C.c_int, C.c_int]
        L.czo_search_destroy.argtypes = [C.c_void_p]
        for name in ("czo_search_reset", "czo_search_select", "czo_search_expand_backup", "czo_search_root_stats",
                     "czo_search_advance", "czo_search_status", "czo_search_root_state", "czo_search_last_depth",
                     "czo_search_tree_dump", "czo_search_select_k", "czo_search_expand_backup_k", "czo_search_set_sim_target"):
            getattr(L, name).restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def labels():
    raw = C.string_at(lib().czo_labels(), NLABELS * 5)
    return [raw[i * 5:i * 5 + 4].decode() for i in range(NLABELS)]


def lut():
    return np.ctypeslib.as_array(C.cast(lib().czo_lut(), C.POINTER(C.c_int16)), shape=(NSQ * NSQ,)).reshape(NSQ, NSQ).copy()


def unflip():
    return np.ctypeslib.as_array(C.cast(lib().czo_unflip(), C.POINTER(C.c_int16)), shape=(NLABELS,)).copy()


def label_srcdst():
    return np.ctypeslib.as_array(C.cast(lib().czo_label_srcdst(), C.POINTER(C.c_uint16)), shape=(NLABELS,)).copy()


def fen_to_board(fen):
    b = np.zeros(NSQ, np.uint8)
    if lib().czo_fen_to_board(fen.encode(), _p(b)) != 0:
        raise ValueError("bad state string: %r" % fen)
    return b


def board_to_fen(board):
    board = np.ascontiguousarray(board, np.uint8)
    out = C.create_string_buffer(128)
    lib().czo_board_to_fen(_p(board), out)
    return out.value.decode()


def legal_moves(board, side):
    """-> np.uint16 label array in reference generation order."""
    board = np.ascontiguousarray(board, np.uint8)
    out = np.zeros(MAXMOVES, np.uint16)
    n = lib().czo_legal_moves(_p(board), int(side), _p(out))
    if n < 0:
        raise RuntimeError("oracle movegen overflow / unlabeled move")
    return out[:n].copy()


def apply_move(board, label):
    """-> (new_board, captured_code, terminal_flags)"""
    nb = np.array(board, np.uint8, copy=True)
    cap = C.c_uint8(0)
    term = lib().czo_apply_move(_p(nb), C.c_uint16(int(label)), C.byref(cap))
    return nb, cap.value, term


def encode_planes(board, side, quirk_q1=True):
    board = np.ascontiguousarray(board, np.uint8)
    out = np.zeros((9, 10, 14), np.float32)
    lib().czo_encode_planes(_p(board), int(side), int(bool(quirk_q1)), _p(out))
    return out


def zhash(board, side):
    board = np.ascontiguousarray(board, np.uint8)
    return int(lib().czo_hash(_p(board), int(side)))


def zobrist_table():
    L = lib()
    t = np.zeros((15, NSQ), np.uint64)
    for c in range(1, 15):
        for q in range(NSQ):
            t[c, q] = L.czo_zobrist_key(c, q)
    return t, int(L.czo_zobrist_side())


class Search:
    """Lock-step search over G trees; same call sequence as the HIP engine."""

    def __init__(self, max_games, max_nodes_per_tree):
        self.h = C.c_void_p(lib().czo_search_create(max_games, max_nodes_per_tree))
        self.max_games = max_games
        self.G = 0

    def close(self):
        if self.h:
            lib().czo_search_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, boards, side, rr=None):
        boards = np.ascontiguousarray(boards, np.uint8).reshape(-1, NSQ)
        side = np.ascontiguousarray(side, np.uint8)
        G = boards.shape[0]
        rr_a = None if rr is None else np.ascontiguousarray(rr, np.int32)
        if lib().czo_search_reset(self.h, _p(boards), _p(side), _p(rr_a), G) != 0:
            raise RuntimeError("czo_search_reset failed")
        self.G = G

    def select(self, mode=1):
        planes = np.zeros((self.G, 9, 10, 14), np.float32)
        need = np.zeros(self.G, np.uint8)
        lib().czo_search_select(self.h, int(mode), _p(planes), _p(need))
        return planes, need

    def expand_backup(self, logits, value):
        logits = np.ascontiguousarray(logits, np.float32).reshape(self.G, NLABELS)
        value = np.ascontiguousarray(value, np.float32).reshape(self.G)
        lib().czo_search_expand_backup(self.h, _p(logits), _p(value))

    def set_sim_target(self, target):
        lib().czo_search_set_sim_target(self.h, int(target))

    def select_k(self, mode, K):
        planes = np.zeros((self.G * K, 9, 10, 14), np.float32)
        need = np.zeros(self.G * K, np.uint8)
        lib().czo_search_select_k(self.h, int(mode), int(K), _p(planes), _p(need))
        return planes, need

    def expand_backup_k(self, K, logits, value):
        logits = np.ascontiguousarray(logits, np.float32).reshape(self.G * K, NLABELS)
        value = np.ascontiguousarray(value, np.float32).reshape(self.G * K)
        lib().czo_search_expand_backup_k(self.h, int(K), _p(logits), _p(value))

    def root_stats(self):
        G = self.G
        label = np.zeros((G, MAXMOVES), np.uint16)
        N = np.zeros((G, MAXMOVES), np.int32)
        Q = np.zeros((G, MAXMOVES), np.float32)
        P = np.zeros((G, MAXMOVES), np.float32)
        W = np.zeros((G, MAXMOVES), np.float32)
        count = np.zeros(G, np.uint16)
        lib().czo_search_root_stats(self.h, _p(label), _p(N), _p(Q), _p(P), _p(W), _p(count))
        return dict(label=label, N=N, Q=Q, P=P, W=W, count=count)

    def advance(self, played):
        played = np.ascontiguousarray(played, np.uint16)
        lib().czo_search_advance(self.h, _p(played))

    def status(self):
        st = np.zeros(self.G, np.int32)
        nodes = np.zeros(self.G, np.int32)
        sims = np.zeros(self.G, np.int32)
        lib().czo_search_status(self.h, _p(st), _p(nodes), _p(sims))
        return st, nodes, sims

    def root_state(self):
        b = np.zeros((self.G, NSQ), np.uint8)
        s = np.zeros(self.G, np.uint8)
        rr = np.zeros(self.G, np.int32)
        lib().czo_search_root_state(self.h, _p(b), _p(s), _p(rr))
        return b, s, rr

    def tree_dump(self, g, max_records=1 << 20):
        out = np.zeros((max_records, 7), np.int32)
        n = lib().czo_search_tree_dump(self.h, int(g), _p(out), int(max_records))
        if n > max_records:
            raise RuntimeError("tree_dump: %d records > %d" % (n, max_records))
        return out[:n].copy()

    def last_depth(self):
        d = np.zeros(self.G, np.int32)
        lib().czo_search_last_depth(self.h, _p(d))
        return d
