"""Batched rules ops K1-K3 over device tensors (host side of cz_movegen / cz_apply_move / cz_encode_planes)."""
import ctypes as C

import numpy as np
import torch

from ._lib import BF16, F16, F32, MASK_WORDS, MAXMOVES, NLABELS, NSQ, check, lib
from .engine import Context, _ptr


# the start position (main.py:585: RNBAKABNR/9/1C5C1/P1P1P1P1P/9/9/p1p1p1p1p/1c5c1/9/rnbakabnr) as piece codes, sq = 9 y + x
START_BOARD = np.array([3, 5, 4, 2, 1, 2, 4, 5, 3] + [0] * 9 + [0, 7, 0, 0, 0, 0, 0, 7, 0] + [6, 0, 6, 0, 6, 0, 6, 0, 6] + [0] * 18 +
                       [13, 0, 13, 0, 13, 0, 13, 0, 13] + [0, 14, 0, 0, 0, 0, 0, 14, 0] + [0] * 9 + [10, 12, 11, 9, 8, 9, 11, 12, 10], np.uint8)


def random_positions(rules, G, seed, max_ply=80):
    """Seeded uniform-random playouts from the start position, ply ~ U[0, max_ply] per game, all on the GPU (K1 movegen ->
    random pick -> K2 apply); a move that would capture a king is not played (both kings stay on the board).  The synthetic
    positions of SURVEY 8(d) -> (boards [G,90] u8, side [G] u8, restrict_round [G] i32), device tensors."""
    dev = rules.dev
    gen = torch.Generator(device=dev).manual_seed(seed)
    boards = torch.from_numpy(np.tile(START_BOARD, (G, 1))).to(dev)
    side = torch.zeros(G, dtype=torch.uint8, device=dev)
    rr = torch.zeros(G, dtype=torch.int32, device=dev)
    target = torch.randint(0, max_ply + 1, (G,), generator=gen, device=dev)
    alive = torch.ones(G, dtype=torch.bool, device=dev)
    for ply in range(max_ply):
        moves, count, _ = rules.movegen(boards, side, want_mask=False)
        cnt = count.to(torch.int64) & 0xFFFF
        go = alive & (target > ply) & (cnt > 0)
        r = (torch.rand(G, generator=gen, device=dev) * cnt.clamp(min=1)).to(torch.int64).clamp(max=127)
        pick = moves.gather(1, r.unsqueeze(1)).squeeze(1)
        nb, ns = boards.clone(), side.clone()
        lab = torch.where(go, pick, torch.full_like(pick, -1))
        cap, term = rules.apply_move(nb, ns, lab)
        ok = go & (term == 0)
        boards = torch.where(ok.unsqueeze(1), nb, boards)
        side = torch.where(ok, ns, side)
        rr = torch.where(ok, torch.where(cap != 0, torch.zeros_like(rr), rr + 1), rr)
        alive = alive & (ok | ~go)
    return boards.contiguous(), side.contiguous(), rr.contiguous()


class Rules:
    def __init__(self, ctx=None, device=0):
        self.ctx = ctx or Context(1, 2, device)
        self.dev = self.ctx.device

    def _dev(self, a, dtype):
        if torch.is_tensor(a):
            return a.to(self.dev).to(dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a)).to(self.dev).to(dtype).contiguous()

    def movegen(self, boards, side, want_mask=True, want_moves=True, pad=True, strict=False):
        """GameBoard.get_legal_moves for G positions -> (moves [G,128] i16(u16 bits), count [G], mask [G,66] i32).
        want_moves=False: only the legal-move SET (mask) and the count — the mask-only kernel (k_movegen_mask), moves is None.
        pad=False (CZ_MOVES_NO_PAD): rows are valid up to their count only, the rest of the (uninitialised) buffer is not written.
        strict=True: raise if a board answered count 0xFFFF (not a Xiangqi set: its list / mask row is undefined; include/
        cchess_hip.h) — synchronises; without it a caller checks `count == -1` (0xFFFF as int16) itself before it reads rows."""
        self.ctx.bind_stream()   # torch's current stream
        boards = self._dev(boards, torch.uint8).reshape(-1, NSQ)
        side = self._dev(side, torch.uint8)
        G = boards.shape[0]
        moves = torch.empty((G, MAXMOVES), dtype=torch.int16, device=self.dev) if want_moves else None
        count = torch.empty(G, dtype=torch.int16, device=self.dev)
        mask = torch.empty((G, MASK_WORDS), dtype=torch.int32, device=self.dev) if want_mask else None
        if pad:
            check(lib().cz_movegen(self.ctx.h, _ptr(boards), _ptr(side), G, _ptr(moves), _ptr(count), _ptr(mask)), "cz_movegen")
        else:
            check(lib().cz_movegen_ex(self.ctx.h, _ptr(boards), _ptr(side), G, _ptr(moves), _ptr(count), _ptr(mask), 1), "cz_movegen_ex")
        if strict:
            self.check_counts(count)
        return moves, count, mask

    @staticmethod
    def check_counts(count):
        """Raises if any position answered count 0xFFFF (k_movegen_list / k_movegen_mask refuse boards that are not a Xiangqi
        set; their rows are undefined and must not be consumed)."""
        bad = (count == -1).nonzero().flatten()
        if bad.numel():
            from ._lib import CchessHipError
            raise CchessHipError("cz_movegen: %d position(s) are not a Xiangqi set (count 0xFFFF), first at index %d: their move rows are undefined"
                                 % (int(bad.numel()), int(bad[0])))

    def apply_move(self, boards, side, labels, hash_=None):
        """In-place GameBoard.sim_do_action for G games -> (captured [G] u8, terminal [G] i8)."""
        self.ctx.bind_stream()   # torch's current stream
        G = boards.shape[0]
        assert boards.is_cuda and side.is_cuda and boards.dtype == torch.uint8 and boards.is_contiguous()
        labels = self._dev(labels, torch.int16) if not (torch.is_tensor(labels) and labels.dtype == torch.int16) else labels.to(self.dev).contiguous()
        cap = torch.empty(G, dtype=torch.uint8, device=self.dev)
        term = torch.empty(G, dtype=torch.int8, device=self.dev)
        check(lib().cz_apply_move(self.ctx.h, _ptr(boards), _ptr(side), _ptr(labels), G, _ptr(hash_), _ptr(cap), _ptr(term)), "cz_apply_move")
        return cap, term

    def hash(self, boards, side):
        self.ctx.bind_stream()   # torch's current stream
        boards = self._dev(boards, torch.uint8).reshape(-1, NSQ)
        side = self._dev(side, torch.uint8)
        h = torch.empty(boards.shape[0], dtype=torch.int64, device=self.dev)
        check(lib().cz_hash(self.ctx.h, _ptr(boards), _ptr(side), boards.shape[0], _ptr(h)), "cz_hash")
        return h

    def encode_planes(self, boards, side, dtype=torch.float32, channels=14, quirk_q1=True):
        self.ctx.bind_stream()   # torch's current stream
        boards = self._dev(boards, torch.uint8).reshape(-1, NSQ)
        side = self._dev(side, torch.uint8)
        G = boards.shape[0]
        out = torch.empty((G, 9, 10, channels), dtype=dtype, device=self.dev)
        check(lib().cz_encode_planes(self.ctx.h, _ptr(boards), _ptr(side), G, _ptr(out), {torch.bfloat16: BF16, torch.float16: F16}.get(dtype, F32),
                                     channels, 1 if quirk_q1 else 0), "cz_encode_planes")
        return out
