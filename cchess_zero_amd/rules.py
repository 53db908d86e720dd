"""Batched rules ops K1-K3 over device tensors (host side of cz_movegen / cz_apply_move / cz_encode_planes)."""
import ctypes as C

import numpy as np
import torch

from ._lib import BF16, F16, F32, MASK_WORDS, MAXMOVES, NLABELS, NSQ, check, lib
from .engine import Context, _ptr


class Rules:
    def __init__(self, ctx=None, device=0):
        self.ctx = ctx or Context(1, 2, device)
        self.dev = self.ctx.device

    def _dev(self, a, dtype):
        if torch.is_tensor(a):
            return a.to(self.dev).to(dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a)).to(self.dev).to(dtype).contiguous()

    def movegen(self, boards, side, want_mask=True, want_moves=True):
        """GameBoard.get_legal_moves for G positions -> (moves [G,128] i16(u16 bits), count [G], mask [G,66] i32).
        want_moves=False: only the legal-move SET (mask) and the count — the mask-only kernel (k_movegen_mask), moves is None."""
        self.ctx.bind_stream()   # torch's current stream
        boards = self._dev(boards, torch.uint8).reshape(-1, NSQ)
        side = self._dev(side, torch.uint8)
        G = boards.shape[0]
        moves = torch.empty((G, MAXMOVES), dtype=torch.int16, device=self.dev) if want_moves else None
        count = torch.empty(G, dtype=torch.int16, device=self.dev)
        mask = torch.empty((G, MASK_WORDS), dtype=torch.int32, device=self.dev) if want_mask else None
        check(lib().cz_movegen(self.ctx.h, _ptr(boards), _ptr(side), G, _ptr(moves), _ptr(count), _ptr(mask)), "cz_movegen")
        return moves, count, mask

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
