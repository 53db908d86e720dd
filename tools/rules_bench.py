#!/usr/bin/env python3
"""Throughput of the stand-alone rules kernels (K1 move generation, K2 make-move, K3 planes) on a batch large
enough to fill the chip: positions/s and the HBM traffic that implies (SURVEY §8d quotes K1 at 312 B/position for
a nibble board + mask; this ABI moves 90 B board + 1 B side in, 256 B ordered list + 264 B mask + 2 B count out).
usage: python tools/rules_bench.py [N positions, default 1048576]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (synthetic position generator)
from cchess_zero_amd.engine import Context  # noqa: E402
from cchess_zero_amd.rules import Rules  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ctx = Context(1, 2, 0)
rules = Rules(ctx)
b0, s0, _ = bench.synth_positions(rules, 8192, seed=5)
rep = (N + 8191) // 8192
boards = b0.repeat(rep, 1)[:N].contiguous()
side = s0.repeat(rep)[:N].contiguous()


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


t = timed(lambda: rules.movegen(boards, side, want_mask=True))
print("K1 movegen (list + mask): %.3f ms for %d positions = %.2f G positions/s, %.1f GB/s of ABI traffic (613 B/position)" % (t * 1e3, N, N / t / 1e9, N * 613 / t / 1e9))
t = timed(lambda: rules.movegen(boards, side, want_mask=True, pad=False))
print("K1 movegen (list + mask, CZ_MOVES_NO_PAD): %.3f ms = %.2f G positions/s, %.1f GB/s algorithmic (312 B/position)" % (t * 1e3, N / t / 1e9, N * 312 / t / 1e9))
t = timed(lambda: rules.movegen(boards, side, want_mask=False, pad=False))
print("K1 movegen (list only, CZ_MOVES_NO_PAD)  : %.3f ms = %.2f G positions/s" % (t * 1e3, N / t / 1e9))
t = timed(lambda: rules.movegen(boards, side, want_mask=True, want_moves=False))
print("K1 movegen (mask only)  : %.3f ms = %.2f G positions/s, %.1f GB/s of ABI traffic (357 B/position), %.1f GB/s algorithmic (312 B)" % (t * 1e3, N / t / 1e9, N * 357 / t / 1e9, N * 312 / t / 1e9))
t = timed(lambda: rules.movegen(boards, side, want_mask=False))
print("K1 movegen (list only)  : %.3f ms = %.2f G positions/s, %.1f GB/s (349 B/position)" % (t * 1e3, N / t / 1e9, N * 349 / t / 1e9))
t = timed(lambda: rules.encode_planes(boards, side, torch.bfloat16, 16))
print("K3 planes (bf16 x16)    : %.3f ms = %.2f G positions/s, %.1f GB/s (2971 B/position)" % (t * 1e3, N / t / 1e9, N * 2971 / t / 1e9))
t = timed(lambda: rules.hash(boards, side))
print("Zobrist hash            : %.3f ms = %.2f G positions/s" % (t * 1e3, N / t / 1e9))
mv, cnt, _ = rules.movegen(boards, side, want_mask=False)
labels = mv[:, 0].contiguous()          # the first legal move of every position (0xFFFF where there is none: left alone)
b2, s2 = boards.clone(), side.clone()
h2 = rules.hash(b2, s2)
t = timed(lambda: rules.apply_move(b2, s2, labels, h2))
print("K2 apply_move (+ hash, capture, terminal flags): %.3f ms = %.2f G positions/s" % (t * 1e3, N / t / 1e9))
