#!/usr/bin/env python3
"""Measured HBM ceilings of the box (torch kernels, HIP events): pure stores (fill_), pure loads (a sum), copy — the context for
the rules kernels' roofline fractions, whose traffic is mostly stores (mask rows, list rows, planes).
usage: python tools/hbm_ceiling.py [GiB per buffer, default 2]"""
import sys

import torch

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n = int(gib * (1 << 30)) // 4
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


b = n * 4
t = timed(lambda: x.fill_(1.0))
print("stores only (fill_, %.1f GiB): %.3f ms = %.2f TB/s" % (gib, t * 1e3, b / t / 1e12))
t = timed(lambda: x.zero_())
print("stores only (zero_ = memset): %.3f ms = %.2f TB/s" % (t * 1e3, b / t / 1e12))
t = timed(lambda: x.sum())
print("loads only (sum): %.3f ms = %.2f TB/s" % (t * 1e3, b / t / 1e12))
t = timed(lambda: y.copy_(x))
print("copy (read + write): %.3f ms = %.2f TB/s of traffic (%.2f TB/s each way)" % (t * 1e3, 2 * b / t / 1e12, b / t / 1e12))
t = timed(lambda: torch.add(x, 1.0, out=y))
print("y = x + 1: %.3f ms = %.2f TB/s of traffic" % (t * 1e3, 2 * b / t / 1e12))
