#!/usr/bin/env python3
"""Experiment: does running the batch as TWO half-batches on two HIP streams hide the launch gaps and the small tree kernels
of one half behind the other half's trunk?  (A trunk workgroup owns its CU, so nothing co-resides with it — but between the
four dependent launches of a step the GPU is idle for ~10 us each, and the tree kernels run alone for ~90 us per step.)
usage: python tools/two_stream_exp.py [games=8192] [steps=600]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cchess_zero_amd.engine import Context, SearchEngine  # noqa: E402
from cchess_zero_amd.net import PolicyValueNet  # noqa: E402
from cchess_zero_amd.rules import Rules  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
cap = bench.default_nodes_per_tree(1600)


def make(g, seed):
    ctx = Context(g, cap, 0)
    rules = Rules(ctx)
    eng = SearchEngine(g, cap, 0, plane_dtype=torch.bfloat16, channels=16, ctx=ctx)
    net = PolicyValueNet(7, "cuda:0", torch.bfloat16, seed=0, ctx=ctx)
    b, s, r = bench.synth_positions(rules, g, seed=seed)
    eng.reset(b, s, r)
    eng.set_terminal_extra(4)
    eng.set_sim_target(1600)
    eng.step(net.forward_device, mode=0)
    return eng, net


def run(parts, streams, n):
    for _ in range(n):
        for (eng, net), st in zip(parts, streams):
            with torch.cuda.stream(st):
                eng.step(net.forward_device, mode=1)


def measure(parts, streams, label):
    run(parts, streams, 16)
    torch.cuda.synchronize()
    s0 = sum(int(e.status()[2].sum().item()) for e, _ in parts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(parts, streams, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1 = sum(int(e.status()[2].sum().item()) for e, _ in parts)
    print("%-34s %8.3f ms per full step  %10.0f sims/s" % (label, dt / steps * 1e3, (s1 - s0) / dt), flush=True)


for rep in range(2):
    one = [make(G, 1000)]
    measure(one, [torch.cuda.current_stream()], "one stream, %d trees" % G)
    del one
    torch.cuda.empty_cache()
    two = [make(G // 2, 1000), make(G // 2, 2000)]
    measure(two, [torch.cuda.Stream(), torch.cuda.Stream()], "two streams, 2 x %d trees" % (G // 2))
    del two
    torch.cuda.empty_cache()
