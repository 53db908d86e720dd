# per-step GPU time right after the synchronisation that opens bench.py's timed region
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from cchess_zero_amd.engine import Context, SearchEngine
from cchess_zero_amd.net import PolicyValueNet
from cchess_zero_amd.rules import Rules
G, playout = 8192, 1600
cap = bench.default_nodes_per_tree(playout)
ctx = Context(G, cap, 0); rules = Rules(ctx)
eng = SearchEngine(G, cap, 0, plane_dtype=torch.float16, channels=16, ctx=ctx)
net = PolicyValueNet(7, "cuda:0", torch.float16, seed=0, ctx=ctx)
boards, side, rr = bench.synth_positions(rules, G, 1000)
eng.reset(boards, side, rr); eng.set_terminal_extra(4); eng.set_sim_target(playout)
eng.step(net.forward_device, mode=0)
for _ in range(300): eng.step(net.forward_device, mode=1)
for gap_ms in (0.0, 0.3, 5.0, 50.0):
    torch.cuda.synchronize()
    if gap_ms: time.sleep(gap_ms / 1e3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(40):
        eng.step(net.forward_device, mode=1)
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(40)]
    print("idle gap %5.1f ms: wall/step %.3f ms; per-step GPU ms: first %s ... mean(last 20) %.3f" % (gap_ms, wall / 40 * 1e3, " ".join("%.3f" % x for x in ms[:8]), float(np.mean(ms[20:]))))
