"""Experiment: does running the lock-step loop as TWO half-batches on two HIP streams (trunk tails and the tree kernels of one
half in the shadow of the other half's trunk) beat one full batch?  One process, one GPU; fresh searches (no ageing), same
positions; prints simulations/s for 1 x 8192 and 2 x 4096 (and 4 x 2048).  Measurement only — not a product path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from cchess_zero_amd.engine import Context, SearchEngine
from cchess_zero_amd.net import PolicyValueNet
from cchess_zero_amd.rules import Rules

G, STEPS, WARM = 8192, int(os.environ.get("STEPS", 300)), 60
dev = torch.device("cuda", 0)
cap = bench.default_nodes_per_tree(1600)
tdt = torch.float16


def run(parts):
    g = G // parts
    streams = [torch.cuda.Stream() for _ in range(parts)]
    engs, nets = [], []
    for p in range(parts):
        with torch.cuda.stream(streams[p]):
            ctx = Context(g, cap, 0)
            rules = Rules(ctx)
            eng = SearchEngine(g, cap, 0, plane_dtype=tdt, channels=16, ctx=ctx)
            net = PolicyValueNet(7, dev, tdt, seed=0, backend="auto", ctx=ctx)
            eng.compact = False
            b, s, rr = bench.synth_positions(rules, g, seed=1000 + p)
            eng.set_terminal_extra(4)
            eng.reset(b, s, rr)
            eng.step(net.forward_device, mode=0)
            engs.append(eng); nets.append(net)
    torch.cuda.synchronize()

    def steps(n):
        for _ in range(n):
            for p in range(parts):
                with torch.cuda.stream(streams[p]):
                    engs[p].step(nets[p].forward_device, mode=1)
    steps(WARM)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps(STEPS)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d x %d games: %.3f ms per full step, %.3f M tree-steps/s" % (parts, g, dt / STEPS * 1e3, G * STEPS / dt / 1e6), flush=True)
    for e in engs:
        e.ctx.close()
    del engs, nets
    torch.cuda.empty_cache()


for rep in range(2):
    for parts in (1, 2, 4):
        run(parts)
