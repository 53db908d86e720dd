#!/usr/bin/env python3
"""How many leaf evaluations of a search are repeats of a position already evaluated (transpositions inside a tree,
or anywhere in the batch)?  Upper bound of what an evaluation cache keyed by position could save.
usage: python tools/dup_rate.py [games=256] [steps=1600]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cchess_zero_amd.engine import Context, SearchEngine  # noqa: E402
from cchess_zero_amd.net import PolicyValueNet  # noqa: E402
from cchess_zero_amd.rules import Rules  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1600
ctx = Context(G, (steps + 2) * 80, 0)
rules = Rules(ctx)
eng = SearchEngine(G, (steps + 2) * 80, 0, plane_dtype=torch.bfloat16, channels=16, ctx=ctx)
net = PolicyValueNet(7, "cuda:0", torch.bfloat16, seed=0, ctx=ctx)
boards, side, rr = bench.synth_positions(rules, G, seed=1000)
eng.reset(boards, side, rr)
gen = torch.Generator(device="cuda").manual_seed(1)
mix = torch.randint(-2**62, 2**62, (360,), dtype=torch.int64, device="cuda", generator=gen)
keys = []
eng.step(net.forward_device, mode=0)
for s in range(steps):
    planes, need = eng.select(1)
    k = (planes.reshape(G, 360, 4).view(torch.int64).reshape(G, 360) * mix).sum(dim=1)   # wraps: a 64-bit position key
    keys.append(torch.where(need.bool(), k, torch.full_like(k, 0)).cpu().numpy())
    z, value = net.search_eval(planes)
    eng.expand_backup_fc(z, value, net.pfc_w_rows, net.pfc_b_f32)
keys = np.stack(keys)          # [steps, G]
tot = int((keys != 0).sum())
per_tree = sum(len(np.unique(keys[:, g][keys[:, g] != 0])) for g in range(G))
glob = len(np.unique(keys[keys != 0]))
print("leaf evaluations %d; distinct within their own tree %d (repeat rate %.1f %%); distinct over the whole batch %d (repeat rate %.1f %%)"
      % (tot, per_tree, 100.0 * (1 - per_tree / tot), glob, 100.0 * (1 - glob / tot)))
for upto in (100, 400, 800, 1600):
    if upto <= steps:
        kk = keys[:upto]
        t = int((kk != 0).sum())
        pt = sum(len(np.unique(kk[:, g][kk[:, g] != 0])) for g in range(G))
        print("  first %4d simulations: in-tree repeat rate %.1f %%" % (upto, 100.0 * (1 - pt / t)))
