#!/usr/bin/env python3
"""GPU box: how representative is the position set precision "strict" measures itself on?  mx6's error against the fp32 module on
(a) the self-check's seeded random-playout positions at several sizes / depths and (b) the corpus positions of the reference's
golden games (tests/golden/rules.npz), for the weight sets of the test suite.  -> the tolerance / probe size in net.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np, torch
import nethelpers as H
from cchess_zero_amd.net import PolicyValueNet, net_error
from cchess_zero_amd.rules import Rules, random_positions
torch.set_grad_enabled(False)


def heavy_(net, scale=80.0, seed=17):
    H.trained_like_(net)
    gen = torch.Generator().manual_seed(seed)
    m = net.module
    for cb in m.convbns()[:-2]:
        cb.moving_var.copy_(torch.exp(torch.randn(cb.moving_var.shape, generator=gen) * 0.7).to(cb.moving_var.device))
    x = torch.from_numpy(H.positions(96, 123)).to(m.policy_fc.weight.device).permute(0, 3, 1, 2)
    lg, _ = m(x)
    m.policy_fc.weight.mul_(scale / float(lg.max(dim=1).values.mean()))
    net.refresh()


r = Rules()
sets = {}
for name, n, ply, seed in (("random playouts 64 x ply<=80", 64, 80, 20260930), ("random playouts 256 x ply<=80", 256, 80, 7), ("random playouts 256 x ply<=160", 256, 160, 8),
                           ("random playouts 1024 x ply<=240", 1024, 240, 9)):
    b, s, _ = random_positions(r, n, seed, ply)
    sets[name] = r.encode_planes(b, s).float()
for seed in (2, 3):
    sets["corpus 64 (seed %d)" % seed] = torch.from_numpy(H.positions(64, seed)).cuda()
sets["corpus 512"] = torch.from_numpy(H.positions(512, 11)).cuda()
for blocks, wname in ((7, "trained_like"), (7, "heavy"), (19, "trained_like"), (7, "glorot")):
    net = PolicyValueNet(blocks, "cuda:0", torch.float16, seed=1, split="mx")
    {"trained_like": H.trained_like_, "heavy": heavy_, "glorot": lambda n: n}[wname](net)
    for name, x in sets.items():
        e = net_error(net, x)
        print("%2d blocks %-12s mx6 on %-32s dlogit %.3g dvalue %.3g (max|logit| %.3g)" % (blocks, wname, name, e["dlogit"], e["dvalue"], e["max_abs_logit"]), flush=True)
