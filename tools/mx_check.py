#!/usr/bin/env python3
"""GPU check of the mx engine (k_trunk_mx_c128): outputs and trunk against its CPU emulation (tests/mxemu.py) and against the
fp32 graph, per depth and weight set; then launch times of the three fp16 engines on 8192 positions.
    python tools/mx_check.py [--blocks 1,2,7] [--time]
"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np, torch
import nethelpers as H
import mxemu
from cchess_zero_amd.net import PolicyValueNet

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", default="1,2,3,7")
ap.add_argument("--wsets", default="glorot,structured,trained_like")
ap.add_argument("--time", action="store_true")
ap.add_argument("--n", type=int, default=37)
ap.add_argument("--engines", default="fp16,x3,mx", help="--time: which engines")
ap.add_argument("--launches", type=int, default=20)
args = ap.parse_args()
torch.set_grad_enabled(False)
for blocks in [int(b) for b in args.blocks.split(",") if b]:
    for wset in args.wsets.split(","):
        if wset == "structured" and blocks > 3:
            continue
        net = PolicyValueNet(blocks, "cuda:0", torch.float16, seed=1, split="mx")
        H.WEIGHT_SETS[wset](net)
        x = H.positions(args.n, 2)
        xd = torch.from_numpy(x).cuda()
        lg, vg = net.forward_device(xd)
        tg = net.tower(xd).float().cpu()                                  # [B,128,9,10]
        mod = net.module.float().cpu()
        xc = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()
        le, ve, te = mxemu.forward_mx(mod, xc)
        m64 = mod.double()
        l64, v64 = m64(xc.double())
        mod.float()
        net.module.to("cuda:0")
        lgc, vgc = lg.cpu().double(), vg.cpu().double().reshape(-1)
        print("%2d blocks %-12s max|logit| %.3g  max trunk %.3g | kernel vs emulation: trunk %.3g (rel %.2g) logit %.3g value %.3g | kernel vs fp64 graph: dlogit %.3g dvalue %.3g | emulation vs fp64: %.3g %.3g"
              % (blocks, wset, float(l64.abs().max()), float(te.abs().max()), float((tg - te).abs().max()), float((tg - te).abs().max() / te.abs().max()),
                 float((lgc - le.double()).abs().max()), float((vgc - ve.double().reshape(-1)).abs().max()),
                 float((lgc - l64).abs().max()), float((vgc - v64.reshape(-1)).abs().max()),
                 float((le.double() - l64).abs().max()), float((ve.double() - v64).abs().max())), flush=True)
        if not torch.isfinite(lg).all():
            print("   NON-FINITE outputs")
if args.time:
    x = torch.from_numpy(H.positions(64, 3)).cuda().repeat(128, 1, 1, 1)
    x16 = torch.zeros((8192, 9, 10, 16), dtype=torch.float16, device="cuda")
    x16[..., :14] = x.to(torch.float16)
    for name, split in (("fp16", False), ("x3", True), ("mx", "mx")):
        if name not in args.engines.split(","):
            continue
        net = PolicyValueNet(7, "cuda:0", torch.float16, seed=0, split=split)
        for rep in range(2):
            net._hip_net_forward(x16)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.launches):
                net._hip_net_forward(x16)
            e1.record()
            torch.cuda.synchronize()
            print("%-5s 7 blocks, 8192 positions: %.1f us per launch" % (name, e0.elapsed_time(e1) / args.launches * 1e3), flush=True)
