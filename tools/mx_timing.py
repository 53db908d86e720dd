#!/usr/bin/env python3
"""Diagnostic (GPU box): where the cycles of a k_trunk_mx2_c128 workgroup go.  Needs a library built with -DMX2_TIMING
(tools/experiments/mx_ablate.sh timing=define:MX2_TIMING; CCHESS_HIP_LIB=tools/ab/lib_mx_timing.so): the clock-probe buffer then
carries, per workgroup (wave 0's view), the shader cycles before the tower, inside the slab loops, in drain + exchange, and in
epilogue + next layer's set-up."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np, torch
import nethelpers as H
from cchess_zero_amd.net import PolicyValueNet
from cchess_zero_amd.telemetry import ClockProbe
import ctypes as C
from cchess_zero_amd._lib import check, lib
torch.set_grad_enabled(False)
x = torch.from_numpy(H.positions(64, 3)).cuda().repeat(128, 1, 1, 1)
x16 = torch.zeros((8192, 9, 10, 16), dtype=torch.float16, device="cuda")
x16[..., :14] = x.to(torch.float16)
net = PolicyValueNet(7, "cuda:0", torch.float16, seed=0, split="mx")
for _ in range(3):
    net._hip_net_forward(x16)
torch.cuda.synchronize()
pr = ClockProbe(net._hip_ctx(), 8200)
pr.arm()
net._hip_net_forward(x16)
torch.cuda.synchronize()
check(lib().cz_set_clock_probe(net._hip_ctx().h, None, 0), "cz_set_clock_probe")
b = pr.buf[:4096].cpu().numpy().astype(np.float64)
b2 = pr.buf[4096:8192].cpu().numpy().astype(np.float64)
if os.environ.get("CCHESS_MX_KERNEL", "1") != "2" or os.environ.get("CCHESS_MX_TIMING_V1"):     # k_trunk_mx_c128 built with -DMX_TIMING: before the tower | slab loops | barrier + epilogue + barrier | bias / x init + address set-up + operand preload
    tot = b.sum(axis=1)
    print("k_trunk_mx_c128 per workgroup (wave 0): before the tower %.0f | slab loops %.0f (%.1f %%) = %.0f per layer = %.0f per slab | barrier + epilogue + barrier %.0f per layer | init + set-up + preload %.0f per layer | sum %.0f"
          % (b[:, 0].mean(), b[:, 1].mean(), 100 * b[:, 1].mean() / tot.mean(), b[:, 1].mean() / 14, b[:, 1].mean() / 14 / 36, b[:, 2].mean() / 14, b[:, 3].mean() / 14, tot.mean()))
    sys.exit(0)
print("exchange split per layer: drain + bias loads + write tiles 0, 1 + x loads %.0f | barrier %.0f | read + add, barrier, write tile 2 %.0f | x load, barrier %.0f | rest (read tile 2, + bias, + x, barrier) %.0f"
      % (b2[:, 0].mean() / 14, b2[:, 1].mean() / 14, b2[:, 2].mean() / 14, b2[:, 3].mean() / 14, (b[:, 2].mean()) / 14))
tot = b.sum(axis=1)
print("per workgroup (wave 0, lane 0), shader cycles: before the tower %.0f | slab loops %.0f (%.1f %%) | drain + x loads + exchange %.0f (%.1f %%) | epilogue + set-up %.0f (%.1f %%) | sum %.0f"
      % (b[:, 0].mean(), b[:, 1].mean(), 100 * b[:, 1].mean() / tot.mean(), b[:, 2].mean(), 100 * b[:, 2].mean() / tot.mean(), b[:, 3].mean(), 100 * b[:, 3].mean() / tot.mean(), tot.mean()))
print("per layer (14): loops %.0f = %.0f per body (18), exchange %.0f, epilogue %.0f" % (b[:, 1].mean() / 14, b[:, 1].mean() / 14 / 18, b[:, 2].mean() / 14, b[:, 3].mean() / 14))
