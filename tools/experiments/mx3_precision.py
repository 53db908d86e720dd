#!/usr/bin/env python3
"""CPU emulation for the 4-positions-per-workgroup variant of the mx engine (round 6 design study): the a_hi half of an fp6
block is converted from the lane's fp16 fragments in registers (channels F(kh) = {8 kh .. 8 kh + 7, 16 + 8 kh .. 16 + 8 kh + 7} of a
32-channel quarter), the a_lo half comes from LDS in the epilogue lane's order (channels E(kh) = {8 q + 4 kh + i}), so the block's
scale must be shared by the whole quarter: ONE scale per (cell, 32 channels) instead of per (cell, 16 channels).  Weights: one
scale per (co, tap, quarter, kh) over [2^11 w_lo on F(kh) | w_hi on E(kh)].  How much does the coarser scale cost?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np, torch, torch.nn.functional as F
import nethelpers as H, mxemu
from mxemu import q_e2m3, block_scale, split16, S_LO
from cchess_zero_amd.net import PolicyValueNet
torch.set_num_threads(8)

c = torch.arange(128)
Fgrp = (c % 16) // 8        # kh of the fragment set a channel belongs to
Egrp = (c % 8) // 4         # kh of the epilogue set


def q_with(x, s):
    return q_e2m3(x / s) * s


def mx3_conv(v, w):
    a_hi, a_lo = split16(v)
    w_hi, w_lo = split16(w)
    w_lo = w_lo.to(torch.float16).float()
    B = v.shape[0]
    # activations: scale per (cell, quarter) from max |v| over the 32 channels
    amax = v.abs().reshape(B, 4, 32, 9, 10).amax(2)
    sa, _ = block_scale(amax)
    sa = sa.unsqueeze(2).expand(B, 4, 32, 9, 10).reshape(B, 128, 9, 10)
    a_h6, a_l6 = q_with(a_hi, sa), q_with(a_lo * S_LO, sa)
    # weights [co][ci][3][3]: scale per (co, tap, quarter, kh) over [w_lo S on F(kh) | w_hi on E(kh)]
    wl, wh = (w_lo * S_LO).abs(), w_hi.abs()
    sw_l = torch.zeros_like(w); sw_h = torch.zeros_like(w)
    for Q in range(4):
        for kh in range(2):
            fm = ((c // 32) == Q) & (Fgrp == kh)
            em = ((c // 32) == Q) & (Egrp == kh)
            m = torch.maximum(wl[:, fm].amax(1), wh[:, em].amax(1))          # [co][3][3]
            s, _ = block_scale(m)
            sw_l[:, fm] = s.unsqueeze(1)
            sw_h[:, em] = s.unsqueeze(1)
    w_l6, w_h6 = q_with(w_lo * S_LO, sw_l), q_with(w_hi, sw_h)
    y = F.conv2d(a_hi, w_hi, None, padding=1)
    return y + (F.conv2d(a_h6, w_l6, None, padding=1) + F.conv2d(a_l6, w_h6, None, padding=1)) / S_LO


def forward(m, x, conv):
    relu = lambda t: t.clamp(min=0.0, max=65504.0)
    w, b = m.conv_in.folded()
    w_hi, w_lo = split16(w); w_lo = w_lo.to(torch.float16).float()
    v = relu(F.conv2d(x, w_hi, None, padding=1) + F.conv2d(x, w_lo, None, padding=1) + b.view(1, -1, 1, 1))
    for a, b2 in m.blocks:
        wa, ba = a.folded(); wb, bb = b2.folded()
        t = relu(conv(v, wa) + ba.view(1, -1, 1, 1))
        v = relu(conv(t, wb) + bb.view(1, -1, 1, 1) + v)
    wp, bp = m.policy_conv.folded(); wv, bv = m.value_conv.folded()
    p = torch.relu(F.conv2d(v, wp, bp)).permute(0, 2, 3, 1).reshape(v.shape[0], 180)
    u = torch.relu(F.conv2d(v, wv, bv)).permute(0, 2, 3, 1).reshape(v.shape[0], 90)
    return m.policy_fc(p), torch.tanh(m.value_fc2(torch.relu(m.value_fc1(u))))


x = torch.from_numpy(H.positions(64, 2)).permute(0, 3, 1, 2).contiguous()
with torch.no_grad():
    for blocks, wset in ((7, "trained_like"), (8, "trained_like"), (7, "glorot"), (3, "structured")):
        net = PolicyValueNet(blocks, "cpu", torch.float32, seed=1, backend="torch")
        H.WEIGHT_SETS[wset](net)
        m = net.module
        lr, vr = m.double()(x.double()); m.float()
        for name, conv in (("mx6 (scale per 16 channels, today)", mxemu.mx_conv), ("mx6 scale per 32-channel quarter", mx3_conv)):
            l, v = forward(m, x, conv)
            print("%d blocks %-12s %-40s dlogit %.3g dvalue %.3g (max|logit| %.3g)" % (blocks, wset, name, float((l.double() - lr).abs().max()),
                  float((v.double() - vr).abs().max()), float(lr.abs().max())), flush=True)
