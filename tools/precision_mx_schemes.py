#!/usr/bin/env python3
"""CPU emulation: split schemes for the strict trunk that cost FEWER than three fp16-MFMA-equivalents
per product and still hold north_star's 1e-3 on trained-like weights at 7 and 19 blocks.

The strict engine (k_trunk_split_c128) computes a*w = a_hi*w_hi + a_hi*w_lo + a_lo*w_hi with fp16 halves: three
v_mfma_f32_32x32x16_f16 (8 passes each) per 16 input channels.  The two CROSS terms are 2^-12 of the product, so their
operands need only a few significant bits: gfx950's block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 runs fp8 (E4M3) at 2x and
fp6 (E2M3) / fp4 (E2M1) at 4x the fp16 rate, with one E8M0 (power of two) scale per lane = per 32 K-elements.  K-order is
free (both operands use the same one), so a lane's 32-element block can hold BOTH cross terms of 16 channels:
[q(a_hi) | q(a_lo * 2^12)] . [q(w_lo * 2^12) | q(w_hi)], scale product 2^-12 * sa * sw.

    cost per 32 input channels, in MFMA passes (fp16 32x32x16 = 8, f8f6f4 32x32x64: fp8 16, fp6/fp4 8):
      strict  3 fp16                       = 48
      mx8     2 fp16 + 1 fp8  (K = 64)     = 32   (2.0 fp16-equivalents per product)
      mx6     2 fp16 + 1 fp6               = 24   (1.5)
      mx6x2   2 fp16 + 2 fp6 (hi operands of the cross terms as two fp6 pieces)  = 32
      mx4     2 fp16 + 1 fp4               = 24
Every variant is fake-quantised (quantise -> dequantise, fp32 conv, fp32 accumulate) and compared with the float64 graph.

    python tools/precision_mx_schemes.py            # -> profiles/r05_precision_mx_schemes.txt
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np, torch, torch.nn.functional as F
import nethelpers as H
from cchess_zero_amd.net import PolicyValueModule
torch.set_num_threads(8)

f16 = torch.float16


def q_e4m3(x):
    return x.clamp(-448, 448).to(torch.float8_e4m3fn).float()


def q_grid(x, mant, emin, vmax):
    """round-to-nearest-even onto a small float grid: `mant` mantissa bits, smallest normal 2^emin (subnormals below),
    saturating at vmax.  E2M3: mant 3, emin 0, vmax 7.5; E2M1: mant 1, emin 0, vmax 6; E3M2: mant 2, emin -2, vmax 28"""
    ax = x.abs().clamp(max=vmax)
    e = torch.floor(torch.log2(ax.clamp(min=2.0 ** emin)))
    step = torch.exp2(e - mant)
    return torch.sign(x) * (torch.round(ax / step) * step).clamp(max=vmax)     # torch.round = half to even


FMT = {"e4m3": (q_e4m3, 448.0), "e2m3": (lambda x: q_grid(x, 3, 0, 7.5), 7.5), "e2m1": (lambda x: q_grid(x, 1, 0, 6.0), 6.0),
       "e3m2": (lambda x: q_grid(x, 2, -2, 28.0), 28.0), "e5m2": (lambda x: x.clamp(-57344, 57344).to(torch.float8_e5m2).float(), 57344.0)}


def mxq(parts, fmt, cdim, blk):
    """Block-scaled quantisation of several tensors that share the lane's 32-element block: `parts` are tensors of one shape
    whose dimension `cdim` (channels) is cut into groups of `blk`; one power-of-two scale per group over ALL parts (amax
    maps below the format's largest value).  Returns the dequantised parts."""
    q, vmax = FMT[fmt]
    sh = list(parts[0].shape)
    C = sh[cdim]
    g = [p.movedim(cdim, -1).reshape(-1, C // blk, blk) for p in parts]
    amax = torch.stack([t.abs().amax(-1) for t in g]).amax(0).clamp(min=1e-30)
    s = torch.exp2(torch.ceil(torch.log2(amax / vmax))).unsqueeze(-1)
    out = []
    for p, t in zip(parts, g):
        d = (q(t / s) * s).reshape(p.movedim(cdim, -1).shape).movedim(-1, cdim)
        out.append(d)
    return out


def conv_scheme(h_hi, h_lo, w, scheme):
    """one conv's products under `scheme`; h_hi + h_lo = the stored activation (fp16 halves), w fp32 folded weights"""
    pad = w.shape[-1] // 2
    cv = lambda a, b: F.conv2d(a, b, None, padding=pad)
    w_hi = w.to(f16).float(); w_lo = (w - w_hi).to(f16).float()
    kind = scheme["kind"]
    if kind == "strict":
        return cv(h_hi, w_hi) + cv(h_hi, w_lo) + cv(h_lo, w_hi)
    if kind == "fp16":
        return cv(h_hi, w_hi)
    fmt, blk = scheme["fmt"], scheme.get("blk", 16)
    S = 4096.0
    if kind == "mx":            # [q(a_hi) | q(a_lo S)] . [q(w_lo S) | q(w_hi)], one scale per (cell / out channel+tap, blk channels)
        if scheme.get("joint", True):
            a1, a2 = mxq([h_hi, h_lo * S], fmt, 1, blk)
            b1, b2 = mxq([w_lo * S, w_hi], fmt, 1, blk)
        else:                   # separate blocks (and scales) per term
            (a1,), (a2,) = mxq([h_hi], fmt, 1, blk), mxq([h_lo * S], fmt, 1, blk)
            (b1,), (b2,) = mxq([w_lo * S], fmt, 1, blk), mxq([w_hi], fmt, 1, blk)
        return cv(h_hi, w_hi) + (cv(a1, b1) + cv(a2, b2)) / S
    if kind == "mx2":           # the HI operand of each cross term as two pieces p + q (second piece has its own block scale)
        (ap,) = mxq([h_hi], fmt, 1, blk); (aq,) = mxq([h_hi - ap], fmt, 1, blk)
        (wp,) = mxq([w_hi], fmt, 1, blk); (wq,) = mxq([w_hi - wp], fmt, 1, blk)
        (al,) = mxq([h_lo * S], fmt, 1, blk); (wl,) = mxq([w_lo * S], fmt, 1, blk)
        return cv(h_hi, w_hi) + (cv(ap, wl) + cv(aq, wl) + cv(al, wp) + cv(al, wq)) / S
    if kind == "mx2lo":         # the LO operands as two pieces instead (representation error of a_lo, w_lo)
        (ap,) = mxq([h_hi], fmt, 1, blk); (wp,) = mxq([w_hi], fmt, 1, blk)
        (al,) = mxq([h_lo * S], fmt, 1, blk); (al2,) = mxq([h_lo * S - al], fmt, 1, blk)
        (wl,) = mxq([w_lo * S], fmt, 1, blk); (wl2,) = mxq([w_lo * S - wl], fmt, 1, blk)
        return cv(h_hi, w_hi) + (cv(ap, wl) + cv(ap, wl2) + cv(al, wp) + cv(al2, wp)) / S
    if kind == "w16x2_a8":      # weights exact in two fp16 MFMAs, activation lo half through fp8 only: a_hi*(w_hi+w_lo) + q(a_lo)q(w_hi)
        (al,) = mxq([h_lo * S], fmt, 1, blk); (wp,) = mxq([w_hi], fmt, 1, blk)
        return cv(h_hi, w_hi) + cv(h_hi, w_lo) + cv(al, wp) / S
    raise KeyError(kind)


def fwd(m, x, scheme):
    def halves(t):
        hi = t.clamp(max=65504.0).to(f16).float()
        if scheme["kind"] == "fp16":
            return hi, torch.zeros_like(hi)
        return hi, (t - hi).to(f16).float()
    def conv(c, hi, lo):
        w, b = c.folded()
        return conv_scheme(hi, lo, w, scheme) + b.view(1, -1, 1, 1)
    z = torch.zeros_like(x)
    sch0 = scheme if scheme["kind"] in ("strict", "fp16") else dict(kind="strict")     # first conv: 0/1 planes are exact, two MFMAs
    w, b = m.conv_in.folded()
    h = torch.relu(conv_scheme(x, z, w, sch0) + b.view(1, -1, 1, 1)); hh, hl = halves(h)
    for a, b2 in m.blocks:
        th, tl = halves(torch.relu(conv(a, hh, hl)))
        y = conv(b2, th, tl) + (hh + hl)
        hh, hl = halves(torch.relu(y))
    hs = hh + hl
    wp, bp = m.policy_conv.folded(); wv, bv = m.value_conv.folded()
    p = torch.relu(F.conv2d(hs, wp, bp)).permute(0, 2, 3, 1).reshape(hs.shape[0], 180)
    v = torch.relu(F.conv2d(hs, wv, bv)).permute(0, 2, 3, 1).reshape(hs.shape[0], 90)
    return m.policy_fc(p), torch.tanh(m.value_fc2(torch.relu(m.value_fc1(v))))


SCHEMES = [
    ("fp16 (fast engine)                1.0", dict(kind="fp16")),
    ("strict: 3 fp16                    3.0", dict(kind="strict")),
    ("mx8  e4m3 joint blk16             2.0", dict(kind="mx", fmt="e4m3", blk=16)),
    ("mx8  e4m3 joint blk32 (K=128/2)   2.0", dict(kind="mx", fmt="e4m3", blk=32)),
    ("mx8  e4m3 per-term blk32          2.0", dict(kind="mx", fmt="e4m3", blk=32, joint=False)),
    ("mx8  e4m3 joint blk128            2.0", dict(kind="mx", fmt="e4m3", blk=128)),
    ("mx8  e5m2 joint blk16             2.0", dict(kind="mx", fmt="e5m2", blk=16)),
    ("mx6  e2m3 joint blk16             1.5", dict(kind="mx", fmt="e2m3", blk=16)),
    ("mx6  e2m3 per-term blk32          1.5", dict(kind="mx", fmt="e2m3", blk=32, joint=False)),
    ("mx6  e3m2 joint blk16             1.5", dict(kind="mx", fmt="e3m2", blk=16)),
    ("mx4  e2m1 joint blk16             1.5", dict(kind="mx", fmt="e2m1", blk=16)),
    ("mx6x2 e2m3 hi operands 2 pieces   2.0", dict(kind="mx2", fmt="e2m3", blk=32)),
    ("mx6x2 e2m3 lo operands 2 pieces   2.0", dict(kind="mx2lo", fmt="e2m3", blk=32)),
    ("mx8x2 e4m3 hi operands 2 pieces   3.0", dict(kind="mx2", fmt="e4m3", blk=32)),
    ("w fp16x2, a_lo via e4m3 (K=64)    2.5", dict(kind="w16x2_a8", fmt="e4m3", blk=32)),
    ("w fp16x2, a_lo via e2m3           2.25", dict(kind="w16x2_a8", fmt="e2m3", blk=32)),
]


class _N:
    pass


def main():
    only = sys.argv[1:]
    for blocks in (7, 19):
        for wset in ("trained_like", "glorot"):
            for seed in ((1, 2), (3, 7)) if wset == "trained_like" else ((1, 2),):
                n = _N(); n.module = PolicyValueModule(blocks, seed=seed[0]); n.refresh = lambda: None
                H.WEIGHT_SETS[wset](n)
                x = torch.from_numpy(H.positions(64, seed[1])).permute(0, 3, 1, 2).contiguous()
                with torch.no_grad():
                    m64 = PolicyValueModule(blocks, seed=seed[0]).double()
                    m64.load_state_dict({k: v.double() for k, v in n.module.state_dict().items()})
                    l64, v64 = m64(x.double())
                    l32, v32 = n.module(x)
                    print("== %d blocks %s (module seed %d, positions seed %d): max|logit| %.3g; fp32 torch graph vs float64: %.2g / %.2g"
                          % (blocks, wset, seed[0], seed[1], l64.abs().max(), (l32.double() - l64).abs().max(), (v32.double() - v64).abs().max()), flush=True)
                    for name, sch in SCHEMES:
                        if only and not any(o in name for o in only):
                            continue
                        l, v = fwd(n.module, x, sch)
                        dl, dv = (l.double() - l64).abs().max(), (v.double() - v64).abs().max()
                        print("  %-40s dlogit %.3g  dvalue %.3g   %s" % (name, dl, dv, "ok" if max(dl, dv) <= 1e-3 else "FAILS 1e-3"), flush=True)


if __name__ == "__main__":
    main()
