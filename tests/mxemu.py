"""CPU emulation of k_trunk_mx_c128's arithmetic (cchess_zero_amd/csrc/cz_trunk_mx.h) — test infrastructure.

Every product of a tower conv is  a*w ~= a_hi*w_hi (fp16 MFMA)  +  2^-11 * [ q6(a_hi) q6(2^11 w_lo) + q6(2^11 a_lo) q6(w_hi) ]
(one block-scaled fp6 MFMA): hi = rn16(x), lo = x - hi, q6 = E2M3 (RNE, saturating at 7.5) under a power-of-two scale shared
by the 32 slots of a block = both cross-term operands of 16 channels {32 t + 8 q + 4 h + i} of one cell (activations) / one
(output channel, tap) (weights); scale = 2^(exponent(amax) - 2), so amax lands in [4, 8).  The block input of a residual
block is added in fp32; the heads read the last layer's fp32 values.  Accumulation order differs from the MFMA's (fp32
either way), so the kernel is held to this emulation with a small tolerance, and both to the fp32 graph with north_star's.
Used by tests/test_net.py, tools/precision_mx_schemes.py and the packing test of cchess_zero_amd.net.mx_pack_layer."""
import numpy as np
import torch
import torch.nn.functional as F

f16 = torch.float16
S_LO = 2048.0          # 2^11: |2^11 lo| <= |hi|


def group_perm(C=128):
    """channel order in which the 16 channels of group g = 2 (c // 32) + ((c % 8) // 4) are contiguous, in slot order
    j = 4 q + i (the accumulator register order of the lane that owns them)"""
    order = []
    for t in range(C // 32):
        for h in range(2):
            for q in range(4):
                for i in range(4):
                    order.append(32 * t + 8 * q + 4 * h + i)
    return torch.tensor(order)


def q_e2m3(x):
    """RNE onto the E2M3 grid (step 1/8 below 2, 1/4 below 4, 1/2 below 8), saturating at +-7.5"""
    ax = x.abs().clamp(max=7.5)
    e = torch.floor(torch.log2(ax.clamp(min=1.0)))
    step = torch.exp2(e - 3)
    return torch.sign(x) * (torch.round(ax / step) * step).clamp(max=7.5)


def block_scale(amax):
    """2^(biased exponent(amax) - 2 - 127), the E8M0 byte clamped at 1 as the kernel does"""
    m, e = torch.frexp(amax.clamp(min=1e-45))                      # amax = m 2^e, m in [0.5, 1)
    byte = (e - 1 + 127 - 2).clamp(min=1)
    return torch.exp2((byte - 127).float()), byte


def mxq_pair(p0, p1, cdim, scale_from=None):
    """p0, p1: the two halves of the blocks (same shape), channels on `cdim` in NATURAL order; one scale per 16-channel
    group from max(|p0|, |p1|) (or from `scale_from`).  -> dequantised (p0, p1), natural order."""
    perm = group_perm(p0.shape[cdim]).to(p0.device)
    inv = torch.argsort(perm)
    a = p0.index_select(cdim, perm).movedim(cdim, -1)
    b = p1.index_select(cdim, perm).movedim(cdim, -1)
    sh = a.shape
    a = a.reshape(-1, sh[-1] // 16, 16); b = b.reshape(-1, sh[-1] // 16, 16)
    if scale_from is None:
        amax = torch.maximum(a.abs().amax(-1), b.abs().amax(-1))
    else:
        amax = scale_from.index_select(cdim, perm).movedim(cdim, -1).reshape(-1, sh[-1] // 16, 16).abs().amax(-1)
    s, _ = block_scale(amax)
    s = s.unsqueeze(-1)
    da = (q_e2m3(a / s) * s).reshape(sh).movedim(-1, cdim).index_select(cdim, inv)
    db = (q_e2m3(b / s) * s).reshape(sh).movedim(-1, cdim).index_select(cdim, inv)
    return da, db


def split16(t):
    hi = t.to(f16).float()
    return hi, t - hi


def mx_conv(v, w):
    """one 3x3 tower conv the way the kernel computes it.  v: post-ReLU fp32 activations [B,128,9,10] (clamped at 65504),
    w: folded fp32 weights [128,128,3,3]"""
    a_hi, a_lo = split16(v)
    w_hi, w_lo = split16(w)
    w_lo = w_lo.to(f16).float()            # the host stores lo as fp16 before it is quantised (as the strict engine's pack)
    a_h6, a_l6 = mxq_pair(a_hi, a_lo * S_LO, 1, scale_from=v)
    w_l6, w_h6 = mxq_pair(w_lo * S_LO, w_hi, 1)
    y = F.conv2d(a_hi, w_hi, None, padding=1)
    y = y + (F.conv2d(a_h6, w_l6, None, padding=1) + F.conv2d(a_l6, w_h6, None, padding=1)) / S_LO
    return y


def forward_mx(m, x_nchw):
    """PolicyValueModule m, planes [B,14,9,10] f32 -> (logits, value) as the mx engine computes them"""
    relu = lambda t: t.clamp(min=0.0, max=65504.0)
    w, b = m.conv_in.folded()
    w_hi, w_lo = split16(w)
    w_lo = w_lo.to(f16).float()
    v = relu(F.conv2d(x_nchw, w_hi, None, padding=1) + F.conv2d(x_nchw, w_lo, None, padding=1) + b.view(1, -1, 1, 1))
    for a, b2 in m.blocks:
        wa, ba = a.folded(); wb, bb = b2.folded()
        t = relu(mx_conv(v, wa) + ba.view(1, -1, 1, 1))
        v = relu(mx_conv(t, wb) + bb.view(1, -1, 1, 1) + v)
    wp, bp = m.policy_conv.folded(); wv, bv = m.value_conv.folded()
    p = torch.relu(F.conv2d(v, wp, bp)).permute(0, 2, 3, 1).reshape(v.shape[0], 180)
    u = torch.relu(F.conv2d(v, wv, bv)).permute(0, 2, 3, 1).reshape(v.shape[0], 90)
    return m.policy_fc(p), torch.tanh(m.value_fc2(torch.relu(m.value_fc1(u)))), v


def e2m3_codes(x_scaled):
    """already-scaled values -> 6-bit codes (sign, 2 exponent, 3 mantissa bits), RNE, saturating"""
    q = q_e2m3(x_scaled)
    a = q.abs()
    e = torch.floor(torch.log2(a.clamp(min=1.0))).clamp(max=2)
    normal = a >= 1.0
    mant = torch.where(normal, (a / torch.exp2(e) - 1.0) * 8.0, a * 8.0).round().to(torch.int64)
    ebits = torch.where(normal, e.to(torch.int64) + 1, torch.zeros_like(mant))
    sign = (torch.signbit(x_scaled)).to(torch.int64)
    return (sign << 5) | (ebits << 3) | mant


def pack_slots(codes):
    """[..., 32] six-bit codes -> [..., 24] bytes, slot i at bits 6 i .. 6 i + 5 (little endian)"""
    c = codes.to(torch.int64)
    out = torch.zeros(c.shape[:-1] + (24,), dtype=torch.int64, device=c.device)
    for i in range(32):
        bit = 6 * i
        byte, sh = bit // 8, bit % 8
        v = c[..., i] << sh
        out[..., byte] |= v & 255
        if sh > 2:
            out[..., byte + 1] |= (v >> 8) & 255
    return out.to(torch.uint8)
