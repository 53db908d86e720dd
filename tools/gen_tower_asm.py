#!/usr/bin/env python3
"""Generates the hand-scheduled slab bodies of the product trunk kernels:

  cchess_zero_amd/csrc/cz_tower_slab_asm.inc   k_tower8_c128 (cz_conv_kernel.h): 4 positions / 8 waves, 16-bit operands
  cchess_zero_amd/csrc/cz_trunk_split_asm.inc  k_trunk_split_c128 (cz_trunk_split.h): 2 positions / 8 waves, every
                                               operand split into two 16-bit halves, three MFMAs per product
  cchess_zero_amd/csrc/cz_trunk_mx_asm.inc     k_trunk_mx_c128 (cz_trunk_mx.h): the same tiling, fp16 hi halves + both cross
                                               terms on one block-scaled fp6 MFMA (1.5 MFMA-equivalents per product)

k_tower8_c128: one slab = 64 input channels of one 3x3 tap = 4 k-steps x 6 v_mfma_f32_32x32x16 per wave; two fragment
sets; each k-step first waits for its own set (the partner wave on the SIMD covers the wait), then interleaves its 6
MFMAs with the 5 ds_read_b128 of the next k-step; the LDS-DMA pieces of slab g+3 go out behind the barrier in the middle
of slab g.  Everything except the MFMAs sits in an MFMA's issue shadow.  Operands are named; the C++ side binds them.
(The bodies of the variants that were measured and not adopted: tools/experiments/gen_experiments_asm.py.)
"""
import os

ACC = [["c00", "c01"], ["c10", "c11"], ["c20", "c21"]]


def mfma(i, j, Y):
    return "v_mfma_f32_32x32x16_bf16 %%[%s], %%[%sb%d], %%[%sa%d], %%[%s]" % (ACC[i][j], Y, j, Y, i, ACC[i][j])


def kstep2(Y, X, CA, OB0, OB1, AB, KEY, VB, extras=None):
    """8-wave variant: two fragment sets, each k-step first waits for its own set (the partner wave on the SIMD
    covers the wait), then interleaves its 6 MFMAs with the 5 reads of the next k-step."""
    ex = extras or [[], [], [], [], [], []]
    L = ["s_waitcnt lgkmcnt(0)"]
    L.append("v_xor_b32 %%[t0], %d, %%[%s0]" % (CA, KEY))
    L.append(mfma(0, 0, Y)); L += ex[0]
    L.append("v_xor_b32 %%[t1], %d, %%[%s1]" % (CA, KEY))
    L.append("v_xor_b32 %%[t2], %d, %%[%s2]" % (CA, KEY))
    L.append("v_lshl_add_u32 %%[t0], %%[t0], 4, %%[%s0]" % AB)
    L.append("v_lshl_add_u32 %%[t1], %%[t1], 4, %%[%s1]" % AB)
    L.append("v_lshl_add_u32 %%[t2], %%[t2], 4, %%[%s2]" % AB)
    L.append(mfma(0, 1, Y)); L += ex[1]
    L.append("ds_read_b128 %%[%sa0], %%[t0]" % X)
    L.append("ds_read_b128 %%[%sa1], %%[t1]" % X)
    L.append(mfma(1, 0, Y)); L += ex[2]
    L.append("ds_read_b128 %%[%sa2], %%[t2]" % X)
    L.append("ds_read_b128 %%[%sb0], %%[%s] offset:%d" % (X, VB, OB0))
    L.append(mfma(1, 1, Y)); L += ex[3]
    L.append("ds_read_b128 %%[%sb1], %%[%s] offset:%d" % (X, VB, OB1))
    L.append(mfma(2, 0, Y)); L += ex[4]
    L.append(mfma(2, 1, Y)); L += ex[5]
    return L


def slab8(H):
    """k0: f0 -> loads f1 (k1) ; k1: f1 -> f0 (k2) ; mid ; k2: f0 -> f1 (k3) + 2 DMA pieces ; k3: f1 -> f0 (next slab k0)."""
    nab, nkey = ("ab", "key") if H == 0 else ("nab", "nkey")
    L = ["s_mov_b32 %[keep], m0"]
    L += kstep2("f0", "f1", H * 8 + 2, 4096, 4608, "ab", "key", "vb")
    L += kstep2("f1", "f0", H * 8 + 4, 8192, 8704, "ab", "key", "vb")
    L += ["s_waitcnt vmcnt(2)", "s_barrier"]
    dma = [["s_mov_b32 m0, %[ldst]", "s_nop 0", "global_load_lds_dwordx4 %[voff0], %[sbase]"], [],
           ["s_add_u32 m0, %[ldst], 0x2000", "s_nop 0", "global_load_lds_dwordx4 %[voff1], %[sbase]"], [], [], []]
    L += kstep2("f0", "f1", H * 8 + 6, 12288, 12800, "ab", "key", "vb", dma)
    L += kstep2("f1", "f0", (H ^ 1) * 8 + 0, 0, 512, nab, nkey, "vbn")
    L += ["s_mov_b32 m0, %[keep]"]
    return L


def kstepX(Y, X, CA, OB, AB, KEY, VB, LO, extras=None):
    """Split-operand variant (k_trunk_split_c128): a wave owns 3 cell tiles x 1 channel tile.  Fragment set = activation
    hi/lo halves of the 3 cell tiles (ah0-2, al0-2; the lo half of a row lives LO bytes behind its hi half) + the weight
    tile's hi/lo halves (wh, wl).  9 MFMAs per k-step: c_i += wh*ah_i ; c_i += wl*ah_i ; c_i += wh*al_i (lo*lo is below
    fp32 resolution of the sum and is dropped) — the three products of one accumulator are three MFMAs apart.  The 8 reads of
    the next k-step's set are spread over the first six MFMA shadows; extras (LDS-DMA pieces) follow."""
    ex = extras or [[] for _ in range(9)]
    m = lambda i, w, a: "v_mfma_f32_32x32x16_bf16 %%[c%d], %%[%s%s], %%[%s%s%d], %%[c%d]" % (i, Y, w, Y, a, i, i)
    L = ["s_waitcnt lgkmcnt(0)"]
    L.append("v_xor_b32 %%[t0], %d, %%[%s0]" % (CA, KEY))
    L.append(m(0, "wh", "ah")); L += ex[0]
    L.append("v_xor_b32 %%[t1], %d, %%[%s1]" % (CA, KEY))
    L.append("v_xor_b32 %%[t2], %d, %%[%s2]" % (CA, KEY))
    L.append("v_lshl_add_u32 %%[t0], %%[t0], 4, %%[%s0]" % AB)
    L.append("v_lshl_add_u32 %%[t1], %%[t1], 4, %%[%s1]" % AB)
    L.append("v_lshl_add_u32 %%[t2], %%[t2], 4, %%[%s2]" % AB)
    L.append(m(1, "wh", "ah")); L += ex[1]
    L.append("ds_read_b128 %%[%sah0], %%[t0]" % X)
    L.append("ds_read_b128 %%[%sal0], %%[t0] offset:%d" % (X, LO))
    L.append(m(2, "wh", "ah")); L += ex[2]
    L.append("ds_read_b128 %%[%sah1], %%[t1]" % X)
    L.append("ds_read_b128 %%[%sal1], %%[t1] offset:%d" % (X, LO))
    L.append(m(0, "wl", "ah")); L += ex[3]
    L.append("ds_read_b128 %%[%sah2], %%[t2]" % X)
    L.append("ds_read_b128 %%[%sal2], %%[t2] offset:%d" % (X, LO))
    L.append(m(1, "wl", "ah")); L += ex[4]
    L.append("ds_read_b128 %%[%swh], %%[%s] offset:%d" % (X, VB, OB))
    L.append("ds_read_b128 %%[%swl], %%[%s] offset:%d" % (X, VB, OB + 8192))
    L.append(m(2, "wl", "ah")); L += ex[5]
    L.append(m(0, "wh", "al")); L += ex[6]
    L.append(m(1, "wh", "al")); L += ex[7]
    L.append(m(2, "wh", "al")); L += ex[8]
    return L


def slabX(hs, LO):
    """One 16 KB slab = 32 input channels of one tap, hi halves (8 KB) then lo halves (8 KB) = 2 k-steps, 4 slabs per tap.
    k0: f0 -> loads f1 (k1) ; vmcnt(2) + barrier (slab g+1 published, slab g-1's buffer free) ; k1: f1 -> loads f0 (k0 of the
    next slab; the next tap's addresses after the fourth slab) + the wave's 2 DMA pieces of slab g+3."""
    nab, nkey = ("nab", "nkey") if hs == 3 else ("ab", "key")
    L = ["s_mov_b32 %[keep], m0"]
    L += kstepX("f0", "f1", hs * 4 + 2, 4096, "ab", "key", "vb", LO)
    L += ["s_waitcnt vmcnt(2)", "s_barrier"]
    dma = [[] for _ in range(9)]
    dma[5] = ["s_mov_b32 m0, %[ldst]", "s_nop 0", "global_load_lds_dwordx4 %[voff0], %[sbase]"]
    dma[7] = ["s_add_u32 m0, %[ldst], 0x2000", "s_nop 0", "global_load_lds_dwordx4 %[voff1], %[sbase]"]
    L += kstepX("f1", "f0", ((hs + 1) % 4) * 4, 0, nab, nkey, "vbn", LO, dma)
    L += ["s_mov_b32 m0, %[keep]"]
    return L


XS_LO_OFF = 181 * 256   # cz_trunk_split.h: XSGeo::LO_OFF (180 activation rows + the zero row)

# ---- k_trunk_mx_c128 (cz_trunk_mx.h): a*w = a_hi*w_hi on two fp16 MFMAs per 32 input channels + BOTH cross terms of those 32
# channels on ONE block-scaled fp6 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, 8 passes): 9 MFMAs per slab and wave instead of 18.
MX_XPLANE, MX_YPLANE = 224 * 16, 224 * 8      # MXGeo: one plane per 16-channel group, 224 entries (180 cells + the zero aliases)
MX_Y_OFF, MX_S_OFF = 8 * MX_XPLANE, 8 * MX_XPLANE + 8 * MX_YPLANE
MX_AX = ["v[232:237]", "v[238:243]", "v[244:249]"]   # fp6 activation blocks of the three cell tiles: loaded and consumed inside
MX_WX = "v[250:255]"                                  # one slab body, so they are hard registers (the body clobbers v232..v255)


def slabMX(hs, layer_last=False):
    """One 16 KB slab = 32 input channels of one tap: [fp16 hi: 2 k-steps x 2 halves x 128 co x 8][fp6 blocks: X 2 x 128 x 16 B,
    Y 2 x 128 x 8 B][E8M0 scale dwords 2 x 128].  Three steps of three MFMAs; operands are requested TWO steps ahead:
      A (k-step 0, set a0*/w0)  waits for its set (the next step's 4 reads may be in flight), requests this slab's fp6 set
      B (k-step 1, set a1*/w1)  vmcnt(2) + barrier first (slab g+1 published, slab g-1's buffer free), requests the NEXT slab's
                                set A (after the fourth slab of a tap: at the next tap's addresses)
      C (fp6, hard registers)   requests the next slab's set B, issues the wave's 2 DMA pieces of slab g+3.
    layer_last (round 6): the LAST slab of a layer requests no operands (the activations are about to be replaced) and waits
    for all of its LDS reads in front of its barrier — behind it no wave reads the activation planes any more, so the epilogue
    needs no barrier of its own in front."""
    nab, nkey = ("nab", "nkey") if hs == 3 else ("ab", "key")
    ca = ((hs + 1) % 4) * 4
    f = lambda i, w, a: "v_mfma_f32_32x32x16_f16 %%[c%d], %%[%s], %%[%s%d], %%[c%d]" % (i, w, a, i, i)
    mx = lambda i: ("v_mfma_scale_f32_32x32x64_f8f6f4 %%[c%d], %s, %s, %%[c%d], %%[ws], %%[sb%d] op_sel:[0,%d,0] op_sel_hi:[0,%d,0] cbsz:2 blgp:2"
                    % (i, MX_WX, MX_AX[i], i, i, hs & 1, hs >> 1))
    sub = lambda r, lo, hi: "v[%d:%d]" % (int(r[2:].split(":")[0]) + lo, int(r[2:].split(":")[0]) + hi)
    xo, yo = hs * 2 * MX_XPLANE, MX_Y_OFF + hs * 2 * MX_YPLANE
    L = ["s_mov_b32 %[keep], m0"]
    # step A
    L += ["s_waitcnt lgkmcnt(4)", f(0, "w0", "a0h")]
    L += ["ds_read_b128 %s, %%[xr0] offset:%d" % (sub(MX_AX[0], 0, 3), xo), "ds_read_b64 %s, %%[yr0] offset:%d" % (sub(MX_AX[0], 4, 5), yo),
          "ds_read_b128 %s, %%[xr1] offset:%d" % (sub(MX_AX[1], 0, 3), xo), "ds_read_b64 %s, %%[yr1] offset:%d" % (sub(MX_AX[1], 4, 5), yo)]
    if hs == 0:
        L += ["v_lshrrev_b32 %[t0], 1, %[yr0]", "v_lshrrev_b32 %[t1], 1, %[yr1]", "v_lshrrev_b32 %[t2], 1, %[yr2]"]
    L += [f(1, "w0", "a0h")]
    L += ["ds_read_b128 %s, %%[xr2] offset:%d" % (sub(MX_AX[2], 0, 3), xo), "ds_read_b64 %s, %%[yr2] offset:%d" % (sub(MX_AX[2], 4, 5), yo)]
    nc = 9
    if hs == 0:   # the tap's activation scales: one dword per cell = the four 32-channel quarters' E8M0 bytes of this lane's half
        L += ["ds_read_b32 %%[sb%d], %%[t%d] offset:%d" % (i, i, MX_S_OFF) for i in range(3)]
        nc = 12
    L += [f(2, "w0", "a0h")]
    L += ["ds_read_b128 %s, %%[vb] offset:8192" % sub(MX_WX, 0, 3), "ds_read_b64 %s, %%[vy]" % sub(MX_WX, 4, 5), "ds_read_b32 %[ws], %[vs]"]
    # step B
    if layer_last:
        L += ["s_waitcnt vmcnt(2) lgkmcnt(0)", "s_barrier"]
        L += [f(0, "w1", "a1h"), f(1, "w1", "a1h"), f(2, "w1", "a1h")]
    else:
        L += ["s_waitcnt vmcnt(2)", "s_barrier", "s_waitcnt lgkmcnt(%d)" % nc]
        L += [f(0, "w1", "a1h")]
        L += ["v_xor_b32 %%[t%d], %d, %%[%s%d]" % (i, ca, nkey, i) for i in range(3)]
        L += ["v_lshl_add_u32 %%[t%d], %%[t%d], 4, %%[%s%d]" % (i, i, nab, i) for i in range(3)]
        L += [f(1, "w1", "a1h")]
        L += ["ds_read_b128 %[a0h0], %[t0]", "ds_read_b128 %[a0h1], %[t1]"]
        L += [f(2, "w1", "a1h")]
        L += ["ds_read_b128 %[a0h2], %[t2]", "ds_read_b128 %[w0], %[vbn]"]
    # step C
    dma1 = ["s_mov_b32 m0, %[ldst]", "s_nop 0", "global_load_lds_dwordx4 %[voff0], %[sbase]"]
    dma2 = ["s_add_u32 m0, %[ldst], 0x2000", "s_nop 0", "global_load_lds_dwordx4 %[voff1], %[sbase]"]
    place = os.environ.get("MX_DMA_PLACE", "C")     # experiment knob: where the wave's two LDS-DMA pieces go
    if place == "B0":       # both right behind the barrier
        i = L.index("s_barrier") + 1
        L[i:i] = dma1 + dma2
    elif place == "B0C0":   # one behind the barrier, one at the top of step C
        i = L.index("s_barrier") + 1
        L[i:i] = dma1
    L += [] if layer_last else ["s_waitcnt lgkmcnt(4)"]
    if place == "B0C0":
        L += dma2
    L += [mx(0)]
    if not layer_last:
        L += ["v_xor_b32 %%[t%d], %d, %%[%s%d]" % (i, ca + 2, nkey, i) for i in range(3)]
        L += ["v_lshl_add_u32 %%[t%d], %%[t%d], 4, %%[%s%d]" % (i, i, nab, i) for i in range(3)]
    if place == "C":
        L += dma1
    L += [mx(1)]
    L += [] if layer_last else ["ds_read_b128 %[a1h0], %[t0]", "ds_read_b128 %[a1h1], %[t1]"]
    if place == "C":
        L += dma2
    L += [mx(2)]
    L += [] if layer_last else ["ds_read_b128 %[a1h2], %[t2]", "ds_read_b128 %[w1], %[vbn] offset:4096"]
    if place == "Cend":
        L += dma1 + dma2
    L += ["s_mov_b32 m0, %[keep]"]
    # timing ablations (tools/experiments/mx_ablate.sh; wrong results): MX_ABLATE = comma list of nolgkmwait, novmwait, nobarrier, bar2 (barrier in
    # every other slab), noc (no fp6 operand reads), nomx (no fp6 MFMAs), nodma, nohi (no fp16 operand reads)
    ab = [a for a in os.environ.get("MX_ABLATE", "").split(",") if a]
    if "nobarrier" in ab or ("bar2" in ab and hs % 2 == 1):
        L = [l for l in L if l != "s_barrier"]
    if "noc" in ab:
        L = [l for l in L if not (l.startswith("ds_read") and ("v[2" in l or "%[ws]" in l or "%[sb" in l))]
        L = [l.replace("lgkmcnt(12)", "lgkmcnt(0)").replace("lgkmcnt(9)", "lgkmcnt(0)") for l in L]
    if "nohi" in ab:
        L = [l for l in L if not (l.startswith("ds_read_b128 %[a") or l.startswith("ds_read_b128 %[w"))]
        L = [l.replace("lgkmcnt(4)", "lgkmcnt(0)") for l in L]
    if "nomx" in ab:
        L = [l for l in L if not l.startswith("v_mfma_scale")]
    if "nolgkmwait" in ab:    # operands are never waited for: what perfect latency hiding of the LDS reads would buy (same LDS traffic)
        L = [l for l in L if not l.startswith("s_waitcnt lgkmcnt")]
    if "novmwait" in ab:      # the DMAs are issued but never waited for: issue cost vs waiting for the data
        L = [l.replace("s_waitcnt vmcnt(2)", "s_nop 0") for l in L]
    if "nodma" in ab:
        L = [l for l in L if not l.startswith("global_load_lds")]
        L = [l.replace("s_waitcnt vmcnt(2)", "s_nop 0") for l in L]
    return L


# ---- k_trunk_mx2_c128 (cz_trunk_mx2.h, round 6): the same arithmetic, LDS layout and slab format as k_trunk_mx_c128 with a 3 x 2
# register tile per wave and the K range of a layer split between the two waves of a pair (wave & 1 = kp handles the slabs
# 2 p + kp): 18 MFMAs per body instead of 9 with 20 + 5 LDS reads per body instead of 2 x (12 + 4) per 18 MFMAs, one barrier
# per PAIR of slabs, the partial sums exchanged once per layer.
MX2_AX = ["v[226:231]", "v[232:237]", "v[238:243]"]   # fp6 activation blocks of the three cell tiles
MX2_WX = ["v[244:249]", "v[250:255]"]                 # fp6 weight blocks of the two channel tiles (the body clobbers v226..v255)


def slabMX2(first, layer_first=False, layer_last=False):
    """One body = one 16 KB slab (32 input channels of one tap) for a 3 cell x 2 channel tile: 12 fp16 MFMAs + 6 fp6 MFMAs.
    A wave sees two bodies per tap: `first` (quarter kp; the next body is quarter kp + 2 of the SAME tap: ab / key) and second
    (quarter kp + 2; the next body is quarter kp of the NEXT tap: nab / nkey) — the C++ side passes the right address operands
    as nab / nkey, and the quarter-dependent numbers as scalars (cba / cbb: 16-byte chunk of the next body's two k-steps; xo / yo:
    plane offsets of this body's fp6 blocks; kp8 = 8 kp: the tap's scale dword is shifted so that bytes 0 / 2 are this wave's
    quarters).  Channel tile j = 1 of the wave is 32 output channels = 512 / 256 / 128 bytes behind tile j = 0 in a slab's parts.
      A (k-step 0, set a0h* / w0*)  waits for its set (the 5 reads of set B may be in flight); requests this body's 12 fp6
                                    operand reads (+ 3 scale dwords in the first body of a tap)
      B (k-step 1, set a1h* / w1*)  every read of this wave has returned (lgkmcnt 0: the pair the DMAs below overwrite is no
                                    longer being read by anyone behind the barrier), its 4 DMA pieces have landed (vmcnt 0), barrier:
                                    the next pair is published; requests the next body's set A; 2 DMA pieces of the pair after next
      C (fp6, hard registers)       requests the next body's set B; the other 2 DMA pieces.
    layer_first: the first body of a layer — its DMAs were waited for when the previous layer drained, and the only vector-memory
    operations in flight are the epilogue's block-input stores, which the barrier need not wait for (no vmcnt wait).
    layer_last: the last body of a layer requests no operands (there is no next body: the activations are about to be replaced),
    so behind its barrier no wave reads the activation planes any more and the exchange may start without another barrier."""
    f = lambda i, j, s: "v_mfma_f32_32x32x16_f16 %%[c%d%d], %%[w%d%d], %%[a%dh%d], %%[c%d%d]" % (i, j, s, j, s, i, i, j)
    mx = lambda i, j: ("v_mfma_scale_f32_32x32x64_f8f6f4 %%[c%d%d], %s, %s, %%[c%d%d], %%[ws%d], %%[sb%d] op_sel:[0,0,0] op_sel_hi:[0,%d,0] cbsz:2 blgp:2"
                       % (i, j, MX2_WX[j], MX2_AX[i], i, j, j, i, 0 if first else 1))
    sub = lambda r, lo, hi: "v[%d:%d]" % (int(r[2:].split(":")[0]) + lo, int(r[2:].split(":")[0]) + hi)
    L = ["s_mov_b32 %[keep], m0"]
    # ---- step A: the reads of THIS pair's buffers (the fp6 weight blocks and their scales) go out first, so that the counted
    # wait in front of the barrier covers them while the activation blocks may still be in flight
    L += ["s_waitcnt lgkmcnt(5)"]
    wreads = [["ds_read_b128 %s, %%[vb] offset:%d" % (sub(MX2_WX[j], 0, 3), 8192 + 512 * j), "ds_read_b64 %s, %%[vy] offset:%d" % (sub(MX2_WX[j], 4, 5), 256 * j),
               "ds_read_b32 %%[ws%d], %%[vs] offset:%d" % (j, 128 * j)] for j in range(2)]
    areads = [["v_add_u32 %%[t0], %%[xo], %%[xr%d]" % i, "v_add_u32 %%[t1], %%[yo], %%[yr%d]" % i,
               "ds_read_b128 %s, %%[t0]" % sub(MX2_AX[i], 0, 3), "ds_read_b64 %s, %%[t1]" % sub(MX2_AX[i], 4, 5)] for i in range(3)]
    sreads = (["v_lshrrev_b32 %%[t%d], 1, %%[yr%d]" % (k, k) for k in range(3)] +
              ["ds_read_b32 %%[sb%d], %%[t%d] offset:%d" % (k, k, MX_S_OFF) for k in range(3)]) if first else []
    fill = [wreads[0], wreads[1], areads[0], areads[1], areads[2], sreads]
    k = 0
    for i in range(3):
        for j in range(2):
            L += [f(i, j, 0)] + fill[k]
            k += 1
    behind_w = 6 + (3 if first else 0)      # reads issued behind the last read of the weight ring
    # ---- step B
    L += ["s_waitcnt lgkmcnt(%d)" % behind_w if layer_first else "s_waitcnt vmcnt(0) lgkmcnt(%d)" % (0 if layer_last else behind_w), "s_barrier"]
    # piece k of the wave: 1 KB at byte 8192 k + 16 tid of the 32 KB pair, to the same offset of the pair's two ring buffers; all four
    # right behind the barrier: they have the whole body to land (the next barrier waits for them)
    dma = [["s_mov_b32 m0, %[ldst]", "s_nop 0", "global_load_lds_dwordx4 %[voff0], %[sbase]"]]
    dma += [["s_add_u32 m0, %%[ldst], 0x%x" % (0x2000 * k), "v_add_u32 %%[t3], 0x%x, %%[voff0]" % (0x2000 * k),
             "global_load_lds_dwordx4 %[t3], %[sbase]"] for k in (1, 2, 3)]
    place = os.environ.get("MX2_DMA_PLACE", "B")

    def half(s, cb, woff, d, pre=None):
        """6 MFMAs of operand set s (fp16 k-step 1, or the fp6 step when s is None) with the requests of the next body's other
        set interleaved: address VALU, 5 reads; d: DMA pieces behind MFMAs 0..3"""
        X = 0 if s == 1 else 1          # the set being requested
        m = (lambda i, j: f(i, j, 1)) if s == 1 else (lambda i, j: mx(i, j))
        o = list(pre or [])
        rq = not layer_last
        o += [m(0, 0)] + d[0]
        o += ["v_xor_b32 %%[t%d], %%[%s], %%[nkey%d]" % (k, cb, k) for k in range(3)] if rq else []
        o += [m(0, 1)] + d[1]
        o += ["v_lshl_add_u32 %%[t%d], %%[t%d], 4, %%[nab%d]" % (k, k, k) for k in range(3)] if rq else []
        o += [m(1, 0)] + d[2]
        o += ["ds_read_b128 %%[a%dh0], %%[t0]" % X, "ds_read_b128 %%[a%dh1], %%[t1]" % X] if rq else []
        o += [m(1, 1)] + d[3]
        o += ["ds_read_b128 %%[a%dh2], %%[t2]" % X, "ds_read_b128 %%[w%d0], %%[vbn] offset:%d" % (X, woff)] if rq else []
        o += [m(2, 0)]
        o += ["ds_read_b128 %%[w%d1], %%[vbn] offset:%d" % (X, woff + 512)] if rq else []
        o += [m(2, 1)]
        return o
    none = [[], [], [], []]
    if place == "B":        # one piece behind each of step B's first four MFMAs
        L += half(1, "cba", 0, dma)
        dC = none
    else:                   # "BC": two in step B, two in step C (the first version: 36 % slower than k_trunk_mx_c128)
        L += half(1, "cba", 0, [dma[0], [], dma[1], []])
        dC = [dma[2], [], dma[3], []]
    # ---- step C
    pre = ["s_waitcnt lgkmcnt(%d)" % (0 if layer_last else 5)] + (["v_lshrrev_b32 %%[sb%d], %%[kp8], %%[sb%d]" % (k, k) for k in range(3)] if first else [])
    L += half(None, "cbb", 4096, dC, pre)
    L += ["s_mov_b32 m0, %[keep]"]
    ab = [a for a in os.environ.get("MX_ABLATE", "").split(",") if a]
    if "nodma" in ab:
        L = [l for l in L if not l.startswith("global_load_lds")]
    if "nobarrier" in ab:
        L = [l for l in L if l != "s_barrier"]
    return L


# ---- k_trunk_mx12_c128 (tools/experiments/cz_trunk_mx12.h, round 6): k_trunk_mx_c128's arithmetic, layout and slab format with TWELVE
# waves per workgroup (three per SIMD), two cell tiles per wave: the same MFMAs per slab and SIMD, a third wave to cover the waits.
MX12_AX = ["v[144:149]", "v[150:155]"]      # fp6 activation blocks of the two cell tiles
MX12_WX = "v[156:161]"                      # (the body clobbers v144..v161: below the 168 registers a wave of three per SIMD has)


def slabMX12(hs, pieces, layer_last=False):
    """slabMX for two cell tiles per wave: three steps of TWO MFMAs.  pieces: LDS-DMA pieces the wave issues per slab (waves 0-3: 2,
    waves 4-11: 1 — sixteen 1 KB pieces over twelve waves); its vmcnt wait lets exactly the previous body's pieces stay in flight."""
    nab, nkey = ("nab", "nkey") if hs == 3 else ("ab", "key")
    ca = ((hs + 1) % 4) * 4
    f = lambda i, w, a: "v_mfma_f32_32x32x16_f16 %%[c%d], %%[%s], %%[%s%d], %%[c%d]" % (i, w, a, i, i)
    mx = lambda i: ("v_mfma_scale_f32_32x32x64_f8f6f4 %%[c%d], %s, %s, %%[c%d], %%[ws], %%[sb%d] op_sel:[0,%d,0] op_sel_hi:[0,%d,0] cbsz:2 blgp:2"
                    % (i, MX12_WX, MX12_AX[i], i, i, hs & 1, hs >> 1))
    sub = lambda r, lo, hi: "v[%d:%d]" % (int(r[2:].split(":")[0]) + lo, int(r[2:].split(":")[0]) + hi)
    xo, yo = hs * 2 * MX_XPLANE, MX_Y_OFF + hs * 2 * MX_YPLANE
    L = ["s_mov_b32 %[keep], m0"]
    # step A
    L += ["s_waitcnt lgkmcnt(3)", f(0, "w0", "a0h")]
    L += ["ds_read_b128 %s, %%[xr0] offset:%d" % (sub(MX12_AX[0], 0, 3), xo), "ds_read_b64 %s, %%[yr0] offset:%d" % (sub(MX12_AX[0], 4, 5), yo),
          "ds_read_b128 %s, %%[xr1] offset:%d" % (sub(MX12_AX[1], 0, 3), xo), "ds_read_b64 %s, %%[yr1] offset:%d" % (sub(MX12_AX[1], 4, 5), yo)]
    nc = 7
    if hs == 0:
        L += ["v_lshrrev_b32 %[t0], 1, %[yr0]", "v_lshrrev_b32 %[t1], 1, %[yr1]"]
        L += ["ds_read_b32 %%[sb%d], %%[t%d] offset:%d" % (i, i, MX_S_OFF) for i in range(2)]
        nc = 9
    L += [f(1, "w0", "a0h")]
    L += ["ds_read_b128 %s, %%[vb] offset:8192" % sub(MX12_WX, 0, 3), "ds_read_b64 %s, %%[vy]" % sub(MX12_WX, 4, 5), "ds_read_b32 %[ws], %[vs]"]
    # step B
    if layer_last:
        L += ["s_waitcnt vmcnt(%d) lgkmcnt(0)" % pieces, "s_barrier", f(0, "w1", "a1h"), f(1, "w1", "a1h")]
    else:
        L += ["s_waitcnt vmcnt(%d)" % pieces, "s_barrier", "s_waitcnt lgkmcnt(%d)" % nc]
        L += [f(0, "w1", "a1h")]
        L += ["v_xor_b32 %%[t%d], %d, %%[%s%d]" % (i, ca, nkey, i) for i in range(2)]
        L += ["v_lshl_add_u32 %%[t%d], %%[t%d], 4, %%[%s%d]" % (i, i, nab, i) for i in range(2)]
        L += [f(1, "w1", "a1h")]
        L += ["ds_read_b128 %[a0h0], %[t0]", "ds_read_b128 %[a0h1], %[t1]", "ds_read_b128 %[w0], %[vbn]"]
    # step C
    dma1 = ["s_mov_b32 m0, %[ldst]", "s_nop 0", "global_load_lds_dwordx4 %[voff0], %[sbase]"]
    dma2 = ["s_add_u32 m0, %[ldst], 0x3000", "s_nop 0", "global_load_lds_dwordx4 %[voff1], %[sbase]"] if pieces == 2 else []
    L += [] if layer_last else ["s_waitcnt lgkmcnt(3)"]
    L += [mx(0)]
    if not layer_last:
        L += ["v_xor_b32 %%[t%d], %d, %%[%s%d]" % (i, ca + 2, nkey, i) for i in range(2)]
        L += ["v_lshl_add_u32 %%[t%d], %%[t%d], 4, %%[%s%d]" % (i, i, nab, i) for i in range(2)]
    L += dma1 + dma2
    L += [mx(1)]
    L += [] if layer_last else ["ds_read_b128 %[a1h0], %[t0]", "ds_read_b128 %[a1h1], %[t1]", "ds_read_b128 %[w1], %[vbn] offset:4096"]
    L += ["s_mov_b32 m0, %[keep]"]
    return L


def emit(name, lines):
    out = ["#define %s \\" % name]
    for l in lines:
        out.append('    "%s\\n\\t" \\' % l)
    out[-1] = out[-1][:-2]
    return "\n".join(out) + "\n"


def main():
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(here), "cchess_zero_amd", "csrc")
    f16 = lambda L: [l.replace("v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16") for l in L]
    txt = "// GENERATED by tools/gen_tower_asm.py — do not edit.  See that script for the issue plan.\n"
    txt += "// k_tower8_c128: 8 waves / 4 positions, two fragment sets, 2 DMA pieces per wave and slab\n"
    txt += emit("TW8_SLAB_ASM_H0", slab8(0)) + "\n" + emit("TW8_SLAB_ASM_H1", slab8(1))
    txt += "\n// the same with fp16 operands\n"
    txt += emit("TW8F_SLAB_ASM_H0", f16(slab8(0))) + "\n" + emit("TW8F_SLAB_ASM_H1", f16(slab8(1)))
    # the dy = -1 taps: a cell group whose FIRST row tile holds only rank-0 cells and padding rows (groups 0 and 1) reads the zero
    # row for all of that tile's inputs and does not issue its two MFMAs per k-step: a scalar branch around each of them on VCC
    # (set from the wave-uniform 32-bit %[skipm] at the top of the body; nothing in the body writes VCC).  One body for all waves —
    # a wave-dependent choice between two bodies makes hipcc copy / spill the accumulators where the paths join.
    def branchy(L, accs=("%[c00],", "%[c01],")):
        out = ["s_cmp_lg_u32 %[skipm], 0", "s_cselect_b64 vcc, -1, 0"]
        for l in L:
            if l.startswith("v_mfma") and any(l.split()[1] == a for a in accs):
                out += ["s_cbranch_vccnz 1f", l, "1:"]
            else:
                out.append(l)
        return out
    # the general form: bit 0 of %[skipm] skips the wave's FIRST row tile (c00, c01: branch on VCC, set once at the top), bit 1 its
    # SECOND (c10, c11: s_bitcmp1 + branch on SCC in front of each MFMA; nothing between the two reads or writes SCC)
    def branchy2(L):
        out = ["s_bitcmp1_b32 %[skipm], 0", "s_cselect_b64 vcc, -1, 0"]
        for l in L:
            acc = l.split()[1] if l.startswith("v_mfma") else ""
            if acc in ("%[c00],", "%[c01],"):
                out += ["s_cbranch_vccnz 1f", l, "1:"]
            elif acc in ("%[c10],", "%[c11],"):
                out += ["s_bitcmp1_b32 %[skipm], 1", "s_cbranch_scc1 2f", l, "2:"]
            else:
                out.append(l)
        return out
    txt += "\n// slab bodies whose first / second row-tile MFMAs sit behind scalar branches (off-board tiles of a tap), bf16 and fp16\n"
    txt += emit("TW8_SKIPG_ASM_H0", branchy2(slab8(0))) + "\n" + emit("TW8_SKIPG_ASM_H1", branchy2(slab8(1))) + "\n"
    txt += emit("TW8F_SKIPG_ASM_H0", f16(branchy2(slab8(0)))) + "\n" + emit("TW8F_SKIPG_ASM_H1", f16(branchy2(slab8(1))))
    open(os.path.join(csrc, "cz_tower_slab_asm.inc"), "w").write(txt)
    txt = "// GENERATED by tools/gen_tower_asm.py — do not edit.  See that script for the issue plan.\n"
    txt += "// k_trunk_split_c128: 8 waves / 2 positions, operands split into hi + lo halves, 9 MFMAs per k-step, bf16 and fp16\n"
    for hs in range(4):
        txt += emit("XS_SLAB_ASM_Q%d" % hs, slabX(hs, XS_LO_OFF)) + "\n" + emit("XSF_SLAB_ASM_Q%d" % hs, f16(slabX(hs, XS_LO_OFF))) + "\n"
    # dy = -1 taps of the strict kernel: the first row tile of cell group 0 (12 padding rows + the 20 rank-0 cells of both positions)
    txt += "// slab bodies whose first-row-tile MFMAs (3 of 9 per k-step) sit behind a scalar branch (dy = -1 taps)\n"
    for hs in range(4):
        b = branchy(slabX(hs, XS_LO_OFF), ("%[c0],",))
        txt += emit("XS_SKIP0_ASM_Q%d" % hs, b) + "\n" + emit("XSF_SKIP0_ASM_Q%d" % hs, f16(b)) + "\n"
    open(os.path.join(csrc, "cz_trunk_split_asm.inc"), "w").write(txt)
    txt = "// GENERATED by tools/gen_tower_asm.py — do not edit.  See that script for the issue plan.\n"
    txt += "// k_trunk_mx_c128: 8 waves / 2 positions, 6 fp16 MFMAs + 3 block-scaled fp6 MFMAs per slab and wave\n"
    for hs in range(4):
        txt += emit("MX_SLAB_ASM_Q%d" % hs, slabMX(hs)) + "\n"
    txt += "// the same with the first-row-tile MFMAs (3 of 9) behind a scalar branch (dy = -1 taps, cell group 0)\n"
    for hs in range(4):
        txt += emit("MX_SKIP0_ASM_Q%d" % hs, branchy(slabMX(hs), ("%[c0],",))) + "\n"
    txt += "// the last slab of a layer: no operand requests, every LDS read of the wave waited for in front of the barrier\n"
    txt += emit("MX_SLAB_ASM_Q3_LAST", slabMX(3, layer_last=True)) + "\n"
    open(os.path.join(csrc, "cz_trunk_mx_asm.inc"), "w").write(txt)
    txt = "// GENERATED by tools/gen_tower_asm.py — do not edit.  See that script for the issue plan.\n"
    txt += "// k_trunk_mx2_c128: 8 waves / 2 positions, 3 x 2 tiles per wave, K split between the waves of a pair: 12 fp16 + 6 fp6 MFMAs per body\n"
    for name, first in (("A", True), ("B", False)):
        txt += emit("MX2_BODY_%s" % name, slabMX2(first)) + "\n"
        txt += emit("MX2_SKIP0_%s" % name, branchy(slabMX2(first), ("%[c00],", "%[c01],"))) + "\n"
    txt += "// the first body of a layer (tap 0: with the skip; no vmcnt wait) and the last (no operand requests)\n"
    txt += emit("MX2_SKIP0_A_FIRST", branchy(slabMX2(True, layer_first=True), ("%[c00],", "%[c01],"))) + "\n"
    txt += emit("MX2_BODY_B_LAST", slabMX2(False, layer_last=True)) + "\n"
    # an experiment (tools/experiments/cz_trunk_mx2.h; built by tools/experiments/mx_ablate.sh with -DCZ_EXPERIMENT_MX2): written next
    # to that header unless an output directory was given (then into it, as for the ablation builds)
    open(os.path.join(csrc if len(sys.argv) > 1 else os.path.join(here, "experiments"), "cz_trunk_mx2_asm.inc"), "w").write(txt)
    txt = "// GENERATED by tools/gen_tower_asm.py — do not edit.  See that script for the issue plan.\n"
    txt += "// k_trunk_mx12_c128: 12 waves / 2 positions, 2 cell tiles per wave: 4 fp16 + 2 fp6 MFMAs per slab and wave\n"
    for hs in range(4):
        txt += emit("MX12_SKIP0_P2_Q%d" % hs, branchy(slabMX12(hs, 2), ("%[c0],",))) + "\n"    # waves 0-3: two DMA pieces, dy = -1 skip of tile 0
        txt += emit("MX12_SLAB_P2_Q%d" % hs, slabMX12(hs, 2)) + "\n"
        txt += emit("MX12_SLAB_P1_Q%d" % hs, slabMX12(hs, 1)) + "\n"
    txt += emit("MX12_SLAB_P2_Q3_LAST", slabMX12(3, 2, layer_last=True)) + "\n" + emit("MX12_SLAB_P1_Q3_LAST", slabMX12(3, 1, layer_last=True)) + "\n"
    open(os.path.join(csrc if len(sys.argv) > 1 else os.path.join(here, "experiments"), "cz_trunk_mx12_asm.inc"), "w").write(txt)
    print("wrote cz_tower_slab_asm.inc (%d instructions per slab), cz_trunk_split_asm.inc (%d), cz_trunk_mx_asm.inc (%d), cz_trunk_mx2_asm.inc (%d per body)" %
          (len(slab8(0)), len(slabX(0, XS_LO_OFF)), len(slabMX(1)), len(slabMX2(False))))


if __name__ == "__main__":
    main()
