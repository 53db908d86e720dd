#!/usr/bin/env python3
"""Generates tools/experiments/cz_experiments_slab_asm.inc: the hand-scheduled slab bodies of the trunk-kernel variants
that were measured and not adopted (cz_trunk_experiments.h).  The product kernels' bodies come from tools/gen_tower_asm.py.

One slab = 64 input channels of one 3x3 tap = 4 k-steps x 6 v_mfma_f32_32x32x16_bf16 per wave.
Issue plan (per wave, everything except the MFMAs sits in an MFMA's ~28-cycle issue shadow):
  k0  MFMAs on fragment set f0 | address math + ds_read_b128 x5 -> f2 (k-step 2 of this slab)
  k1  MFMAs on f1              | ... -> f3 (k-step 3)
  mid s_waitcnt vmcnt(4) ; s_barrier        (slab g+1 published, slab g-1's buffer free)
  k2  MFMAs on f2              | 4 x (m0 = LDS dst ; global_load_lds_dwordx4) of slab g+3, reads -> f0 (next slab k0)
  k3  MFMAs on f3              | reads -> f1 (next slab k1) ; s_waitcnt lgkmcnt(0)
Operands are named; the C++ side (cz_conv_kernel.h) binds them.  H = which 64-channel half of the tap.
"""
import os

ACC = [["c00", "c01"], ["c10", "c11"], ["c20", "c21"]]


def mfma(i, j, Y):
    return "v_mfma_f32_32x32x16_bf16 %%[%s], %%[%sb%d], %%[%sa%d], %%[%s]" % (ACC[i][j], Y, j, Y, i, ACC[i][j])


def kstep(Y, X, CA, OB0, OB1, AB, KEY, VB, wait, extras=None):
    ex = extras or [[], [], [], [], [], []]
    L = []
    if wait is not None:
        L.append("s_waitcnt lgkmcnt(%d)" % wait)
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


def slab(H):
    nab, nkey = ("ab", "key") if H == 0 else ("nab", "nkey")
    L = ["s_mov_b32 %[keep], m0"]
    L += kstep("f0", "f2", H * 8 + 4, 8192, 8704, "ab", "key", "vb", None)
    L += kstep("f1", "f3", H * 8 + 6, 12288, 12800, "ab", "key", "vb", None)
    L += ["s_waitcnt vmcnt(4)", "s_barrier"]
    dma = [["s_mov_b32 m0, %[ldst]", "s_nop 0", "global_load_lds_dwordx4 %[voff0], %[sbase]"],
           ["s_add_u32 m0, %[ldst], 0x1000", "s_nop 0", "global_load_lds_dwordx4 %[voff1], %[sbase]"],
           ["s_add_u32 m0, %[ldst], 0x2000", "s_nop 0", "global_load_lds_dwordx4 %[voff2], %[sbase]"],
           ["s_add_u32 m0, %[ldst], 0x3000", "s_nop 0", "global_load_lds_dwordx4 %[voff3], %[sbase]"], [], []]
    L += kstep("f2", "f0", (H ^ 1) * 8 + 0, 0, 512, nab, nkey, "vbn", 5, dma)
    L += kstep("f3", "f1", (H ^ 1) * 8 + 2, 4096, 4608, nab, nkey, "vbn", 5)
    L += ["s_waitcnt lgkmcnt(0)", "s_mov_b32 m0, %[keep]"]
    return L


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


def slabQ(hs):
    """Two-workgroups-per-CU variant (k_tower2x_c128): 8 KB slabs = 2 k-steps (32 input channels), 4 per tap.
    k0: f0 -> loads f1 (k1) ; vmcnt(2) + barrier ; k1: f1 -> f0 (k0 of the next slab; the next tap's addresses after
    the fourth slab) + 2 DMA pieces."""
    nab, nkey = ("nab", "nkey") if hs == 3 else ("ab", "key")
    L = ["s_mov_b32 %[keep], m0"]
    L += kstep2("f0", "f1", hs * 4 + 2, 4096, 4608, "ab", "key", "vb")
    L += ["s_waitcnt vmcnt(2)", "s_barrier"]
    dma = [["s_mov_b32 m0, %[ldst]", "s_nop 0", "global_load_lds_dwordx4 %[voff0], %[sbase]"], [],
           ["s_add_u32 m0, %[ldst], 0x1000", "s_nop 0", "global_load_lds_dwordx4 %[voff1], %[sbase]"], [], [], []]
    L += kstep2("f1", "f0", ((hs + 1) % 4) * 4, 0, 512, nab, nkey, "vbn", dma)
    L += ["s_mov_b32 m0, %[keep]"]
    return L


def slabSK(hs):
    """Skewed half-workgroup variant (k_towersk_c128): 4 positions / 8 waves, 8 KB slabs in an 8-slot ring.  Waves 0-3 (half A,
    positions 0-1) run SKEW slabs ahead of waves 4-7 (half B, positions 2-3) on the same slab sequence, so that one half's
    layer boundary (VALU) falls under the other half's MFMAs.  Only half A feeds the ring (its four waves move the whole 8 KB
    slab, two 1 KB pieces each, like the 2-position variant); half B skips the DMA with a scalar branch on its wave index %[wv] and finds its
    slabs published by the barriers it shares with A."""
    nab, nkey = ("nab", "nkey") if hs == 3 else ("ab", "key")
    L = ["s_mov_b32 %[keep], m0"]
    L += kstep2("f0", "f1", hs * 4 + 2, 4096, 4608, "ab", "key", "vb")
    L += ["s_waitcnt vmcnt(2)", "s_barrier"]
    dma = [["s_cmp_gt_u32 %[wv], 3", "s_cbranch_scc1 1f", "s_mov_b32 m0, %[ldst]", "s_nop 0",
            "global_load_lds_dwordx4 %[voff0], %[sbase]", "1:"], [],
           ["s_cmp_gt_u32 %[wv], 3", "s_cbranch_scc1 2f", "s_add_u32 m0, %[ldst], 0x1000", "s_nop 0",
            "global_load_lds_dwordx4 %[voff1], %[sbase]", "2:"], [], [], []]
    L += kstep2("f1", "f0", ((hs + 1) % 4) * 4, 0, 512, nab, nkey, "vbn", dma)
    L += ["s_mov_b32 m0, %[keep]"]
    return L


def kstepD(k):
    """Ring-free variant (k_towerd_c128): the weight fragments come straight from global memory (L1 / L2) into registers, so
    there is no LDS weight ring, no DMA and no per-slab barrier.  One macro per k-step, k = 0..23 (three taps = the period of
    the register rotation: activation fragments A0/A1 are requested from LDS one k-step ahead, weight fragments W0/W1/W2 from
    global two k-steps ahead).  k-step k computes on A[k%2], W[k%3], requests A[(k+1)%2] for k+1 (the next tap's row addresses
    when k is a tap's last k-step) and W[(k+2)%3] for k+2 (%[wn] = that k-step's 4 KB of the packed weights)."""
    a, an, w, wn = "A%d" % (k % 2), "A%d" % ((k + 1) % 2), "W%d" % (k % 3), "W%d" % ((k + 2) % 3)
    kk = k % 8
    ab, key = ("nab", "nkey") if kk == 7 else ("ab", "key")
    ca = ((kk + 1) % 8) * 2
    m = lambda i, j: "v_mfma_f32_32x32x16_bf16 %%[%s], %%[%sb%d], %%[%sa%d], %%[%s]" % (ACC[i][j], w, j, a, i, ACC[i][j])
    L = ["s_waitcnt lgkmcnt(0)", "s_waitcnt vmcnt(2)"]
    L.append("v_xor_b32 %%[t0], %d, %%[%s0]" % (ca, key))
    L.append(m(0, 0))
    L.append("v_xor_b32 %%[t1], %d, %%[%s1]" % (ca, key))
    L.append("v_xor_b32 %%[t2], %d, %%[%s2]" % (ca, key))
    L.append("v_lshl_add_u32 %%[t0], %%[t0], 4, %%[%s0]" % ab)
    L.append("v_lshl_add_u32 %%[t1], %%[t1], 4, %%[%s1]" % ab)
    L.append("v_lshl_add_u32 %%[t2], %%[t2], 4, %%[%s2]" % ab)
    L.append(m(0, 1))
    L.append("ds_read_b128 %%[%sa0], %%[t0]" % an)
    L.append("ds_read_b128 %%[%sa1], %%[t1]" % an)
    L.append(m(1, 0))
    L.append("ds_read_b128 %%[%sa2], %%[t2]" % an)
    L.append("global_load_dwordx4 %%[%sb0], %%[voff], %%[wn]" % wn)
    L.append(m(1, 1))
    L.append("global_load_dwordx4 %%[%sb1], %%[voff], %%[wn] offset:512" % wn)
    L.append(m(2, 0))
    L.append(m(2, 1))
    return L


ACCP = [["p%d%d" % (i, j) for j in range(4)] for i in range(3)]


def mfmaP(i, j, Y):
    return "v_mfma_f32_32x32x16_bf16 %%[%s], %%[%sb%d], %%[%sa%d], %%[%s]" % (ACCP[i][j], Y, j, Y, i, ACCP[i][j])


def kstepP(Y, X, CA, OB, AB, KEY, VB, extras=None, exp=0, zero_c=False):
    """Position-per-wave variant (one wave per SIMD, 3 cell tiles x 4 channel tiles = 12 MFMAs per k-step against
    7 fragment reads).  Two complete fragment sets: the k-step waits for its own set (requested during the previous
    k-step), then requests the other set for the next k-step, one ds_read_b128 per MFMA gap (a gap hides about five
    single-issue instructions; clustering the reads, or single-buffering the weight fragments behind counted
    waits, measured 4-7 % slower).  extras (LDS-DMA pieces, vmcnt-counted) go behind MFMAs 8..11."""
    ex = extras or []
    order = [(i, j) for j in range(4) for i in range(3)]
    L = [] if exp & 4 else ["s_waitcnt lgkmcnt(0)"]
    reads = ["ds_read_b128 %%[%sa%d], %%[t%d]" % (X, n, n) for n in range(3)]
    reads += ["ds_read_b128 %%[%sb%d], %%[%s] offset:%d" % (X, j, VB, OB + 512 * j) for j in range(4)]
    if exp & 8:
        reads = []
    for n in range(12):
        L.append(mfmaP(*order[n], Y))
        if n == 0 and not exp & 16:
            for m in range(3):
                L.append("v_xor_b32 %%[t%d], %d, %%[%s%d]" % (m, CA, KEY, m))
            for m in range(3):
                L.append("v_lshl_add_u32 %%[t%d], %%[t%d], 4, %%[%s%d]" % (m, m, AB, m))
        if 1 <= n <= len(reads):
            L.append(reads[n - 1])
        if 8 <= n < 8 + len(ex):
            L += ex[n - 8]
    if zero_c:   # first k-step of a layer: the accumulators start from an inline 0 instead of their old contents
        L = [l[:l.rindex(",")] + ", 0" if l.startswith("v_mfma") else l for l in L]
    return L


def slabP(H, exp=0, first=False):
    """k0: f0 -> loads f1 (k1) ; k1: f1 -> f0 (k2) ; vmcnt(4) + barrier ; k2: f0 -> f1 (k3) + 4 DMA pieces ;
    k3: f1 -> f0 (k0 of the next slab).  The DMA pieces share one lane-offset register; the 4 KB steps are four
    scalar bases (the 13-bit instruction offset cannot hold them)."""
    nab, nkey = ("ab", "key") if H == 0 else ("nab", "nkey")
    L = ["s_mov_b32 %[keep], m0"]
    L += kstepP("f0", "f1", H * 8 + 2, 4096, "ab", "key", "vb", exp=exp, zero_c=first)
    L += kstepP("f1", "f0", H * 8 + 4, 8192, "ab", "key", "vb", exp=exp)
    if not exp & 1:
        L += ["s_waitcnt vmcnt(4)", "s_barrier"]
    dma = [["s_mov_b32 m0, %[ldst]", "s_nop 0", "global_load_lds_dwordx4 %[voff0], %[sbase]"]]
    for q in (1, 2, 3):
        dma.append(["s_add_u32 m0, %%[ldst], 0x%x" % (q * 0x1000), "s_nop 0",
                    "global_load_lds_dwordx4 %%[voff0], %%[sbase%d]" % q])
    L += kstepP("f0", "f1", H * 8 + 6, 12288, "ab", "key", "vb", None if exp & 2 else dma, exp=exp)
    L += kstepP("f1", "f0", (H ^ 1) * 8 + 0, 0, nab, nkey, "vbn", exp=exp)
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
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "cz_experiments_slab_asm.inc")
    txt = "// GENERATED by tools/experiments/gen_experiments_asm.py — do not edit.  See that script for the issue plan.\n"
    txt += emit("TW_SLAB_ASM_H0", slab(0)) + "\n" + emit("TW_SLAB_ASM_H1", slab(1))
    f16 = lambda L: [l.replace("v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16") for l in L]
    txt += "\n// skewed half-workgroup variant (4 positions / 8 waves, 8 KB slabs, 8-slot ring fed by waves 0-3), bf16 and fp16\n"
    for hs in range(4):
        txt += emit("TWS_SLAB_ASM_Q%d" % hs, slabSK(hs)) + "\n" + emit("TWSF_SLAB_ASM_Q%d" % hs, f16(slabSK(hs))) + "\n"
    txt += "\n// ring-free variant: weight fragments from global memory, one macro per k-step of the 24-k-step rotation period, bf16 and fp16\n"
    for k in range(24):
        txt += emit("TWD_KSTEP_%d" % k, kstepD(k)) + "\n" + emit("TWDF_KSTEP_%d" % k, f16(kstepD(k))) + "\n"
    txt += "\n// position-per-wave variant (4 waves, 3 cell tiles x 4 channel tiles each, two fragment sets)\n"
    txt += emit("TWP_SLAB_ASM_H0", slabP(0)) + "\n" + emit("TWP_SLAB_ASM_H1", slabP(1))
    txt += "\n" + emit("TWP_SLAB_ASM_FIRST", slabP(0, first=True))
    # zero-work elasticity experiment (tower_skip_ubench.hip): k_tower8_c128's slab with the MFMAs of the wave's third cell tile
    # removed (a third of the MFMAs; wrong results, timing only) — what does issuing fewer MFMAs buy under the power governor?
    drop = lambda L: [l for l in L if not (l.startswith("v_mfma") and ("%[c20]" in l or "%[c21]" in l))]
    txt += "\n// k_tower8_c128 slab without the third cell tile's MFMAs (timing experiment)\n"
    txt += emit("TW8F_SKIP_ASM_H0", f16(drop(slab8(0)))) + "\n" + emit("TW8F_SKIP_ASM_H1", f16(drop(slab8(1)))) + "\n"
    txt += emit("TW8_SKIP_ASM_H0", drop(slab8(0))) + "\n" + emit("TW8_SKIP_ASM_H1", drop(slab8(1))) + "\n"
    if os.environ.get("CZ_TP_EXP"):   # timing experiments only (wrong results): 1 = no barrier, 2 = no DMA, 4 = no LDS waits, 8 = no fragment reads, 16 = no address math
        e = int(os.environ["CZ_TP_EXP"])
        txt = txt.replace("#define TWP_SLAB_ASM_H", "#define TWP_REAL_SLAB_ASM_H")
        txt += emit("TWP_SLAB_ASM_H0", slabP(0, e)) + "\n" + emit("TWP_SLAB_ASM_H1", slabP(1, e))
        txt += "\n" + emit("TWP_SLAB_ASM_FIRST", slabP(0, e, first=True))
    open(dst, "w").write(txt)
    print("wrote", dst, len(slab(0)), "instructions per slab")


if __name__ == "__main__":
    main()
