#!/usr/bin/env python3
"""Generates cchess_zero_amd/csrc/cz_tower_slab_asm.inc: the hand-scheduled slab body of k_tower_c128.

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


def emit(name, lines):
    out = ["#define %s \\" % name]
    for l in lines:
        out.append('    "%s\\n\\t" \\' % l)
    out[-1] = out[-1][:-2]
    return "\n".join(out) + "\n"


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(os.path.dirname(here), "cchess_zero_amd", "csrc", "cz_tower_slab_asm.inc")
    txt = "// GENERATED by tools/gen_tower_asm.py — do not edit.  See that script for the issue plan.\n"
    txt += emit("TW_SLAB_ASM_H0", slab(0)) + "\n" + emit("TW_SLAB_ASM_H1", slab(1))
    txt += "\n// 8-wave / 4-position variant (two fragment sets, 2 DMA pieces per wave)\n"
    txt += emit("TW8_SLAB_ASM_H0", slab8(0)) + "\n" + emit("TW8_SLAB_ASM_H1", slab8(1))
    open(dst, "w").write(txt)
    print("wrote", dst, len(slab(0)), "instructions per slab")


if __name__ == "__main__":
    main()
