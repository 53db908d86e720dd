#!/usr/bin/env python3
"""Reads tools/experiments/bin/mx_probe's dump and checks the hypotheses the MX trunk kernel is built on:
  H1  v_cvt_scalef32_2xpk16_fp6_f32 d, s0, s1, scale: slot 2i = q(s0[i] / 2^floor(log2 scale)), slot 2i+1 = q(s1[i] / ...),
      six-bit E2M3 codes packed little-endian, round-to-nearest-even, saturating at +-7.5 (inf and NaN too)
  H2  v_mfma_scale_f32_32x32x64_f8f6f4 (A = B = fp6): D[m][n] = sum over kb in {0,1}, slot < 32 of
      A[lane m + 32 kb].slot * B[lane n + 32 kb].slot * 2^(sa[m + 32 kb].byte(OA) - 127) * 2^(sb[n + 32 kb].byte(OB) - 127),
      D in the usual 32x32 layout (lane l: column l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5))
      — operand 0 ("A") supplies the ROWS.
    python tools/experiments/mx_probe_check.py gpurun_out/r05a/mx_probe2.bin
"""
import struct, sys
import numpy as np


def sections(path):
    d = open(path, "rb").read()
    o, S = 0, {}
    while o < len(d):
        name = d[o:o + 16].split(b"\0")[0].decode()
        n = struct.unpack("<Q", d[o + 16:o + 24])[0]
        S[name] = d[o + 24:o + 24 + n]
        o += 24 + n
    return S


def dq(c):
    s, e, m = c >> 5, (c >> 3) & 3, c & 7
    v = (1 + m / 8) * 2.0 ** (e - 1) if e else m / 8
    return -v if s else v


GRID = np.array(sorted(set(dq(c) for c in range(64))))


def q(x):
    """RNE onto the E2M3 grid, saturating"""
    if np.isnan(x):
        return 7.5
    x = max(-7.5, min(7.5, x))
    i = np.searchsorted(GRID, x)
    cands = [GRID[j] for j in (i - 1, i) if 0 <= j < len(GRID)]
    best = min(cands, key=lambda g: abs(g - x))
    ties = [g for g in cands if abs(g - x) == abs(best - x)]
    if len(ties) == 2:   # even mantissa code
        code = lambda g: [c for c in range(64) if dq(c) == g and (g != 0 or c in (0, 32))][0]
        best = [g for g in ties if code(g) % 2 == 0][0]
    return best


def fields(row):
    v = 0
    for i in range(6):
        v |= int(row[i]) << (32 * i)
    return [(v >> (6 * i)) & 63 for i in range(32)]


def main():
    S = sections(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05a/mx_probe2.bin")
    cin = np.frombuffer(S["cvt_in"], np.float32).reshape(64, 32)
    sc = np.frombuffer(S["cvt_scale"], np.float32)
    out = np.frombuffer(S["cvt_out"], np.uint32).reshape(64, 6)
    bad = 0
    for l in range(64):
        got = [dq(c) for c in fields(out[l])]
        e = 2.0 ** np.floor(np.log2(sc[l]))
        want = []
        for i in range(16):
            want += [q(cin[l, i] / e), q(cin[l, 16 + i] / e)]
        for i, (g, w) in enumerate(zip(got, want)):
            if g != w and not (g == 0 and w == 0):
                bad += 1
                if bad < 12:
                    print("H1 mismatch lane %d slot %d: in %r scale %r got %r want %r" % (l, i, cin[l, (i >> 1) + 16 * (i & 1)], sc[l], g, w))
    print("H1 (cvt: interleaved slots, scale exponent only, RNE, saturating): %s (%d mismatches of %d)" % ("HOLDS" if bad == 0 else "FAILS", bad, 64 * 32))
    A = np.frombuffer(S["mf_a"], np.float32).reshape(64, 32).astype(np.float64)
    B = np.frombuffer(S["mf_b"], np.float32).reshape(64, 32).astype(np.float64)
    sa = np.frombuffer(S["mf_sa"], np.uint32)
    sb = np.frombuffer(S["mf_sb"], np.uint32)
    for tag, oa, ob in (("mf_d00", 0, 0), ("mf_d10", 1, 0), ("mf_d01", 0, 1), ("mf_d23", 2, 3), ("mf_d32", 3, 2)):
        D = np.frombuffer(S[tag], np.float32).reshape(64, 16)
        ea = 2.0 ** (((sa >> (8 * oa)) & 255).astype(np.float64) - 127)
        eb = 2.0 ** (((sb >> (8 * ob)) & 255).astype(np.float64) - 127)
        want = np.zeros((32, 32))
        for kb in range(2):
            Ak = A[32 * kb:32 * kb + 32] * ea[32 * kb:32 * kb + 32, None]     # rows m
            Bk = B[32 * kb:32 * kb + 32] * eb[32 * kb:32 * kb + 32, None]     # columns n
            want += Ak @ Bk.T
        got = np.zeros((32, 32))
        for l in range(64):
            for r in range(16):
                got[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31] = D[l, r]
        err = np.abs(got - want).max()
        errT = np.abs(got - want.T).max()
        print("H2 op_sel A %d B %d: max |D - expected| = %.3g (max |D| %.3g)%s" % (oa, ob, err, np.abs(want).max(), "" if err < 1e-3 else "   transposed: %.3g" % errT))


if __name__ == "__main__":
    main()
