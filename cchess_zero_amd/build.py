"""Builds libcchess_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcchess_hip.so")
SOURCES = ["cz_api.hip", "cz_tables.hip", "cz_rules.hip", "cz_search.hip", "cz_selfplay.hip", "cz_conv.hip", "cz_heads.hip", "cz_probe.hip"]
# -ffp-contract=off and correctly rounded f32 divide: the tree statistics are bit-exact
# restatements of the reference's float32/float64 arithmetic.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function"]


def hipcc():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    return "hipcc"


def _deps():
    out = [os.path.join(CSRC, s) for s in SOURCES]
    out += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]   # .inc: generated slab asm
    out.append(os.path.join(os.path.dirname(HERE), "include", "cchess_hip.h"))
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False, extra=()):
    extra = list(extra)
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + FLAGS + extra + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
