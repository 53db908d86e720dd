#!/usr/bin/env python3
"""Summarise the two PMC passes of tools/pmc_ubench.sh for the tower kernel: effective clock, MFMA pipe
utilisation, per-wave cycle split and LDS conflict rate."""
import collections
import csv
import sys


def load(d):
    c, dur = collections.defaultdict(list), []
    for r in csv.DictReader(open(d + "/p_counter_collection.csv")):
        if "tower" in r["Kernel_Name"]:
            c[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in c.items()}, (sum(dur) / len(dur) if dur else 0.0)


def main():
    a, dur = load(sys.argv[1])
    b, _ = load(sys.argv[2])
    B, nb = int(sys.argv[3]), int(sys.argv[4])
    slabs = (B / 4) * 2 * nb * 18          # workgroup-slabs per launch (4 positions per workgroup)
    clk = a["GRBM_GUI_ACTIVE"] / 8 / dur   # GHz; GRBM_GUI_ACTIVE sums the 8 XCDs
    wc = a["SQ_WAVE_CYCLES"] * 4           # quad-cycles -> cycles
    mfma_total = B * 90 / 32 * 4 * 2 * nb * 72  # 32x32x16 MFMAs incl. no padding
    print("kernel %.1f us, effective clock %.2f GHz" % (dur / 1e3, clk))
    print("MFMA busy / (SIMD-cycles available) = %.3f" % (a["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur * clk * 1024)))
    print("per workgroup-slab: wave-cycles %.0f, active %.0f, wait_inst %.0f, wait_any %.0f, mfma_busy %.0f"
          % (wc / slabs, a["SQ_ACTIVE_INST_ANY"] * 4 / slabs, a["SQ_WAIT_INST_ANY"] * 4 / slabs,
             a["SQ_WAIT_ANY"] * 4 / slabs, a["SQ_VALU_MFMA_BUSY_CYCLES"] / slabs))
    print("VALU %.1f SALU %.1f LDS %.1f VMEM %.2f instructions per workgroup-slab"
          % (a["SQ_INSTS_VALU"] / slabs, a["SQ_INSTS_SALU"] / slabs, b["SQ_INSTS_LDS"] / slabs, b["SQ_INSTS_VMEM"] / slabs))
    print("LDS bank conflict / idx active = %.3f ; LDS idx active per slab %.0f ; wait_inst_lds per slab %.0f"
          % (b["SQ_LDS_BANK_CONFLICT"] / b["SQ_LDS_IDX_ACTIVE"], b["SQ_LDS_IDX_ACTIVE"] / slabs, b["SQ_WAIT_INST_LDS"] * 4 / slabs))
    print("raw:", {k: "%.4g" % v for k, v in {**a, **b}.items()})


if __name__ == "__main__":
    main()
