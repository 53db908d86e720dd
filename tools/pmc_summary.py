#!/usr/bin/env python3
"""Summarise the two PMC passes of tools/pmc_ubench.sh for the tower kernel: effective clock, MFMA pipe
utilisation, per-wave cycle split and LDS conflict rate."""
import collections
import csv
import os
import sys

KSUB = os.environ.get("PMC_KERNEL", "tower")          # "trunk_split" for the strict engine's kernel
WG_P = int(os.environ.get("PMC_WG_POSITIONS", "4"))   # positions per workgroup (k_trunk_split_c128: 2)
SLABS = int(os.environ.get("PMC_SLABS_PER_LAYER", "18"))   # (k_trunk_split_c128: 36)


def load(d):
    c, dur = collections.defaultdict(list), []
    for r in csv.DictReader(open(d + "/p_counter_collection.csv")):
        if KSUB in r["Kernel_Name"]:
            c[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in c.items()}, (sum(dur) / len(dur) if dur else 0.0)


def main():
    a, dur = load(sys.argv[1])
    b, _ = load(sys.argv[2])
    B, nb = int(sys.argv[3]), int(sys.argv[4])
    slabs = (B / WG_P) * 2 * nb * SLABS    # workgroup-slabs per launch
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
    if len(sys.argv) > 5:   # machine-readable copy for profiles/
        import json
        label = sys.argv[6] if len(sys.argv) > 6 else "tools/ubench/tower_base %d positions, %d blocks, random bf16 data" % (B, nb)
        out = {"kernel": "%s (%s)" % ("k_trunk_split_c128" if KSUB == "trunk_split" else "k_tower8_c128", label),
               "method": "rocprofv3 --kernel-trace --pmc, two separate passes (tools/pmc_ubench.sh): a = GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU "
                         "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES; b = SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM "
                         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles",
               "kernel_us": dur / 1e3, "effective_clock_GHz": clk,
               "mfma_busy_frac_of_simd_cycles": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur * clk * 1024),
               "per_workgroup_slab": {"wave_cycles": wc / slabs, "active_inst": a["SQ_ACTIVE_INST_ANY"] * 4 / slabs,
                                      "wait_inst_any": a["SQ_WAIT_INST_ANY"] * 4 / slabs, "wait_any": a["SQ_WAIT_ANY"] * 4 / slabs,
                                      "mfma_busy": a["SQ_VALU_MFMA_BUSY_CYCLES"] / slabs, "insts_valu": a["SQ_INSTS_VALU"] / slabs,
                                      "insts_salu": a["SQ_INSTS_SALU"] / slabs, "insts_lds": b["SQ_INSTS_LDS"] / slabs,
                                      "insts_vmem": b["SQ_INSTS_VMEM"] / slabs, "lds_idx_active": b["SQ_LDS_IDX_ACTIVE"] / slabs,
                                      "wait_inst_lds": b["SQ_WAIT_INST_LDS"] * 4 / slabs},
               "lds_bank_conflict_frac": b["SQ_LDS_BANK_CONFLICT"] / b["SQ_LDS_IDX_ACTIVE"],
               "algorithmic_TFLOPs": 2.0 * B * 90 * 1152 * 128 * 2 * nb / (dur * 1e-9) / 1e12,
               "raw_counters": {**a, **b}}
        json.dump(out, open(sys.argv[5], "w"), indent=1)


if __name__ == "__main__":
    main()
