#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (CSV):
util = MFMA-busy cycles (summed over the 1024 SIMDs) / (GPU-active cycles x 1024).  On this rocprofv3 GRBM_GUI_ACTIVE is
reported summed over the 8 XCDs (the LDS-patch stem kernel: 53.4 M "cycles" for a 2.5 ms launch = 8 x 6.7 M), so it is divided
by 8; cross-check: the stem kernel's 77 % agrees with its measured 108 of 157 TFLOP/s (95 % useful MFMA steps, 98 % useful
rows).  Usage: pmc_mfma_util.py <dir> [top]"""
import csv, glob, os, sys


def main(path, top=14):
    agg = {}
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                a = agg.setdefault(row["Kernel_Name"], {"n": 0})
                a[row["Counter_Name"]] = a.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    a["n"] += 1
    print("%-90s %7s %14s %16s %8s" % ("kernel", "calls", "gpu_cycles/call", "mfma_busy/call", "util"))
    rows = sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:top]
    for k, a in rows:
        n = max(a["n"], 1)
        act, busy = a.get("GRBM_GUI_ACTIVE", 0.0), a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        print("%-90s %7d %14.0f %16.0f %7.1f%%" % (k[:90], n, act / 8 / n, busy / n, 100.0 * busy / (act / 8 * 1024) if act else 0.0))
    # all kernels of the pass together: MFMA-busy SIMD-cycles over GPU-active SIMD-cycles, i.e. the time-weighted utilisation
    act = sum(a.get("GRBM_GUI_ACTIVE", 0.0) for a in agg.values())
    busy = sum(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for a in agg.values())
    conv = {k: a for k, a in agg.items() if ("conv" in k or "stem" in k or "skinny" in k)}
    act_c = sum(a.get("GRBM_GUI_ACTIVE", 0.0) for a in conv.values())
    busy_c = sum(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for a in conv.values())
    if act:
        print("ALL %d kernels of the run, time-weighted: %.1f %% MFMA-busy; convolution kernels only (%.0f %% of the GPU-active cycles): %.1f %%"
              % (len(agg), 100.0 * busy / (act / 8 * 1024), 100.0 * act_c / act, 100.0 * busy_c / (act_c / 8 * 1024) if act_c else 0.0))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 14)
