#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name and counter -> calls, mean, total.
Usage: pmc_summary.py <dir-or-csv> [top]"""
import csv
import glob
import os
import sys


def main(path, top=25):
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    agg = {}
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = (row["Kernel_Name"], row["Counter_Name"])
                a = agg.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    print("%-100s %-14s %8s %16s %18s" % ("kernel", "counter", "calls", "mean", "total"))
    for (k, c), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-100s %-14s %8d %16.1f %18.1f" % (k[:100], c, n, tot / n, tot))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 25)
