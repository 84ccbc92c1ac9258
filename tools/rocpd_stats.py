#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite (.db) kernel trace: per-kernel calls / total / avg / min / max / %.
(rocprofv3 on this image writes rocpd databases; this prints the same table `--stats` would in CSV mode.)"""
import sqlite3
import sys


def main(path, top=0):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for name, s, e in rows:
        d = e - s
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    span = max(r[2] for r in rows) - min(r[1] for r in rows)
    print("# %d dispatches, sum of kernel time %.3f ms, first-start..last-end span %.3f ms" % (len(rows), tot / 1e6, span / 1e6))
    print("%-110s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top or None]:      # (top = 0: every kernel -- a trace kept under profiles/ is never cut)
        print("%-110s %8d %12.3f %10.2f %10.2f %10.2f %6.2f" % (name[:110], a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
