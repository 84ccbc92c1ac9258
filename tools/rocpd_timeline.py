#!/usr/bin/env python3
"""Timeline of a rocprofv3 rocpd kernel trace around the middle of the run: for N consecutive inner steps (anchored on the
dominant kernel's trunk.7.C2 launch) every kernel with start / duration / queue, and per queue the gaps between consecutive kernels.
    rocpd_timeline.py <db> [steps]"""
import sqlite3
import sys


def main(path, n_steps=2):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = "name, start, end" + (", %s" % qcol if qcol else "")
    rows = cur.execute("select %s from kernels order by start" % sel).fetchall()
    big = [i for i, r in enumerate(rows) if "wgrad_adam_fwd_kernel<2" in r[0]]
    if not big:                                   # unfused path: every third launch of the rows kernel is trunk.7.C2's
        anchors = [i for i, r in enumerate(rows) if "wgrad_adam_rows_kernel" in r[0]]
        med = sorted(rows[i][2] - rows[i][1] for i in anchors)[len(anchors) // 2]
        big = [i for i in anchors if rows[i][2] - rows[i][1] > 1.3 * med]
    mid = big[len(big) // 2]
    nxt = big[len(big) // 2 + n_steps]
    t0, t1 = rows[mid][2], rows[nxt][2]
    print("# window: %d inner steps, %.1f us (%.1f us per step)" % (n_steps, (t1 - t0) / 1e3, (t1 - t0) / 1e3 / n_steps))
    last_end = {}
    busy = {}
    for r in rows:
        name, s, e = r[0], r[1], r[2]
        q = r[3] if qcol else 0
        if e <= t0 or s >= t1:
            if e <= t0:
                last_end[q] = e
            continue
        gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
        last_end[q] = e
        busy[q] = busy.get(q, 0) + (min(e, t1) - max(s, t0))
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:58]
        print("q%-3s +%9.1f us  dur %8.1f us  gap before %7.1f us  %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, short))
    for q, b in sorted(busy.items()):
        print("# queue %s busy %.1f us of %.1f (%.1f %%)" % (q, b / 1e3, (t1 - t0) / 1e3, 100.0 * b / (t1 - t0)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
