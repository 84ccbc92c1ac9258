#!/usr/bin/env python3
"""One replayed meta-training step from a rocprofv3 kernel_trace csv: the dispatches between two consecutive pack_oihw_multi
launches in the middle of the trace (= one hipGraph replay + the optimizer), with each kernel's duration and the gap since the
previous kernel's end; totals by kernel name.   Usage: metatrain_graph_timeline.py <kernel_trace.csv> [--sequence]"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        wg = max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1), 1)
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(int(r.get("Grid_Size_Y", 1) or 1), 1) * max(int(r.get("Grid_Size_Z", 1) or 1), 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid // wg))
rows.sort()
marks = [i for i, r in enumerate(rows) if "pack_oihw_multi" in r[2]]
# steps alternate: pack (forward) ... adam; take the last complete pair of packs that are a full step apart
starts = [i for k, i in enumerate(marks) if k == 0 or i - marks[k - 1] > 150]          # (two packs per step: backbone, then the head ~70 kernels later)
mid = len(starts) // 2                                   # (the trace ends with bench.py's three eager, event-bracketed steps)
a, b = starts[mid], starts[mid + 1]
step = rows[a:b]
t0, t1 = step[0][0], rows[b][0]
busy = sum(r[1] - r[0] for r in step)
gaps = [step[i][0] - step[i - 1][1] for i in range(1, len(step))]
print("# one replayed step: %d kernels, wall %.1f us (start of this step's first kernel to the next step's), kernel time %.1f us, gaps %.1f us "
      "(%.2f us per boundary; negative = overlap)" % (len(step), (t1 - t0) / 1e3, busy / 1e3, sum(gaps) / 1e3, sum(gaps) / 1e3 / max(len(gaps), 1)))
by = collections.defaultdict(lambda: [0, 0])
for s, e, n, _ in step:
    k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    by[k][0] += 1
    by[k][1] += e - s
print("# by kernel (calls, total us, avg us)")
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %4d %9.1f %8.2f" % (k, c, t / 1e3, t / 1e3 / c))
print("# gap histogram (us): ", {b_: sum(1 for g in gaps if lo <= g / 1e3 < hi) for b_, lo, hi in
                               (("<0", -1e9, 0), ("0-1", 0, 1), ("1-2", 1, 2), ("2-4", 2, 4), ("4-8", 4, 8), (">=8", 8, 1e9))})
if "--sequence" in sys.argv:
    print("# in launch order: us since the step's start, duration us, workgroups, kernel")
    for s, e, n, wgs in step:
        k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        print("%9.1f %8.2f %7d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, wgs, k))
