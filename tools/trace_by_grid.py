#!/usr/bin/env python3
"""Development aid: a rocprofv3 kernel trace (kernel_trace.csv) of ONE engine's decode loop, per (kernel, grid): launches,
mean duration, and the mean gap between a launch's start and the end of the launch before it on the same queue.
Only the second half of the trace is read (the steady decode loop).  Usage: python tools/trace_by_grid.py kernel_trace.csv"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"mw::\(anonymous namespace\)::", "", name)
    name = re.sub(r"_ZN2mw12_GLOBAL__N_1\d+", "", name)
    return name[:44]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), short(r["Kernel_Name"]),
                     int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
lo = t0 + (t1 - t0) // 2
last_end = {}
agg = defaultdict(lambda: [0, 0, 0])
for s, e, q, name, grid in rows:
    if s >= lo and q in last_end:
        a = agg[(name, grid)]
        a[0] += 1
        a[1] += e - s
        a[2] += s - last_end[q]
    last_end[q] = e
tot = sum(a[1] + a[2] for a in agg.values())
print(f"{'kernel':44s} {'grid':>9s} {'calls':>7s} {'mean us':>8s} {'gap us':>7s} {'share':>6s}")
for (name, grid), a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:18]:
    print(f"{name:44s} {grid:9d} {a[0]:7d} {a[1] / a[0] / 1e3:8.2f} {a[2] / a[0] / 1e3:7.2f} {100.0 * (a[1] + a[2]) / tot:5.1f}%")
