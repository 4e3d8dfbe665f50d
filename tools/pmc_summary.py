#!/usr/bin/env python3
"""Aggregate rocprofv3 counter CSVs (`--pmc ... --output-format csv`, one pass per directory) into one
table: per (kernel, grid size) the number of dispatches and the mean of every counter.
Usage: python tools/pmc_summary.py out.csv pass_dir [pass_dir ...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN2mw12_GLOBAL__N_1\d+([A-Za-z_0-9]+?)(?:I|E)", name)
    if m:
        return m.group(1)
    return name.split("(")[0][:60]


acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for v in acc.values() for c in v})
with open(sys.argv[1], "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["kernel", "grid_threads", "dispatches"] + [c + "_mean" for c in counters])
    for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(sum(x) for x in kv[1].values())):
        n = max(len(x) for x in v.values())
        w.writerow([k, g, n] + [f"{sum(v[c]) / len(v[c]):.1f}" if c in v else "" for c in counters])
print("wrote", sys.argv[1])
