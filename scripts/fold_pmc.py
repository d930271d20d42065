#!/usr/bin/env python3
"""Fold rocprofv3 counter_collection.csv files (one row per dispatch and counter) into one row per (kernel, grid,
counter): dispatches, sum, min, max — what profiles/ keeps of a PMC pass (the raw files run to megabytes).
usage: scripts/fold_pmc.py file_counter_collection.csv ...   (writes *_folded.csv next to each, removes the input)"""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_pmc import short  # noqa: E402

for f in sys.argv[1:]:
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        key = (k, r["Grid_Size"], r["Workgroup_Size"], r["Counter_Name"])
        v = float(r["Counter_Value"])
        a = acc.setdefault(key, [0, 0.0, v, v])
        a[0] += 1
        a[1] += v
        a[2] = min(a[2], v)
        a[3] = max(a[3], v)
    out = f.replace("_counter_collection.csv", "_folded.csv")
    with open(out, "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(["kernel", "grid_size", "workgroup_size", "counter", "dispatches", "sum", "min", "max"])
        for (k, gs, ws, c), a in acc.items():
            w.writerow([k, gs, ws, c, a[0], repr(a[1]), repr(a[2]), repr(a[3])])
    os.remove(f)
