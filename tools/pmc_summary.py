#!/usr/bin/env python
"""Summarises rocprofv3 --pmc passes (counter_collection csv) for the headline's forward kernel: per-dispatch averages.
usage: pmc_summary.py <dir with p1 .. pN> [kernel name substring; default = the no-grad f16x3 forward of the bench model]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "siren16w_kernel<256, true, 0, false, 0>"      # <H, GRID, SAVE, FUSED, TERMS2>
agg = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    per_dispatch = defaultdict(dict)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if KERNEL not in row.get("Kernel_Name", ""):
                continue
            per_dispatch[row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
    for d, cs in per_dispatch.items():
        for k, v in cs.items():
            agg[k].append(v)
print(f"# kernel: {KERNEL}")
print("counter,avg_per_siren_dispatch,n_dispatches")
for k in sorted(agg):
    v = agg[k]
    print(f"{k},{sum(v) / len(v):.6g},{len(v)}")
if "FETCH_SIZE" in agg:
    f = sum(agg["FETCH_SIZE"]) / len(agg["FETCH_SIZE"])
    print(f"# FETCH_SIZE is in KiB; gfx950 correction x2 for wide coalesced reads (MI355X_MICROARCH.md §HBM): "
          f"{f * 1024 * 2 / 1e6:.2f} MB fetched per launch (uncorrected {f * 1024 / 1e6:.2f} MB)")
if "WRITE_SIZE" in agg:
    w = sum(agg["WRITE_SIZE"]) / len(agg["WRITE_SIZE"])
    print(f"# WRITE_SIZE (KiB, uncalibrated on gfx950): {w * 1024 / 1e6:.2f} MB written per launch")
