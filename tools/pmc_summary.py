#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel and counter.

    python tools/pmc_summary.py <dir with */*/*_counter_collection.csv> [kernel substring]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
match = sys.argv[2] if len(sys.argv) > 2 else ""
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if match and match not in k:
            continue
        k = k.split("(")[0][-60:]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
for k in sorted(tot):
    print(k)
    for c in sorted(tot[k]):
        n = len(calls[(k, c)])
        print(f"  {c:32s} total {tot[k][c]:.6g}  per-dispatch {tot[k][c] / max(n, 1):.6g}  dispatches {n}")
