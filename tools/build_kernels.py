#!/usr/bin/env python3
"""Average duration of the LBVH build's kernels from the kernel trace tools/profile_build.sh writes:
    python tools/build_kernels.py gpurun_out/prof_build_<tag>/kernel_stats.csv
Prints microseconds per build for every kernel of the build (calls per build x average) and their sum."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
builds = max(int(r["Calls"]) for r in rows if "k_refit_local" in r["Name"])
total = 0.0
for r in rows:
    name = r["Name"]
    if not any(k in name for k in ("k_refit", "k_rs_", "k_scan", "k_hierarchy", "k_leaf_boxes", "k_permute", "k_morton",
                                   "k_expand", "fillBuffer", "k_collapse", "k_tile_order", "k_build_init")):
        continue
    per = int(r["Calls"]) / builds
    us = per * float(r["AverageNs"]) / 1000.0
    total += us
    short = name.replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1]
    print(f"{short:28s} x{per:5.2f}  {float(r['AverageNs']) / 1000.0:7.1f} us  min {int(r['MinNs']) / 1000.0:7.1f}  -> {us:7.1f} us per build")
print(f"sum {total:.1f} us per build ({builds} builds)")
