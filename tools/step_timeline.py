#!/usr/bin/env python3
"""Where one step of the terrain workload spends its time with one frame in flight: reads the kernel trace
of `rocprofv3 --kernel-trace --output-format csv -- python3 bench.py --config 3 --images-in-flight 1 ...`
and prints, for the median step, every kernel's start relative to the step's first kernel, its duration
and the idle gap before it.
    python tools/step_timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
k = [(r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("::")[-1][:28], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
starts = [i for i, x in enumerate(k) if x[0].startswith("k_expand_triangles")]
steps = [(starts[j], starts[j + 1]) for j in range(len(starts) - 1)]
steps = steps[len(steps) // 2:]                     # warm steps
lens = sorted((k[b][1] - k[a][1], a, b) for a, b in steps)
span, a, b = lens[len(lens) // 2]
print(f"median step {span / 1000:.1f} us, {b - a} kernels")
t0 = k[a][1]
prev_end = None
busy = 0
for name, s, e in k[a:b]:
    gap = 0 if prev_end is None else s - prev_end
    busy += e - s
    if e - s > 8000 or gap > 4000:
        print(f"{(s - t0) / 1000:9.1f} us  {name:28s} {(e - s) / 1000:8.1f} us   gap before {gap / 1000:6.1f} us")
    prev_end = max(prev_end or 0, e)
print(f"busy {busy / 1000:.1f} us of {span / 1000:.1f}")
