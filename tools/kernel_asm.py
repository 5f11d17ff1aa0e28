#!/usr/bin/env python3
"""Extracts one kernel from a device assembly file (hipcc -S --cuda-device-only) and tallies its
instructions by unit: python tools/kernel_asm.py /tmp/mega.s <mangled-name substring> [out.s]"""
import collections
import re
import sys

path, needle = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and needle in l and l.rstrip().endswith(tuple("E:")) or (l.startswith("_Z") and needle in l and ": ;" in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(body))
tally = collections.Counter()
for l in body:
    m = re.match(r"\s+([a-z_0-9]+)", l)
    if not m:
        continue
    op = m.group(1)
    unit = ("salu" if op.startswith("s_") and not op.startswith(("s_load", "s_buffer", "s_waitcnt", "s_cbranch", "s_branch", "s_nop", "s_endpgm", "s_setprio", "s_barrier"))
            else "smem" if op.startswith(("s_load", "s_buffer")) else "branch" if op.startswith(("s_cbranch", "s_branch"))
            else "wait" if op.startswith(("s_waitcnt", "s_nop")) else "valu" if op.startswith("v_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_"))
            else "lds" if op.startswith("ds_") else "other")
    tally[unit] += 1
print(lines[start].split(":")[0][:90])
print(dict(tally), "total", sum(tally.values()))
