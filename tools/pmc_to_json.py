#!/usr/bin/env python3
"""Folds the pmc_cfgN.txt summaries of tools/profile_configs.sh into profiles/pmc_latest.json,
the file bench.py reads its `traffic` and counter ratios from.

    python tools/pmc_to_json.py profiles/r2 [tag shown in the source strings]

Per workload and per step of its dominant kernel (the variant with the most dispatches; if a
step launches several kernels that share that count, their counts are added, and since a counter
run serialises them the cycles of the step then come from the measured step time in
bench_cfgN.json at 2.4 GHz):
  hbm_bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE (KB -> bytes; FETCH_SIZE reads half the
  bytes of wide coalesced reads on gfx950), cycles = GRBM_GUI_ACTIVE / 8 XCDs,
  valu_instructions_per_simd_cycle = SQ_INSTS_VALU / (1024 SIMDs * cycles),
  valu_lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU),
  waves_per_simd = 4 * SQ_WAVE_CYCLES / (1024 * cycles)   (the counter ticks every 4 cycles),
  wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES, l2_hit_rate = TCC_HIT / (TCC_HIT + TCC_MISS),
  l1_hit_rate = 1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES.
"""
import json
import os
import re
import sys

src = sys.argv[1]
ISSUE_CAPACITY = 0.40   # wave-level VALU instructions one SIMD issues per cycle (tools/micro/issue_cost.hip)
HW_ISSUE_RATE = 0.5     # the hardware rate: a wave64 VALU instruction occupies a SIMD-32 for 2 cycles


def parse(path):
    kernels, cur = {}, None
    for line in open(path):
        if not line.startswith(" ") and line.strip() and not line.startswith("#"):
            cur = line.strip()
            kernels[cur] = {}
        elif cur and line.startswith("  "):
            m = re.match(r"\s+(\S+)\s+total\s+(\S+)\s+per-dispatch\s+(\S+)\s+dispatches\s+(\d+)", line)
            if m:
                kernels[cur][m.group(1)] = (float(m.group(3)), int(m.group(4)))
    return kernels


out = {"configs": {}}
for c in ("0", "1", "2", "3", "4"):
    path = os.path.join(src, f"pmc_cfg{c}.txt")
    if not os.path.exists(path):
        continue
    kernels = parse(path)
    if not kernels:
        continue
    count = lambda kk: max((v[1] for v in kernels[kk].values()), default=0)
    top = max(count(kk) for kk in kernels)
    names = [kk for kk in kernels if count(kk) == top and "true>" not in kk.split(",")[-1]] or \
            [max(kernels, key=count)]
    name = " + ".join(names)
    k = {}
    for kk in names:
        for n, v in kernels[kk].items():
            k[n] = k.get(n, 0.0) + v[0]
    step_cycles = None
    if len(names) > 1:
        k.pop("GRBM_GUI_ACTIVE", None)
        try:
            line = [l for l in open(os.path.join(src, f"bench_cfg{c}.json")) if l.startswith("{")][-1]
            step_cycles = json.loads(line)["config"]["ms_per_launch"] * 2.4e6
        except (OSError, KeyError, IndexError, ValueError):
            pass
    e = {"kernel": name, "source": f"{path} (rocprofv3 --pmc passes over `bench.py --config {c} --steps 4 "
                                   "--no-side-runs`, one run per pass; FETCH_SIZE doubled, gfx950)"}
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        e["FETCH_SIZE_KB_per_launch"] = k["FETCH_SIZE"]
        e["WRITE_SIZE_KB_per_launch"] = k["WRITE_SIZE"]
        e["hbm_bytes_per_launch"] = int((2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024)
    if "GRBM_GUI_ACTIVE" in k or step_cycles:
        cycles = step_cycles or k["GRBM_GUI_ACTIVE"] / 8.0
        e["cycles_per_launch"] = cycles
        if step_cycles:
            e["cycles_source"] = "ms_per_launch of bench_cfg%s.json x 2.4 GHz (the kernels of a step overlap)" % c
        if "SQ_INSTS_VALU" in k:
            e["valu_instructions_per_launch"] = k["SQ_INSTS_VALU"]
            e["valu_instructions_per_simd_cycle"] = round(k["SQ_INSTS_VALU"] / (1024 * cycles), 4)
            e["valu_issue_capacity_per_simd_cycle"] = ISSUE_CAPACITY
            e["valu_issue_frac"] = round(k["SQ_INSTS_VALU"] / (1024 * cycles) / ISSUE_CAPACITY, 3)
            e["valu_issue_frac_of_hw_rate"] = round(k["SQ_INSTS_VALU"] / (1024 * cycles) / HW_ISSUE_RATE, 3)
        if "SQ_WAVE_CYCLES" in k:
            e["waves_per_simd"] = round(4 * k["SQ_WAVE_CYCLES"] / (1024 * cycles), 2)
    if "SQ_THREAD_CYCLES_VALU" in k and "SQ_ACTIVE_INST_VALU" in k:
        e["valu_lane_utilisation"] = round(k["SQ_THREAD_CYCLES_VALU"] / (64 * k["SQ_ACTIVE_INST_VALU"]), 3)
    if "SQ_WAIT_ANY" in k and "SQ_WAVE_CYCLES" in k:
        e["wait_share_of_wave_cycles"] = round(k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"], 3)
    if "TCC_HIT_sum" in k and "TCC_MISS_sum" in k:
        e["l2_hit_rate"] = round(k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]), 3)
    if "TCP_TCC_READ_REQ_sum" in k and "TCP_TOTAL_CACHE_ACCESSES_sum" in k:
        e["l1_hit_rate"] = round(1 - k["TCP_TCC_READ_REQ_sum"] / k["TCP_TOTAL_CACHE_ACCESSES_sum"], 3)
        # vector L1 -> L2 read requests, taken as 64 bytes each (the size FETCH_SIZE tallies requests at)
        e["l1_to_l2_bytes_per_launch"] = int(k["TCP_TCC_READ_REQ_sum"] * 64)
    for extra in ("SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU_TRANS"):
        if extra in k:
            e[extra.lower() + "_per_launch"] = k[extra]
    out["configs"][c] = e
# what ties these figures to the code they were measured on: the fingerprint of the kernel sources
# (bench.py recomputes it and flags a mismatch) and the commit this file was built at
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import subprocess  # noqa: E402

from bench import source_fingerprint  # noqa: E402

out["source_sha"] = source_fingerprint()
try:
    out["commit"] = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    out["commit"] = None
lat = os.path.join(src, "load_latency.json")
if os.path.exists(lat):
    d = json.load(open(lat))
    out["load_latency_cycles"] = {k: v for k, v in d.items() if isinstance(v, float)}
    out["load_latency_source"] = f"{lat} (tools/micro/load_latency.hip: {d.get('unit')})"
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(src.rstrip("/"))), "pmc_latest.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
