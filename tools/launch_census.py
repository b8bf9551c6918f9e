#!/usr/bin/env python3
"""Launches per training step by kernel name, from the *_kernel_stats.csv of a `rocprofv3 --kernel-trace --stats` run of bench.py.
    python3 tools/launch_census.py <p_kernel_stats.csv> <steps traced (timed + warm-up)>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
rows.sort(key=lambda r: -int(r["Calls"]))
tot = sum(int(r["Calls"]) for r in rows)
print(f"{tot} launches over {steps} steps = {tot / steps:.1f} per step (includes the few launches outside the steps)")
aten = sum(int(r["Calls"]) for r in rows if "at::native" in r["Name"] or "rocclr" in r["Name"])
print(f"of them from PyTorch (at::native, copyBuffer): {aten / steps:.1f} per step")
for r in rows:
    if int(r["Calls"]) < steps:
        break
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{int(r['Calls']) / steps:7.1f}  {int(r['TotalDurationNs']) / steps / 1e3:9.0f} us/step  {name[:120]}")
