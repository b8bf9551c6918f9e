"""Summarise a rocprofv3 --kernel-trace --stats run of bench.py into profiles/<tag>_*.  Usage:
   python tools/summarize_prof.py gpurun_out/prof_r01/runc/953 profiles/r01"""
import csv, sys, collections, shutil
src, dst = sys.argv[1], sys.argv[2]
shutil.copy(src + "_kernel_stats.csv", dst + "_bench_kernel_stats.csv")
rows = list(csv.DictReader(open(src + "_kernel_trace.csv")))
by = collections.defaultdict(list)
for r in rows:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    name = name.split("(")[0].replace("void ", "")
    key = (name, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"]), int(r.get("Grid_Size_Y", 1) or 1))
    by[key].append(dur)
tot = sum(sum(v) for v in by.values())
lines = ["| kernel | grid.x (threads) | grid.y | launches | avg us | total ms | % |", "|---|---|---|---|---|---|---|"]
for (name, gx, gy), v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:40]:
    lines.append(f"| {name} | {gx} | {gy} | {len(v)} | {sum(v)/len(v):.1f} | {sum(v)/1e3:.2f} | {100*sum(v)/tot:.2f} |")
open(dst + "_bench_by_shape.md", "w").write("\n".join(lines) + f"\n\ntotal kernel time {tot/1e3:.1f} ms\n")
print("\n".join(lines[:16]))
