"""Raw time line of a window of a rocprofv3 --kernel-trace CSV: every launch between the n-th and (n + 2)-th launch of a kernel whose name
contains <pattern>, with start offset, duration, queue and grid.   python3 tools/trace_window.py <trace.csv> <pattern> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat, n = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 40
ev = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(ev) if pat in r["Kernel_Name"]]
a, b = idx[n], idx[n + 2]
t0 = int(ev[a]["Start_Timestamp"])
for r in ev[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:70]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>9}  {name}")
