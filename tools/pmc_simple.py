"""average of every counter per kernel name over the launches with the largest grid:  python tools/pmc_simple.py <dir> [name filter]"""
import collections, csv, glob, os, sys
root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "h2_kernel"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if flt in n and int(r["Grid_Size"]) >= 500000:
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in sorted(acc.items()):
    print(n, {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}, "launches", {c: len(v) for c, v in cs.items()})
