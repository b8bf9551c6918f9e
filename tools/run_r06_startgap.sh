#!/bin/bash
# what occupies the device between the optimiser's zero fill and the encoder's first kernel?  kernel + memory-copy trace of 3 steps
O=gpurun_out/r06x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof -o p -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dense-leg --no-dropin-leg --no-length-leg > $O/bench_prof.json 2> $O/bench_prof.err
ls $O/prof/*/ 2>/dev/null | head; ls $O/prof | head
t=$(find $O/prof -name "p_kernel_trace.csv" | head -1)
m=$(find $O/prof -name "p_memory_copy_trace.csv" | head -1)
python3 - "$t" "$m" <<'PY' > $O/startgap.log 2>&1
import csv, sys
k = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70]) for r in k)
try:
    m = list(csv.DictReader(open(sys.argv[2])))
    print("memory copies:", len(m), list(m[0].keys()) if m else None)
    mc = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY " + r.get("Direction", "") ) for r in m)
except Exception as e:
    print("no memory copy trace:", e); mc = []
allv = sorted(ev + mc)
steps = [i for i, e in enumerate(allv) if e[2].startswith("clip_adam_kernel")]
for si in steps[:-1]:
    t0 = allv[si][1]
    print("---- after clip_adam ending at", t0)
    for s, e, n in allv[si + 1: si + 14]:
        print(f"  start +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:8.1f} us  {n}")
PY
find $O -name "*trace.csv" -delete
cat $O/startgap.log | head -80
