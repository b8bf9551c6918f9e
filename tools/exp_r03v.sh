#!/bin/bash
O=gpurun_out/r03v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
f=$(find $O/prof -name "p_kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.Counter(); tim=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    if any(k in n for k in ("CUDAFunctor_add","direct_copy","FillFunctor","CatArray","copyBuffer","amax_kernel","colsum","sp_zero")):
        key=(n.replace("void at::native::","")[:60], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X","?"))
        acc[key]+=1; tim[key]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(acc.items(), key=lambda x:-x[1])[:45]:
    print(v/3, round(tim[k]/v/1e3,1), k)
PY
find $O -name "*trace.csv" -delete
