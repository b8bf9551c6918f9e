#!/bin/bash
# fabric traffic of the encoder's pointwise convolutions (separate PMC passes)
O=gpurun_out/r03p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/bench_pointwise.py > $O/pointwise.log 2>&1
i=0
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctr --kernel-include-regex 'h2_kernel' --output-format csv -d $O/pmc_$i -o p -- python3 tools/bench_pointwise.py > $O/pmc_$i.log 2>&1
done
python3 - <<'PY'
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r03p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
        acc[(n, r["Grid_Size"], r["LDS_Block_Size"] if "LDS_Block_Size" in r else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}, len(next(iter(cs.values()))))
PY
find $O -name "*.csv" -size +3M -delete
tail -8 $O/pointwise.log
